#!/bin/bash
# usage (GPU box): scripts/ab_run_variants.sh "<specs>" <rounds> name1 name2 ...   -- bench_chz.py with each scripts/variants/<name>.so in turn
SPECS=${1:-sine}; N=${2:-2}; shift 2
for i in $(seq $N); do
  for v in "$@"; do
    echo -n "$v: "; AMPS_RECC_LIB=$PWD/scripts/variants/$v.so timeout 300 python scripts/bench_chz.py 40 $SPECS 2>&1 | tail -$(echo $SPECS | tr ',' '\n' | wc -l) | cut -c1-28 | tr '\n' ' '; echo
  done
done
