#!/bin/bash
# usage (GPU box): scripts/ab_run_front.sh <rounds> name1 name2 ...   -- bench_front.py with each scripts/variants/<name>.so in turn
N=${1:-2}; shift
for i in $(seq $N); do
  for v in "$@"; do
    echo -n "$v: "; AMPS_RECC_LIB=$PWD/scripts/variants/$v.so timeout 300 python scripts/bench_front.py 40 2>&1 | tail -3 | cut -c1-40 | tr '\n' ' '; echo
  done
done
