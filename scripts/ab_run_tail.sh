#!/bin/bash
# usage (GPU box): scripts/ab_run_tail.sh <rounds> name1 name2 ...  -- scripts/ubench_tail.py (wideband rows) with each scripts/variants/<name>.so in turn
N=${1:-2}; shift
for i in $(seq $N); do
  for v in "$@"; do
    echo "$v: $(AMPS_RECC_LIB=$PWD/scripts/variants/$v.so timeout 300 python scripts/ubench_tail.py wide 2>&1 | grep 'bursts=416' | cut -c30-120)"
  done
done
