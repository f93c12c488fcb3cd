#!/bin/bash
# usage (GPU box): scripts/ab_run.sh [spec] [rounds]   -- alternates base.so and the working tree's library
SPEC=${1:-sine}; N=${2:-3}
for i in $(seq $N); do
  echo -n "base: "; AMPS_RECC_LIB=$PWD/scripts/variants/base.so python scripts/bench_chz.py 40 $SPEC 2>&1 | tail -1
  echo -n "new:  "; python scripts/bench_chz.py 40 $SPEC 2>&1 | tail -1
done
