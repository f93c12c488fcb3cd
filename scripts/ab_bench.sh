#!/bin/bash
# usage (GPU box): scripts/ab_bench.sh [workload] [rounds]  -- bench.py's timed region with base.so (scripts/ab_build.sh) and the working tree's library, alternating
W=${1:-direct832}; N=${2:-3}
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["other_kernels_ms_per_step"])'
for i in $(seq $N); do
  echo -n "base: "; AMPS_RECC_LIB=$PWD/scripts/variants/base.so python bench.py --workload $W --secondary none --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"
  echo -n "new:  "; python bench.py --workload $W --secondary none --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"
done
