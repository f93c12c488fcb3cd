#!/bin/bash
# first GPU call of round 3: new three-role filter-bank kernel -- parity tests of the wideband seam, then A/B against round 2's library
mkdir -p gpurun_out
{
echo "== channelizer tests"; timeout 600 python -m pytest tests/test_gpu_channelizer.py tests/test_gpu_slicer_specs.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15
echo "== A/B (base = round 2)"; 
for i in 1 2; do
  echo -n "base: "; AMPS_RECC_LIB=$PWD/scripts/variants/base.so timeout 300 python scripts/bench_chz.py 40 sine,atan 2>&1 | tail -2 | tr '\n' ' '; echo
  echo -n "new:  "; timeout 300 python scripts/bench_chz.py 40 sine,atan,product 2>&1 | tail -3 | tr '\n' ' '; echo
done
} > gpurun_out/call1.log 2>&1
tail -30 gpurun_out/call1.log
