#!/bin/bash
# EXPERIMENT (round 6): L2 touch of the fold role's input by the pass-2 role, CHZ_TOUCH_AHEAD = 1 .. 4 half-steps ahead (D = 768, spec D)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/chz_touch_ab.txt; : > $OUT
for i in 1 2 3; do
  for v in base touch1 touch2 touch3 touch4; do
    echo -n "$v: " >> $OUT; CHZ_DECIM=768 AMPS_RECC_LIB=$PWD/scripts/variants/$v.so timeout 300 python scripts/bench_chz.py 200 exact 400 2>&1 | tail -1 >> $OUT
  done
done
echo "== parity with touch2.so" >> $OUT
AMPS_RECC_LIB=$PWD/scripts/variants/touch2.so timeout 900 python -m pytest tests/test_gpu_channelizer.py tests/test_gpu_wideband_vs_reference.py tests/test_gpu_fullsize.py -x -q -k "D768" 2>&1 | tail -2 >> $OUT
cat $OUT
