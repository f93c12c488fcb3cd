set -x
mkdir -p gpurun_out/r04
scripts/ab_run_variants.sh exact 3 cur split splitp > gpurun_out/r04/call11_split.log 2>&1
AMPS_RECC_LIB=$PWD/scripts/variants/split.so python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_slicer_specs.py tests/test_gpu_channel_groups.py -q -x -k "exact or groups" 2>&1 | tail -4 >> gpurun_out/r04/call11_split.log
cat gpurun_out/r04/call11_split.log
