set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_slicer_specs.py tests/test_gpu_channel_groups.py tests/test_gpu_channelizer.py tests/test_gpu_fuzz.py tests/test_gpu_wideband_vs_reference.py -q -x 2>&1 | tail -4 > gpurun_out/r04/call15_pytest.log
python scripts/bench_chz.py 40 > gpurun_out/r04/call15_chz.log 2>&1
AMPS_RECC_LIB=$PWD/scripts/variants/cur.so python scripts/bench_chz.py 40 exact,atan >> gpurun_out/r04/call15_chz.log 2>&1
python scripts/bench_chz.py 40 exact >> gpurun_out/r04/call15_chz.log 2>&1
cat gpurun_out/r04/call15_pytest.log; grep -v amdgpu gpurun_out/r04/call15_chz.log
