set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_slicer_specs.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15 > gpurun_out/r04/call1_pytest.log
python scripts/bench_chz.py 30 > gpurun_out/r04/call1_chz.log 2>&1
python scripts/bench_front.py 30 > gpurun_out/r04/call1_front.log 2>&1
python scripts/slicer_sensitivity.py 1000 > gpurun_out/r04/call1_sens.log 2>&1
cat gpurun_out/r04/call1_pytest.log gpurun_out/r04/call1_chz.log gpurun_out/r04/call1_front.log; tail -5 gpurun_out/r04/call1_sens.log
