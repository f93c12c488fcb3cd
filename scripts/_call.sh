set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pins.py tests/test_gpu_resolve_forms.py tests/test_second_restatement.py tests/test_gpu_host_blocks.py -q -x 2>&1 | tail -4 > gpurun_out/r04/call14_pytest.log
python scripts/ubench_tail.py > gpurun_out/r04/call14_tail.log 2>&1
cat gpurun_out/r04/call14_pytest.log; tail -6 gpurun_out/r04/call14_tail.log
