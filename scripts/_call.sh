set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_channel_groups.py -q -x 2>&1 | tail -5 > gpurun_out/r04/call9_pytest.log
timeout 600 python scripts/fuzz_parity.py 6000 4 device > gpurun_out/r04/fuzz_iq_device.log 2>&1
timeout 300 python scripts/fuzz_parity.py 1500 5 > gpurun_out/r04/fuzz_iq_host.log 2>&1
timeout 300 python scripts/fuzz_wideband.py 400 40 > gpurun_out/r04/fuzz_wideband.log 2>&1
timeout 200 python scripts/fuzz_symbols_decode.py 150 > gpurun_out/r04/fuzz_symbols.log 2>&1
cat gpurun_out/r04/call9_pytest.log; tail -3 gpurun_out/r04/fuzz_iq_device.log gpurun_out/r04/fuzz_iq_host.log gpurun_out/r04/fuzz_wideband.log gpurun_out/r04/fuzz_symbols.log
