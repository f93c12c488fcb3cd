set -x
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py tests/test_gpu_slicer_specs.py tests/test_gpu_sync_tolerance.py -q -x 2>&1 | tail -5 > gpurun_out/r04/call8_pytest.log
( echo "current library (A, D depth 1; B, C depth 2 at 3 waves)"; python scripts/bench_front.py 40
  echo "depth 2 everywhere, 3 waves"; AMPS_RECC_DEPTH=2 python scripts/bench_front.py 40
  echo "depth 2 everywhere, compiled for 4 waves per SIMD"; AMPS_RECC_DEPTH=2 AMPS_RECC_LIB=$PWD/scripts/variants/d2o4.so python scripts/bench_front.py 40
  echo "current again"; python scripts/bench_front.py 40 ) > gpurun_out/r04/call8_front.log 2>&1
cat gpurun_out/r04/call8_pytest.log; grep -v amdgpu.ids gpurun_out/r04/call8_front.log
