set -x
mkdir -p gpurun_out/r04
python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r04/final_pytest.log
timeout 500 python scripts/fuzz_parity.py 10000 7 device > gpurun_out/r04/fuzz_iq_device.log 2>&1
timeout 200 python scripts/fuzz_parity.py 2000 8 > gpurun_out/r04/fuzz_iq_host.log 2>&1
timeout 400 python scripts/fuzz_wideband.py 600 100 > gpurun_out/r04/fuzz_wideband.log 2>&1
timeout 200 python scripts/fuzz_symbols_decode.py 300 > gpurun_out/r04/fuzz_symbols.log 2>&1
bash scripts/profile_round.sh r04 default > gpurun_out/profile_r04.log 2>&1
python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
python bench.py --workload direct1 --secondary none --no-cpu-baseline --steps 2000 > gpurun_out/r04/bench_direct1.json 2>/dev/null
python bench.py --no-pipeline --no-cpu-baseline --no-other-specs --steps 2000 > gpurun_out/r04/bench_nopipeline.json 2>/dev/null
python scripts/bench_front.py 40 > gpurun_out/r04/final_front.log 2>&1
cat gpurun_out/r04/final_pytest.log
for f in fuzz_iq_device fuzz_iq_host fuzz_wideband fuzz_symbols; do grep -v amdgpu.ids gpurun_out/r04/$f.log | tail -n 2 | cut -c1-300; done
head -5 gpurun_out/prof_r04/kernel_stats.csv | cut -c1-150
grep -v amdgpu.ids gpurun_out/r04/final_front.log
python - <<'PY'
import json
for n in ("bench_default","bench_direct1","bench_nopipeline"):
    d=json.loads([l for l in open('gpurun_out/r04/%s.json'%n) if l.startswith('{')][-1])
    print(n, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('power'))
    if 'secondary' in d: s=d['secondary']; print('  secondary', s['value'], s['roofline']['kernel_ms'], s['roofline']['frac'], s['roofline']['other_kernels_ms_per_step'], s.get('latency',{}).get('wideband_seam_832_channels'))
    if 'other_slicer_specs' in d: print('  ', {k:(v['kernel_ms'],v['value']) for k,v in d['other_slicer_specs'].items()})
PY
