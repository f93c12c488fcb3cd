set -x
mkdir -p gpurun_out/r04
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r04/final2_pytest.log
bash scripts/profile_round.sh r04 default > gpurun_out/profile_r04.log 2>&1
python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
python scripts/bench_chz.py 40 > gpurun_out/r04/final2_chz.log 2>&1
timeout 200 python scripts/fuzz_wideband.py 900 60 > gpurun_out/r04/fuzz_wideband2.log 2>&1
cat gpurun_out/r04/final2_pytest.log; head -4 gpurun_out/prof_r04/kernel_stats.csv | cut -c1-140; grep -v amdgpu gpurun_out/r04/final2_chz.log; tail -n 1 gpurun_out/r04/fuzz_wideband2.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline_compute']['frac'], d.get('power'))
s=d['secondary']; print('secondary', s['value'], s['roofline']['kernel_ms'], s['roofline']['frac'])
print({k:(v['kernel_ms'],v['value']) for k,v in d['other_slicer_specs'].items()})
PY
