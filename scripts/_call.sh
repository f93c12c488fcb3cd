set -x
mkdir -p gpurun_out/r04
python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/call6_pytest.log
python scripts/ubench_tail.py > gpurun_out/r04/call6_tail.log 2>&1
python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
python scripts/bench_chz.py 30 > gpurun_out/r04/call6_chz.log 2>&1
python scripts/bench_front.py 30 > gpurun_out/r04/call6_front.log 2>&1
python scripts/symbol_seam_rate.py > gpurun_out/r04/call6_symbols.log 2>&1
cat gpurun_out/r04/call6_pytest.log; tail -7 gpurun_out/r04/call6_tail.log; cat gpurun_out/r04/call6_chz.log gpurun_out/r04/call6_front.log gpurun_out/r04/call6_symbols.log; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline_compute']['frac'], d['power'])
print(d['roofline']['other_kernels_ms_per_step'])
print({k:(v['kernel_ms'],v['value']) for k,v in d['other_slicer_specs'].items()})
s=d['secondary']; print(s['value'], s['roofline']['kernel_ms'], s['roofline']['frac']); print(s['latency'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['all_cores_value'])
"
