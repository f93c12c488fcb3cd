set -x
mkdir -p gpurun_out/r04
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r04/call3_pytest.log
( AMPS_RECC_LIB=$PWD/scripts/variants/tl.so python scripts/chz_timeline.py exact; AMPS_RECC_LIB=$PWD/scripts/variants/tlp.so python scripts/chz_timeline.py exact; AMPS_RECC_LIB=$PWD/scripts/variants/tl.so python scripts/chz_timeline.py sine ) > gpurun_out/r04/call3_timeline.log 2>&1
python scripts/bench_front.py 30 > gpurun_out/r04/call3_front.log 2>&1
python bench.py --steps 3000 > gpurun_out/r04/call3_bench.json 2> gpurun_out/r04/call3_bench.err
cat gpurun_out/r04/call3_pytest.log gpurun_out/r04/call3_timeline.log gpurun_out/r04/call3_front.log; tail -3 gpurun_out/r04/call3_bench.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/call3_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline_compute']['frac'])
print({k:(v['kernel_ms'],v['value']) for k,v in d['other_slicer_specs'].items()})
s=d['secondary']; print(s['value'], s['roofline']['kernel_ms'], s['roofline']['frac']); print(s['latency'])
print(d['cpu_baseline'])
"
