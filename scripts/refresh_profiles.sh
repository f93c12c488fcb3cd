set -x
# one round's profile evidence: rocprofv3 stats + PMC of the default bench command, then the bench lines themselves (GPU box)
T=${1:-r03}
bash scripts/profile_round.sh $T atan > gpurun_out/profile_round.log 2>&1
mkdir -p gpurun_out/$T
cp gpurun_out/prof_$T/kernel_stats.csv gpurun_out/prof_$T/pmc_kernels.txt gpurun_out/prof_$T/traffic.json gpurun_out/$T/ 2>/dev/null
python bench.py > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err
python bench.py --slicer sine --no-cpu-baseline --steps 2000 > gpurun_out/$T/bench_sine.json 2>/dev/null
python bench.py --workload direct1 --secondary none --no-cpu-baseline --steps 2000 > gpurun_out/$T/bench_direct1.json 2>/dev/null
python bench.py --no-pipeline --no-cpu-baseline --steps 2000 > gpurun_out/$T/bench_nopipeline.json 2>/dev/null
tail -c 600 gpurun_out/$T/bench_default.json
