set -x
# one round's profile evidence: rocprofv3 stats + PMC of the default bench command, then the bench lines themselves (GPU box)
T=${1:-r06}
mkdir -p gpurun_out/$T
# the bench lines first, on a box nothing has run on yet (round 6: a sustained run right behind the five rocprofv3 passes read its sampled
# events 20 us long on one box), then the profile passes
python bench.py > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/$T/bench_steps20_as_the_driver_runs_it.json 2>/dev/null
bash scripts/profile_round.sh $T default > gpurun_out/profile_round.log 2>&1
cp gpurun_out/prof_$T/kernel_stats.csv gpurun_out/prof_$T/pmc_kernels.txt gpurun_out/prof_$T/traffic.json gpurun_out/$T/ 2>/dev/null
grep -h "^{" gpurun_out/prof_$T/trace.log | tail -1 > gpurun_out/$T/bench_under_rocprofv3.json
python bench.py --workload direct1 --secondary none --no-cpu-baseline --steps 2000 > gpurun_out/$T/bench_direct1.json 2>/dev/null
python bench.py --decim 512 --no-cpu-baseline --no-other-decim --no-other-specs --secondary none --steps 4000 > gpurun_out/$T/bench_decim512.json 2>/dev/null
tail -c 400 gpurun_out/$T/bench_default.json
