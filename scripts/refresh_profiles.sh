set -x
bash scripts/profile_round.sh r02 sine > gpurun_out/profile_round.log 2>&1
mkdir -p gpurun_out/r02
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err
python bench.py --slicer atan --no-cpu-baseline --steps 2000 > gpurun_out/r02/bench_atan.json 2>/dev/null
python bench.py --workload direct1 --secondary none --no-cpu-baseline --steps 2000 > gpurun_out/r02/bench_direct1.json 2>/dev/null
python bench.py --no-pipeline --no-cpu-baseline --steps 2000 > gpurun_out/r02/bench_nopipeline.json 2>/dev/null
tail -c 600 gpurun_out/r02/bench_default.json
