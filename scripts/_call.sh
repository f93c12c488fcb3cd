set -x
mkdir -p gpurun_out/r04
python scripts/impairment_sweep.py 500 > gpurun_out/r04/impairments_final.txt 2>&1
head -3 gpurun_out/r04/impairments_final.txt; tail -3 gpurun_out/r04/impairments_final.txt
