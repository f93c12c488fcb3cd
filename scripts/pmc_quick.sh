#!/bin/bash
# quick PMC passes on the bench (front kernel): usage scripts/pmc_quick.sh <tag>
TAG=${1:-q}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
SHORT="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o pmc -- $SHORT > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o pmc -- $SHORT > $OUT/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc -- $SHORT > $OUT/pmc3.log 2>&1
python $R/scripts/pmc_summary.py $OUT
tail -3 $OUT/pmc3.log | cut -c1-200
