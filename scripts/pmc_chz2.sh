#!/bin/bash
# PMC passes over the filter-bank kernel alone (scripts/bench_chz.py, one slicer spec).  usage: scripts/pmc_chz2.sh <tag> <spec>
TAG=${1:-chz}; SPEC=${2:-sine}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/bench_chz.py 4 $SPEC 4"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o pmc -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVE_DEP_WAIT SQ_INST_LEVEL_LDS SQ_IFETCH SQ_ACTIVE_INST_FLAT -d $OUT/pmc3 -o pmc -- $CMD > $OUT/pmc3.log 2>&1
for k in chz; do echo "== $k"; python $R/scripts/pmc_summary.py $OUT $k; done | tee $OUT/summary.txt
find $OUT -name "*counter_collection.csv" -delete
