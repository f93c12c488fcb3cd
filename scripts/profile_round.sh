#!/bin/bash
# rocprofv3 evidence for one round: kernel-trace stats of the default bench command and separate PMC passes.
# usage (GPU box, repo root): scripts/profile_round.sh <tag> [slicer]      -> gpurun_out/prof_<tag>/
TAG=${1:-r04}; SL=${2:-default}     # default = the library default (spec D since round 4) = what `python bench.py` runs
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-other-specs --no-latency --slicer $SL"          # the default command (12000 steps of wideband832 + 2000 of direct832) without its CPU legs
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
SHORT="python $R/bench.py --steps 4 --warmup 2 --prewarm-ms 50 --no-cpu-baseline --no-other-specs --no-latency --slicer $SL"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o pmc -- $SHORT > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc -- $SHORT > $OUT/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $SHORT > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $SHORT > $OUT/pmc_write.log 2>&1
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
for k in ", 768>(" ", 512>(" "recc_front_kernel<10" "recc_bits_kernel<2" "recc_bits_kernel<3" "recc_resolve_kernel<256, 512, true" "recc_resolve_kernel<256, 512, false" "recc_symbols_kernel"; do echo "== $k"; python $R/scripts/pmc_summary.py $OUT "$k"; done | tee $OUT/pmc_kernels.txt
SLNAME=$SL; if [ "$SL" = "default" ]; then SLNAME=$(cd $R && python -c "from gr_amps_amd import capi; print(capi.SLICER_NAMES[capi.load().amps_recc_default_slicer()])"); fi
python $R/scripts/make_traffic_json.py $OUT $SLNAME > $OUT/traffic.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
# keep the summaries, drop the raw per-dispatch tables (tens of MB: gpurun merges at most 64 MiB back)
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
du -sh $OUT
