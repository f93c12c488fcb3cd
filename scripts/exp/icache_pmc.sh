#!/bin/bash
# EXPERIMENT (round 6): instruction-cache behaviour of the filter bank (67 KB of code against a 64 KB instruction cache shared by two CUs)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" | head -40 > $OUT/counters.txt
SHORT="python $R/bench.py --steps 4 --warmup 2 --prewarm-ms 50 --no-cpu-baseline --no-other-specs --no-latency --no-other-decim --secondary none"
rocprofv3 --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/pmc -o pmc -- $SHORT > $OUT/pmc.log 2>&1
python $R/scripts/pmc_summary.py $OUT ", 768>(" > $OUT/summary.txt 2>&1
python $R/scripts/pmc_summary.py $OUT "recc_resolve_kernel<256, 512, true" >> $OUT/summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -delete
cat $OUT/counters.txt | head -30; cat $OUT/summary.txt; tail -3 $OUT/pmc.log
