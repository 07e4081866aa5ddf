#!/bin/bash
# tools/profile_round.sh <tag> -- run on the GPU box from the repo root (gpurun): rocprofv3 kernel-trace stats and the two HBM PMC passes
# of the default bench command, summarised into gpurun_out/prof_<tag>/ ; tools/pmc_summary.py turns them into profiles/<tag>_*.csv.
# PMC passes use --kernel-trace only (never combined with sys/hip/hsa tracing).
TAG=${1:-r01_c}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $BENCH > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o sq -- $BENCH > $OUT/sq.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -20
python tools/pmc_summary.py $OUT $TAG
