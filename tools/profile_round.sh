#!/bin/bash
# tools/profile_round.sh <tag> -- run on the GPU box from the repo root (gpurun): rocprofv3 kernel-trace stats and PMC passes
#   (a) of the default bench command (the extract + match path): stats, FETCH_SIZE, WRITE_SIZE, SQ pass, SQ_INSTS_VALU pass
#   (b) of tools/all_kernels.py (every other kernel of SURVEY 8a / 8f): stats, FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU pass
# summarised into gpurun_out/prof_<tag>/ ; tools/pmc_summary.py turns them into <tag>_*.csv, traffic.json and valu_mix.json (copy into profiles/).
# PMC passes use --kernel-trace only (never combined with sys / hip / hsa tracing).
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --passes 1 --no-cpu-baseline --no-extras"
ALL="python $REPO/tools/all_kernels.py 2"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $BENCH > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o sq -- $BENCH > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/insts -o insts -- $BENCH > $OUT/insts.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/all_stats -o stats -- $ALL > $OUT/all_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/all_fetch -o fetch -- $ALL > $OUT/all_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/all_write -o write -- $ALL > $OUT/all_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/all_insts -o insts -- $ALL > $OUT/all_insts.log 2>&1
cd $REPO
tail -2 $OUT/all_stats.log
python tools/pmc_summary.py $OUT $TAG
# drop the raw rocprofv3 trees (hundreds of MB): the summaries stay
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/insts $OUT/all_stats $OUT/all_fetch $OUT/all_write $OUT/all_insts
ls $OUT
