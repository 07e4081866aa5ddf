#!/bin/bash
# tools/profile_shapes.sh <tag> [shapes...] -- run on the GPU box from the repo root (gpurun): rocprofv3 kernel-trace stats and the two HBM
# counter passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) of the bench command for the workloads beside the default one:
#   uhd    3840x2160 / 12 levels / 8000 features, stereo pairs (BASELINE config 5 shape)     sub-batches of 64 frames
#   fhd    1920x1080 / 8 levels / 4000 features (config 4 shape)                             sub-batches of 128 frames
#   align  752x480 / 8 / 1000 + SparseImgAlign of every frame against its predecessor (config 3)  sub-batches of 256 frames
# tools/pmc_shapes_summary.py condenses them into <tag>_<shape>_kernel_stats.csv, <tag>_<shape>_pmc_hbm.csv and a traffic.json fragment.
TAG=${1:-r04}
shift
SHAPES=${@:-uhd fhd align}
REPO=$(pwd)
OUT=$REPO/gpurun_out/shapes_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for s in $SHAPES; do
  case $s in
    uhd)   W="--workload uhd3840x2160_12lvl_8000feat --stereo --distinct 8";;
    fhd)   W="--workload fhd1920x1080_8lvl_4000feat --distinct 24";;
    align) W="--align";;
  esac
  BENCH="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $W"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${s}_stats -o stats -- $BENCH > $OUT/${s}_stats.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${s}_fetch -o fetch -- $BENCH > $OUT/${s}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${s}_write -o write -- $BENCH > $OUT/${s}_write.log 2>&1
  tail -1 $OUT/${s}_stats.log | cut -c1-400
done
cd $REPO
python tools/pmc_shapes_summary.py $OUT $TAG $SHAPES
for s in $SHAPES; do rm -rf $OUT/${s}_stats $OUT/${s}_fetch $OUT/${s}_write; done
ls $OUT
