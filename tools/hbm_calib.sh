#!/bin/bash
# tools/hbm_calib.sh -- run on the GPU box from the repo root: FETCH_SIZE / WRITE_SIZE of tools/micro/hbm_calib's known-size streams (separate
# --pmc passes, --kernel-trace only) -> gpurun_out/hbm_calib/hbm_calibration.json  (copy into profiles/; tools/pmc_summary.py applies it)
REPO=$(pwd)
OUT=$REPO/gpurun_out/hbm_calib
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $REPO/tools/micro/bin/hbm_calib > $OUT/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $REPO/tools/micro/bin/hbm_calib > $OUT/write.log 2>&1
cd $REPO
python tools/hbm_calib_summary.py $OUT
rm -rf $OUT/fetch $OUT/write
cat $OUT/hbm_calibration.json
