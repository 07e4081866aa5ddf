#!/bin/bash
# round 5, GPU call 1: copy shapes, upload A/B, stream-partition sweep, new tests
O=gpurun_out/r05c1
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/micro/bin/h2d_shapes > $O/h2d_shapes.txt 2>&1
timeout 60 tools/micro/bin/h2d_shapes 1920 1080 46 > $O/h2d_shapes_fhd.txt 2>&1
python -c "import torch" 2>/dev/null
for k in 0 8 60 1; do
  YGZF_UPLOAD_K=$k timeout 120 python tools/e2e_ab.py euroc752x480_8lvl_1000feat -1 2 >> $O/e2e_ab.txt 2>&1
done
timeout 120 python tools/e2e_ab.py euroc752x480_8lvl_1000feat 0 2 >> $O/e2e_ab.txt 2>&1
timeout 120 python tools/e2e_ab.py euroc752x480_8lvl_1000feat -1 3 >> $O/e2e_ab.txt 2>&1
YGZF_FILL_CUS=64 timeout 120 python tools/e2e_ab.py euroc752x480_8lvl_1000feat -1 2 >> $O/e2e_ab.txt 2>&1
timeout 120 python tools/e2e_ab.py fhd1920x1080_8lvl_4000feat -1 2 >> $O/e2e_ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_partition_upload.py -x -q > $O/tests.txt 2>&1
timeout 600 python tools/partition_sweep.py --streams 3,4 --fills 0,-1,32,64,96,128 --complement 0,1 > $O/sweep.jsonl 2> $O/sweep.txt
tail -3 $O/tests.txt; cat $O/e2e_ab.txt | grep -v Warning | tail -12; cat $O/sweep.txt | tail -30; cat $O/h2d_shapes.txt
