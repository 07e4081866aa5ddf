#!/bin/bash
# round 5, GPU call 4: after the translation-unit split, the SE3 exp fix and the mgpu reordering -- whole GPU tier + mgpu rates
O=gpurun_out/r05c4
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1
tail -4 $O/tests_gpu.txt
cp gpurun_out/shell_latency.txt gpurun_out/boundary_latency.txt $O/ 2>/dev/null
cat $O/shell_latency.txt $O/boundary_latency.txt
python - <<'PY' > $O/mgpu.txt 2>&1
import json, sys
sys.path.insert(0, '.')
import torch
torch.cuda.init()
import bench
cfg = bench.WORKLOADS["euroc752x480_8lvl_1000feat"]
fr = bench.make_frames(512, 752, 480)
print(json.dumps(bench.mgpu_end_to_end([0], cfg, fr)))
print(json.dumps(bench.mgpu_literal_configs([0])))
PY
tail -3 $O/mgpu.txt | cut -c1-1500
