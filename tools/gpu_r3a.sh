timeout 900 python -m pytest tests/test_gpu_mgpu.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -3
python - <<'PY'
import sys, os
sys.path.insert(0,'.')
import numpy as np
import bench
from orb_ygz_slam_amd.synth import synth_frame
frames=np.stack([synth_frame(100+i,752,480) for i in range(64)])
cfg=(752,480,8,1.2,1000,20,7)
r=bench.mgpu_end_to_end([0], cfg, frames, min_seconds=2.0); print('registered', r['value'], r['calls'])
os.environ['YGZF_MGPU_NO_REGISTER']='1'
r=bench.mgpu_end_to_end([0], cfg, frames, min_seconds=2.0); print('staged', r['value'], r['calls'])
PY
