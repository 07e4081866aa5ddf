timeout 900 python -m pytest tests/test_gpu_mgpu.py tests/test_gpu_match.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -4
python - <<'PY'
import sys, time, os
sys.path.insert(0,'.')
import numpy as np
import bench
from orb_ygz_slam_amd.synth import synth_frame
frames=np.stack([synth_frame(100+i,752,480) for i in range(64)])
cfg=(752,480,8,1.2,1000,20,7)
for T in (1,2,4,8):
    os.environ['YGZF_MGPU_COPY_THREADS']=str(T)
    r=bench.mgpu_end_to_end([0], cfg, frames, min_seconds=1.5)
    print('copy threads',T, r['value'], r['calls'])
PY
