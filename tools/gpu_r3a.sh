YGZF_REPEATS=400 timeout 1500 python -m pytest tests/test_gpu_repeat.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -3
python - <<'PY'
# many single-pair matcher launches in split mode against one reference result: the hand-over (fence + counter) under repetition
import sys, numpy as np
sys.path.insert(0,'.')
from orb_ygz_slam_amd import Extractor, make_camera, EUROC
from orb_ygz_slam_amd.scene import two_view_scene
w,h=752,480
A,B,(R,t),bp=two_view_scene(9,w,h,EUROC)
ex=Extractor(1000,1.2,8,20,7,max_width=w,max_height=h,max_batch=1)
cam=make_camera(w,h)
ka,da=ex.extract(A); kb,db=ex.extract(B)
world=bp(ka["x"],ka["y"])
I3,z3=np.eye(3,dtype=np.float32),np.zeros(3,np.float32)
ref=None; bad=0
for i in range(20000):
    r=ex.search_by_projection_last(cam,kb,db,ka,world,da,I3,z3,I3,z3,15.0)
    key=(int(r[0]), r[1].tobytes(), r[2].tobytes())
    if ref is None: ref=key
    elif key!=ref: bad+=1
print('split matcher repeats 20000, mismatches', bad, 'matches', ref[0])
PY
