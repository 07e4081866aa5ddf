"""tools/match_phases.py -- phase times of one SearchByProjection(cur, last) launch (YGZF_DEBUG=match prints them): grid build, candidate scan,
hand-over between the workgroups of a pair, in-order resolution (fixpoint rounds / rescans), commit.  YGZF_FORCE=match_split=n / match_serial=1 select the plans."""
import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from orb_ygz_slam_amd import Extractor, make_camera, EUROC
from orb_ygz_slam_amd.scene import two_view_scene
w,h=752,480
imgA, imgB, (R,t), bp = two_view_scene(9, w, h, EUROC, Z=4.0)
ex = Extractor(1000,1.2,8,20,7,max_width=w,max_height=h,max_batch=1)
ka,da = ex.extract(imgA); kb,db = ex.extract(imgB)
cam = make_camera(w,h)
world = bp(ka["x"], ka["y"])
I=np.eye(3,dtype=np.float32); z=np.zeros(3,np.float32)
for k in range(3):
    r = ex.search_by_projection_last(cam, kb, db, ka, world, da, R.astype(np.float32), t.astype(np.float32), I, z, 15.0, True, True, True)
    print("nmatches", r[0])

# mode 1: SearchByProjection(F, MapPoints) with the projections of the same points (what Tracking::SearchLocalPoints runs per frame)
Rf, tf = R.astype(np.float32), t.astype(np.float32)
pc = world @ Rf.T + tf
px = (np.float32(EUROC["fx"]) * pc[:, 0] / pc[:, 2] + np.float32(EUROC["cx"])).astype(np.float32)
py = (np.float32(EUROC["fy"]) * pc[:, 1] / pc[:, 2] + np.float32(EUROC["cy"])).astype(np.float32)
M = len(ka)
tiv = np.ones(M, np.uint8)
vc = np.full(M, 0.9995, np.float32)
lvl = ka["octave"].astype(np.int32)
for k in range(3):
    r = ex.search_by_projection_mappoints(cam, kb, db, tiv, px, py, vc, lvl, da, 3.0, True, 0.8)
    print("mode 1 nmatches", r[0])
