"""tools/oct_phases.py [workload] [frames] -- per-phase clock of k_octree for frame 0 of a batch (YGZF_DEBUG=oct timestamps, 10 ns ticks).
workload: a key of bench.WORKLOADS (default euroc752x480_8lvl_1000feat); frames: batch size (default: the bench's sub-batch of the workload).
With YGZF_LIBRARY=orb_ygz_slam_amd/lib_ab/libygzf_clk.so (python -m orb_ygz_slam_amd.build --phase-clock) every tree pass is stamped as well."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from orb_ygz_slam_amd import Extractor
wl = sys.argv[1] if len(sys.argv) > 1 else "euroc752x480_8lvl_1000feat"
w, h, nl, sf, nf, ini, mn = bench.WORKLOADS[wl][:7]
B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.SHAPES[wl][0]
distinct = min(B, 8 if "uhd" in wl else 24 if "fhd" in wl else B)
base = bench.make_frames(distinct, w, h, seed0=1000)
frames = np.concatenate([base] * ((B + distinct - 1) // distinct))[:B]
os.environ["YGZF_DEBUG"] = "oct"     # read once, when the context is created
ex = Extractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=B)
ex.extract_batch_host(frames); ex.sync()
sys.stderr.write("---- second (warm) launch: %s, %d frames ----\n" % (wl, B))
ex.extract_batch_host(frames); ex.sync()
