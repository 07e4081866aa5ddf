"""tools/oct_phases.py -- per-phase clock of k_octree for frame 0 of a 256-frame batch (YGZF_OCT_DEBUG timestamps, 10 ns ticks)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from orb_ygz_slam_amd import Extractor
frames = bench.make_frames(256, 752, 480, seed0=1000)
os.environ["YGZF_OCT_DEBUG"] = "1"      # read once, when the context is created
ex = Extractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=256)
ex.extract_batch_host(frames); ex.sync()
sys.stderr.write("---- second (warm) launch ----\n")
ex.extract_batch_host(frames); ex.sync()
