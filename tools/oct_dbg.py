import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import make_frames
from orb_ygz_slam_amd import Extractor
fr = make_frames(8, 752, 480)
ex = Extractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=8)
ex.extract_batch_host(fr); ex.sync()
os.environ["YGZF_OCT_DEBUG"] = "1"
ex.extract_batch_host(fr); ex.sync()
