"""Host frames laid out at ygzf_host_row_pitch (whole-frame uploads) give the bytes of tight frames, through ygzf_extract_batch_host,
ygzf_extract_batch_host_frames and ygzf_mgpu_* (page-locked and pageable).  (Round 5's CU-masked stream partition, which this file also covered, was
measured slower in every setting -- profiles/r05_a_partition_sweep.txt -- and left the library in round 6.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _clip(n, w, h, seed=900):
    frames = np.empty((n, h, w), np.uint8)
    for i in range(n):
        if i % 4 == 0:
            scene = synth_frame(seed + i // 4, w + 16, h + 16)
        frames[i] = scene[2 * (i % 4):2 * (i % 4) + h, 3 * (i % 4):3 * (i % 4) + w]
    return frames


def _run(ex, frames, cam):
    ex.extract_batch_host(frames)
    ex.match_batch_prev(cam, 15.0, True, True, True)
    n = len(frames)
    out = []
    for f in range(n):
        k, d = ex.batch_fetch(f)
        m, o = ex.match_fetch(f)
        out.append((k.copy(), d.copy(), m.copy(), o.copy()))
    return out, ex.match_counts().copy()


def _same(a, b):
    # (frame 0 is matched against whatever the context's previous batch left behind: its keypoints / descriptors are compared, its matches are not)
    (ra, ca), (rb, cb) = a, b
    assert (np.asarray(ca)[1:] == np.asarray(cb)[1:]).all()
    for f, (x, y) in enumerate(zip(ra, rb)):
        for u, v in list(zip(x, y))[:2 if f == 0 else 4]:
            assert np.array_equal(u, v)


@pytest.mark.parametrize("w,h", [(752, 480), (500, 376), (640, 480), (322, 250)])
def test_pitched_host_frames(w, h):
    """(a process of its own: torch initialises the device first, as bench.py does)"""
    code = r"""
import numpy as np, sys
import torch
torch.cuda.init()            # before the library creates its own HIP context (as bench.py does)
sys.path.insert(0, %r)
from orb_ygz_slam_amd import Extractor, MultiGpu, make_camera
from orb_ygz_slam_amd.capi import host_row_pitch
from tests.test_gpu_partition_upload import _clip, _run, _same
w, h, n = %d, %d, 10
frames = _clip(n, w, h)
cam = make_camera(w, h)
P = host_row_pitch(w)
assert P %% 64 == 0 and P >= w
ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n, device=0)
ref = _run(ex, frames, cam)
wide = np.full((n, h + 3, P), 0xAB, np.uint8)            # rows at the device pitch, frames three rows apart: padding full of garbage
wide[:, :h, :w] = frames
_same(ref, _run(ex, wide[:, :h, :w], cam))
back = np.full((n, h, P), 0xCD, np.uint8)                # frames back to back at the pitch
back[:, :, :w] = frames
_same(ref, _run(ex, back[:, :, :w], cam))
ex.extract_batch_host(back[:1, :, :w]); k1, d1 = ex.batch_fetch(0)
assert np.array_equal(k1, ref[0][0][0]) and np.array_equal(d1, ref[0][0][1])
# frame-pointer lists: pitched frames, runs of 2 at a distance (a slot's round-robin share)
views = [back[f, :, :w] for f in range(n)]
ex.extract_batch_host_frames(views); ex.match_batch_prev(cam, 15.0, True, True, True)
got = [(ex.batch_fetch(f) + ex.match_fetch(f)) for f in range(n)]
_same(ref, (got, ex.match_counts()))
share = [back[f, :, :w] for f in range(n) if (f // 2) %% 2 == 0]
ex.extract_batch_host_frames(share)
for j, f in enumerate([f for f in range(n) if (f // 2) %% 2 == 0]):
    kk, dd = ex.batch_fetch(j)
    assert np.array_equal(kk, ref[0][f][0]) and np.array_equal(dd, ref[0][f][1])
ex.close()
# the multi-GPU entry point: pageable tight / pageable pitched / page-locked pitched
one = MultiGpu([0], max_width=w, max_height=h, max_frames_per_device=n)
r0 = one.extract_match(frames, unit=2, cam=cam)
one.close()
pin = torch.zeros((n, h, P), dtype=torch.uint8).pin_memory(); pin.numpy()[:, :, :w] = frames
for slots in ([0, 0], [0, 0, 0]):
    mg = MultiGpu(slots, max_width=w, max_height=h, max_frames_per_device=n)
    for src in (frames, back[:, :, :w], pin.numpy()[:, :, :w]):
        g = mg.extract_match(src, unit=2, cam=cam)
        for a, b in zip(r0, g):
            assert np.array_equal(a, b)
    mg.close()
print("OK")
""" % (ROOT, w, h)
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
