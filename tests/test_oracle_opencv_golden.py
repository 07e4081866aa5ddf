"""CPU tier: the oracle's restatements of cv::resize / cv::FAST / cv::GaussianBlur / cv::fastAtan2 against outputs of a REAL OpenCV, when somebody
has provided them (tools/make_golden_opencv.py writes tests/golden/opencv_<version>.npz on any machine with cv2; this environment has none,
so without a fixture the test is skipped and DESIGN.md keeps saying "parity unpinned" for these four primitives)."""
import glob
import os

import numpy as np
import pytest

from tests.conftest import ROOT

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "opencv_*.npz")))


@pytest.mark.skipif(not FIXTURES, reason="no tests/golden/opencv_<version>.npz: run tools/make_golden_opencv.py where cv2 is installed")
@pytest.mark.parametrize("path", FIXTURES)
def test_oracle_primitives_equal_opencv(oracle, path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_golden_opencv import fast_windows, inputs, level_size
    g = np.load(path)
    print("\nOpenCV", str(g["cv_version"]))
    blur_votes = {0: 0, 1: 0, 2: 0}
    for name, img in inputs().items():
        h, w = img.shape
        for sf in (1.2, 1.5, 2.0):
            lw, lh = level_size(w, h, sf)
            assert np.array_equal(oracle.resize(img, lw, lh), g["%s/resize_%.1f" % (name, sf)]), (name, "resize", sf)
        want = g[name + "/blur"]
        for mode in (0, 1, 2):
            with oracle.cv_mode(mode):
                blur_votes[mode] += int(np.array_equal(oracle.blur(img), want))
        for th in (20, 7):
            rows = []
            for k, (x0, y0, x1, y1) in enumerate(fast_windows(h, w)):
                xs, ys, sc = oracle.fast9(np.ascontiguousarray(img[y0:y1, x0:x1]), th, True)
                rows += [(k, int(x), int(y), int(s)) for x, y, s in zip(xs, ys, sc)]
            assert np.array_equal(np.array(rows, np.int32).reshape(-1, 4), g["%s/fast_%d" % (name, th)]), (name, "FAST", th)
    yy, xx = g["atan2/y"], g["atan2/x"]
    got = np.array([[oracle.fast_atan2(float(a), float(b)) for a, b in zip(ry, rx)] for ry, rx in zip(yy, xx)], np.float32)
    assert np.array_equal(got, g["atan2/deg"])
    n = len(inputs())
    print("GaussianBlur: this OpenCV matches ygzf_cv_mode", [m for m, v in blur_votes.items() if v == n], "(0 legacy SSE2, 1 legacy integer, 2 >= 3.4.11 / 4.x)")
    assert max(blur_votes.values()) == n, blur_votes
