#!/usr/bin/env python3
"""tools/make_golden_opencv.py -- for anyone who HAS OpenCV: pin the four OpenCV primitives the oracle had to restate from memory.

The extractor's results rest on cv::resize (INTER_LINEAR), cv::FAST (9/16, non-max suppression), cv::GaussianBlur (7x7, sigma 2, REFLECT_101)
and cv::fastAtan2 (reference src/ORBextractor.cc:1139, :765-769, :1010 / :1083, :100).  OpenCV is neither vendored by the reference nor installed
in this project's build environment, so oracle/oracle_cvprims.cpp restates them and says "parity unpinned" for exactly
these.  Run this script once on any machine with `cv2`:

    python tools/make_golden_opencv.py            # writes tests/golden/opencv_<cv2.__version__>.npz

and commit the file: tests/test_oracle_opencv_golden.py (CPU tier) then checks the oracle against the real library -- resize, FAST and fastAtan2
must be identical; for GaussianBlur it reports which of the oracle's three definitions (ygzf_cv_mode: legacy SSE2 / legacy integer / OpenCV >=
3.4.11) this OpenCV build implements and requires one of them to match.  Inputs: the seeded synthetic frames 0..7 of the test suite and the
reference's own Thirdparty/fast/test/data/test1.png (pixels from tests/golden/fast10_test1.npz).  Without cv2 the script says so and exits 2.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inputs():
    from orb_ygz_slam_amd.synth import synth_frame
    imgs = {"seed%d" % s: synth_frame(s, 640, 480) for s in range(8)}
    imgs["test1png"] = np.load(os.path.join(ROOT, "tests", "golden", "fast10_test1.npz"))["image"]
    return imgs


def level_size(w, h, sf):      # cvRound((float) cols * inverse scale), src/ORBextractor.cc:1131-1132
    inv = np.float32(1.0) / np.float32(sf)
    return int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))


def fast_windows(h, w):
    """the extractor's own 30-px cells of level 0 (every third one): window = cell + 3-px apron, src/ORBextractor.cc:733-764"""
    minB, maxBX, maxBY = 16, w - 16, h - 16
    nCols, nRows = (maxBX - minB) // 30, (maxBY - minB) // 30
    wC, hC = -(-(maxBX - minB) // nCols), -(-(maxBY - minB) // nRows)
    out = []
    for i in range(nRows):
        for j in range(nCols):
            if (i * nCols + j) % 3:
                continue
            x0, y0 = minB + j * wC, minB + i * hC
            x1, y1 = min(x0 + wC + 6, maxBX), min(y0 + hC + 6, maxBY)
            if x0 >= maxBX - 6 or y0 >= maxBY - 3:
                continue
            out.append((x0, y0, x1, y1))
    return out


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 is not importable here: run this on a machine with OpenCV (any 2.4 / 3.x / 4.x Python build)")
        return 2
    out = {"cv_version": np.array(cv2.__version__)}
    for name, img in inputs().items():
        h, w = img.shape
        for sf in (1.2, 1.5, 2.0):
            lw, lh = level_size(w, h, sf)
            out["%s/resize_%.1f" % (name, sf)] = cv2.resize(img, (lw, lh), interpolation=cv2.INTER_LINEAR)
        out[name + "/blur"] = cv2.GaussianBlur(img, (7, 7), 2, None, 2, cv2.BORDER_REFLECT_101)   # (src, ksize, sigmaX, dst, sigmaY, borderType)
        for th in (20, 7):
            try:
                det = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
            except AttributeError:                      # OpenCV 2.4
                det = cv2.FastFeatureDetector(th, True)
            rows = []
            for k, (x0, y0, x1, y1) in enumerate(fast_windows(h, w)):
                for kp in det.detect(np.ascontiguousarray(img[y0:y1, x0:x1]), None):
                    rows.append((k, int(round(kp.pt[0])), int(round(kp.pt[1])), int(round(kp.response))))
            out["%s/fast_%d" % (name, th)] = np.array(rows, np.int32).reshape(-1, 4)   # (window, x, y, score) in detection order
    g = np.arange(-40, 41, dtype=np.float32) * np.float32(37.25)
    yy, xx = np.meshgrid(g, g, indexing="ij")
    out["atan2/y"], out["atan2/x"] = yy, xx
    out["atan2/deg"] = np.array([[cv2.fastAtan2(float(a), float(b)) for a, b in zip(ry, rx)] for ry, rx in zip(yy, xx)], np.float32)
    path = os.path.join(ROOT, "tests", "golden", "opencv_%s.npz" % cv2.__version__)
    np.savez_compressed(path, **out)
    print("wrote", path, "(%d arrays); now run: python -m pytest tests/test_oracle_opencv_golden.py -q -s" % len(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
