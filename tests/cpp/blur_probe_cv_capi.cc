// tests/test_cv_blur_probe.py: the probe's cv:: adapter (host/cv_blur_probe.h, blur_probe_run_opencv) compiled over the oracle-backed OpenCV stand-in
// (oracle/ref_shim/mini_cv: its GaussianBlur is the oracle's, in the generation yo_set_cv_mode selects) -- the code path ORBextractor.cc runs against
// a real OpenCV, end to end minus the real OpenCV.
#include "mini_cv.h"
#define YGZF_BLUR_PROBE_WITH_CV 1
#include "../../orb_ygz_slam_amd/csrc/host/cv_blur_probe.h"
extern "C" int bp_run_opencv() { return ygzf_host::blur_probe_run_opencv(); }
