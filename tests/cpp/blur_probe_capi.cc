// C entry points over host/cv_blur_probe.h for tests/test_cv_blur_probe.py (CPU tier; built there with g++).
#include "../../orb_ygz_slam_amd/csrc/host/cv_blur_probe.h"
extern "C" {
int bp_width() { return ygzf_host::kBlurProbeW; }
int bp_height() { return ygzf_host::kBlurProbeH; }
void bp_image(uint8_t *img) { ygzf_host::blur_probe_image(img); }
void bp_expected(int mode, uint8_t *out) { ygzf_host::blur_probe_expected(mode, out); }
int bp_classify(const uint8_t *blurred) { return ygzf_host::blur_probe_classify(blurred); }
}
