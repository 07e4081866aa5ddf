// FrameStereo.cc -- ygz::Frame::ComputeStereoMatches (src/Frame.cc:509-682) bound to the device form (product code, host side; optional).
// Compiled inside the reference tree against the reference's own, unchanged include/Frame.h, it is the strong definition of that one member:
// the link drops the reference's CPU body (objcopy --weaken on Frame.o, the recipe INTEGRATION.md gives for ORBmatcher.o) and every other
// member of Frame -- constructors, ExtractFeatures with its two extraction threads, which calls this at :742 -- stays the reference's.
// One call: both level-0 images are on the device already when the extractors still hold them; row-band Hamming search, 11 x 11 SAD refinement,
// parabola fit and the median cut run in three launches (stereo_kernels.hip); mvuRight / mvDepth come back in one copy.
#include "ORBextractor.h"   // first: inside the reference tree this is the replacement header (same include guard)
#include "ygz_compat.h"

#include "ygzf_pool.h"

namespace ygz {
void Frame::ComputeStereoMatches() {
    if (!mpORBextractorLeft) {
        mvuRight = std::vector<float>(N, -1.0f);
        mvDepth = std::vector<float>(N, -1.0f);
        ygzf_host::report_failure("ygz::Frame::ComputeStereoMatches", "the frame has no left extractor");
        return;
    }
    mpORBextractorLeft->ComputeStereoMatches(*this);
}
}  // namespace ygz
