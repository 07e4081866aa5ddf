// oracle_stereo.cpp -- CPU ORACLE (test infrastructure): restatement of ygz::Frame::ComputeStereoMatches,
// reference src/Frame.cc:509-682.  PARITY: PINNED to the reference's own src/Frame.cc (tests/test_ref_frame.py: real Frame.h, compiled where it lies); cv::Mat::convertTo /
// cv::norm(NORM_L1) on the 11x11 patches are exact integer arithmetic in float, restated as such.
#include <algorithm>
#include <climits>
#include <cmath>
#include <utility>
#include <vector>

#include "ygz_oracle.h"

namespace ygzo {

static const int TH_HIGH = 100, TH_LOW = 50;  // src/ORBmatcher.cc:31-32

void compute_stereo_matches(int N, const KeyPoint *keysL, const uint8_t *descL, int Nr, const KeyPoint *keysR, const uint8_t *descR,
                            const std::vector<const Image *> &pyrL, const std::vector<const Image *> &pyrR, const float *mvScaleFactors,
                            const float *mvInvScaleFactors, float mb, float mbf, float *mvuRight, float *mvDepth) {
    for (int i = 0; i < N; i++) mvuRight[i] = mvDepth[i] = -1.0f;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = pyrL[0]->h;
    // :519-538 assign right keypoints to the row table
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    for (int iR = 0; iR < Nr; iR++) {
        const KeyPoint &kp = keysR[iR];
        const float kpY = kp.y;
        const float r = 2.0f * mvScaleFactors[kp.octave];
        const int maxr = (int) std::ceil(kpY + r);
        const int minr = (int) std::floor(kpY - r);
        for (int yi = std::max(minr, 0); yi <= std::min(maxr, nRows - 1); yi++) vRowIndices[yi].push_back(iR);
    }
    const float minZ = mb;
    const float minD = 0;
    const float maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const KeyPoint &kpL = keysL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        if (!(vL >= 0 && vL < (float) nRows)) continue;  // vRowIndices[vL] out of range in the reference
        const std::vector<size_t> &vCandidates = vRowIndices[(size_t) vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD;
        const float maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t *dL = &descL[32 * (size_t) iL];
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const KeyPoint &kpR = keysR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = descriptor_distance(dL, &descR[32 * iR]);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = iR;
                }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = keysR[bestIdxR].x;
            const float scaleFactor = mvInvScaleFactors[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5;
            const Image &imL = *pyrL[kpL.octave], &imR = *pyrR[kpL.octave];
            const int cxL = (int) scaleduL, cyL = (int) scaledvL;
            // rowRange/colRange outside the level throw in OpenCV; left keys sit >= 16 px inside their level
            if (cxL - w < 0 || cyL - w < 0 || cxL + w >= imL.w || cyL + w >= imL.h || cyL + w >= imR.h) continue;
            float IL[11][11];
            for (int r = 0; r < 11; r++)
                for (int c = 0; c < 11; c++) IL[r][c] = (float) imL.d[(size_t) (cyL - w + r) * imL.w + cxL - w + c];
            const float cL = IL[w][w];
            for (int r = 0; r < 11; r++)
                for (int c = 0; c < 11; c++) IL[r][c] = IL[r][c] - cL * 1.0f;
            int bestDistS = INT_MAX;
            int bestincR = 0;
            const int L = 5;
            std::vector<float> vDists(2 * L + 1);
            const float iniu = scaleduR0 + L - w;
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= imR.w) continue;
            const int cxR0 = (int) scaleduR0;
            if (cxR0 - L - w < 0) continue;   // colRange would throw (the reference's own guard misses this side)
            for (int incR = -L; incR <= +L; incR++) {
                const int cxR = cxR0 + incR;
                const float cR = (float) imR.d[(size_t) cyL * imR.w + cxR];
                double acc = 0;   // cv::norm(IL, IR, NORM_L1) accumulates in double
                for (int r = 0; r < 11; r++)
                    for (int c = 0; c < 11; c++) {
                        const float ir = (float) imR.d[(size_t) (cyL - w + r) * imR.w + cxR - w + c] - cR * 1.0f;
                        acc += std::fabs(IL[r][c] - ir);
                    }
                const float dist = (float) acc;
                if (dist < bestDistS) {
                    bestDistS = (int) dist;
                    bestincR = incR;
                }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1];
            const float dist2 = vDists[L + bestincR];
            const float dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = mvScaleFactors[kpL.octave] * ((float) scaleduR0 + (float) bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01;
                    bestuR = uL - 0.01;
                }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
            }
        }
    }
    if (vDistIdx.empty()) return;
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int) vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        else {
            mvuRight[vDistIdx[i].second] = -1;
            mvDepth[vDistIdx[i].second] = -1;
        }
    }
}

}  // namespace ygzo
