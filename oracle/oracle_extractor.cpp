// oracle_extractor.cpp -- CPU ORACLE (test infrastructure): restatement of ygz::ORBextractor,
// reference src/ORBextractor.cc (line numbers cited per function).  Built with -ffp-contract=off.
// PARITY UNPINNED (see ygz_oracle.h): the reference has no test for this path; this file defines the semantics
// the HIP implementation is graded against.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <list>

#include "orb_pattern_table.h"
#include "ygz_oracle.h"

namespace ygzo {

static const int PATCH_SIZE = 31;       // src/ORBextractor.cc:73
static const int HALF_PATCH_SIZE = 15;  // :74
static const int EDGE_THRESHOLD = 19;   // :75

// src/ORBextractor.cc:412-470
Extractor::Extractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST),
      scaleFactor(_scaleFactor) {
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;
        mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
        mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mvImagePyramid.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / scaleFactor;
    float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float) std::pow((double) factor, (double) nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        mnFeaturesPerLevel[level] = cv_round(nDesiredFeaturesPerScale);
        sumFeatures += mnFeaturesPerLevel[level];
        nDesiredFeaturesPerScale *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

    // orientation: end of each row of the circular patch (:455-469)
    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = (int) std::floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = (int) std::ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// :1131-1132
void Extractor::LevelSize(int w, int h, int level, int *lw, int *lh) const {
    float scale = mvInvScaleFactor[level];
    *lw = cv_round((float) w * scale);
    *lh = cv_round((float) h * scale);
}

// :1129-1150.  The 19-px REFLECT_101 border the reference adds around every level is never read by any stage of
// the hot path, so the oracle keeps tight images (what Frame clones at src/Frame.cc:810-813).
void Extractor::ComputePyramid(const uint8_t *img, int w, int h, int stride) {
    for (int level = 0; level < nlevels; ++level) {
        int lw, lh;
        LevelSize(w, h, level, &lw, &lh);
        mvImagePyramid[level] = Image(lw, lh);
        if (level != 0) {
            resize_linear_u8(mvImagePyramid[level - 1], mvImagePyramid[level]);
        } else {
            for (int y = 0; y < h; y++) std::memcpy(&mvImagePyramid[0].d[(size_t) y * w], img + (size_t) y * stride, w);
        }
    }
}

// :77-101
float Extractor::ICAngle(const Image &image, float ptx, float pty) const {
    int m_01 = 0, m_10 = 0;
    const int cx = cv_round(ptx), cy = cv_round(pty);
    const uint8_t *center = &image.d[(size_t) cy * image.w + cx];
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    int step = image.w;
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2_deg((float) m_01, (float) m_10);
}

// :105-149
void Extractor::ComputeDescriptor(const KeyPoint &kpt, const Image &img, uint8_t *desc) const {
    float a, b;
    sincos_deg(kpt.angle, &a, &b);  // a = cosf(angle*factorPI), b = sinf(...)
    const int cx = cv_round(kpt.x), cy = cv_round(kpt.y);
    const uint8_t *center = &img.d[(size_t) cy * img.w + cx];
    const int step = img.w;
    const int8_t *pattern = kOracleBriefPattern;
#define GET_VALUE(idx)                                                                        \
    center[cv_round((float) pattern[2 * (idx)] * b + (float) pattern[2 * (idx) + 1] * a) * step + \
           cv_round((float) pattern[2 * (idx)] * a - (float) pattern[2 * (idx) + 1] * b)]
    for (int i = 0; i < 32; ++i, pattern += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int t0 = GET_VALUE(2 * j), t1 = GET_VALUE(2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t) val;
    }
#undef GET_VALUE
}

// ---- octree (:479-723) ------------------------------------------------------------------------------------
namespace {
struct P2i { int x, y; };
struct Node {
    std::vector<KeyPoint> vKeys;
    P2i UL, UR, BL, BR;
    std::list<Node>::iterator lit;
    bool bNoMore = false;
    long seq = 0;  // creation sequence number: replaces the heap-pointer tie-break of :653-657
    // :479-531
    void DivideNode(Node &n1, Node &n2, Node &n3, Node &n4) const {
        const int halfX = (int) std::ceil(static_cast<float>(UR.x - UL.x) / 2);
        const int halfY = (int) std::ceil(static_cast<float>(BR.y - UL.y) / 2);
        n1.UL = UL;
        n1.UR = P2i{UL.x + halfX, UL.y};
        n1.BL = P2i{UL.x, UL.y + halfY};
        n1.BR = P2i{UL.x + halfX, UL.y + halfY};
        n2.UL = n1.UR;
        n2.UR = UR;
        n2.BL = n1.BR;
        n2.BR = P2i{UR.x, UL.y + halfY};
        n3.UL = n1.BL;
        n3.UR = n1.BR;
        n3.BL = BL;
        n3.BR = P2i{n1.BR.x, BL.y};
        n4.UL = n3.UR;
        n4.UR = n2.BR;
        n4.BL = n3.BR;
        n4.BR = BR;
        for (size_t i = 0; i < vKeys.size(); i++) {
            const KeyPoint &kp = vKeys[i];
            if (kp.x < n1.UR.x) {
                if (kp.y < n1.BR.y) n1.vKeys.push_back(kp);
                else n3.vKeys.push_back(kp);
            } else if (kp.y < n1.BR.y)
                n2.vKeys.push_back(kp);
            else
                n4.vKeys.push_back(kp);
        }
        if (n1.vKeys.size() == 1) n1.bNoMore = true;
        if (n2.vKeys.size() == 1) n2.bNoMore = true;
        if (n3.vKeys.size() == 1) n3.bNoMore = true;
        if (n4.vKeys.size() == 1) n4.bNoMore = true;
    }
};
}  // namespace

std::vector<KeyPoint> Extractor::DistributeOctTree(const std::vector<KeyPoint> &vToDistributeKeys, int minX, int maxX,
                                                   int minY, int maxY, int N) const {
    // :537  (nIni == 0 for tall-narrow regions is UB in the reference; defined here as 1)
    int nIni = (int) std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni < 1) nIni = 1;
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> lNodes;
    std::vector<Node *> vpIniNodes(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.UL = P2i{(int) (hX * static_cast<float>(i)), 0};
        ni.UR = P2i{(int) (hX * static_cast<float>(i + 1)), 0};
        ni.BL = P2i{ni.UL.x, maxY - minY};
        ni.BR = P2i{ni.UR.x, maxY - minY};
        ni.seq = seq++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
        const KeyPoint &kp = vToDistributeKeys[i];
        size_t idx = (size_t) (kp.x / hX);
        if (idx >= (size_t) nIni) idx = nIni - 1;  // cannot happen for x <= width-4; guards the nIni clamp above
        vpIniNodes[idx]->vKeys.push_back(kp);
    }
    auto lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->vKeys.size() == 1) {
            lit->bNoMore = true;
            lit++;
        } else if (lit->vKeys.empty())
            lit = lNodes.erase(lit);
        else
            lit++;
    }
    bool bFinish = false;
    typedef std::pair<int, long> SizeSeq;  // (size, creation seq) -- see header note
    std::vector<std::pair<SizeSeq, Node *>> vSizeAndPointerToNode;
    auto push_child = [&](Node &n, int *nToExpand) {
        if (n.vKeys.size() > 0) {
            n.seq = seq++;
            lNodes.push_front(n);
            if (n.vKeys.size() > 1) {
                if (nToExpand) (*nToExpand)++;
                vSizeAndPointerToNode.push_back(std::make_pair(SizeSeq((int) n.vKeys.size(), n.seq), &lNodes.front()));
                lNodes.front().lit = lNodes.begin();
            }
        }
    };
    while (!bFinish) {
        int prevSize = (int) lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->bNoMore) {
                lit++;
                continue;
            } else {
                Node n1, n2, n3, n4;
                lit->DivideNode(n1, n2, n3, n4);
                push_child(n1, &nToExpand);
                push_child(n2, &nToExpand);
                push_child(n3, &nToExpand);
                push_child(n4, &nToExpand);
                lit = lNodes.erase(lit);
                continue;
            }
        }
        if ((int) lNodes.size() >= N || (int) lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int) lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int) lNodes.size();
                std::vector<std::pair<SizeSeq, Node *>> vPrev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(vPrev.begin(), vPrev.end(),
                          [](const std::pair<SizeSeq, Node *> &a, const std::pair<SizeSeq, Node *> &b) {
                              return a.first < b.first;
                          });
                for (int j = (int) vPrev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    vPrev[j].second->DivideNode(n1, n2, n3, n4);
                    push_child(n1, nullptr);
                    push_child(n2, nullptr);
                    push_child(n3, nullptr);
                    push_child(n4, nullptr);
                    lNodes.erase(vPrev[j].second->lit);
                    if ((int) lNodes.size() >= N) break;
                }
                if ((int) lNodes.size() >= N || (int) lNodes.size() == prevSize) bFinish = true;
            }
        }
    }
    // :702-720 retain the best point of each node
    std::vector<KeyPoint> vResultKeys;
    vResultKeys.reserve(nfeatures);
    for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
        std::vector<KeyPoint> &vNodeKeys = it->vKeys;
        KeyPoint *pKP = &vNodeKeys[0];
        float maxResponse = pKP->response;
        for (size_t k = 1; k < vNodeKeys.size(); k++) {
            if (vNodeKeys[k].response > maxResponse) {
                pKP = &vNodeKeys[k];
                maxResponse = vNodeKeys[k].response;
            }
        }
        vResultKeys.push_back(*pKP);
    }
    return vResultKeys;
}

// :738-781 the per-cell FAST loop of ComputeKeyPointsOctTree for one level
void Extractor::CellCandidates(int level, std::vector<KeyPoint> &vToDistributeKeys) const {
    vToDistributeKeys.clear();
    const float W = 30;
    const Image &img = mvImagePyramid[level];
    const int minBorderX = EDGE_THRESHOLD - 3;
    const int minBorderY = minBorderX;
    const int maxBorderX = img.w - EDGE_THRESHOLD + 3;
    const int maxBorderY = img.h - EDGE_THRESHOLD + 3;
    const float width = (float) (maxBorderX - minBorderX);
    const float height = (float) (maxBorderY - minBorderY);
    const int nCols = (int) (width / W);
    const int nRows = (int) (height / W);
    if (nCols < 1 || nRows < 1) return;  // reference: division by zero / UB -> defined: no keypoints
    const int wCell = (int) std::ceil(width / nCols);
    const int hCell = (int) std::ceil(height / nRows);
    std::vector<FastPt> cell;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float) (minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float) maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float) (minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float) maxBorderX;
            const int x0 = (int) iniX, y0 = (int) iniY, cw = (int) maxX - x0, ch = (int) maxY - y0;
            const uint8_t *win = &img.d[(size_t) y0 * img.w + x0];
            fast9(win, img.w, cw, ch, iniThFAST, true, cell);
            if (cell.empty()) fast9(win, img.w, cw, ch, minThFAST, true, cell);
            for (const FastPt &p : cell) {
                KeyPoint kp;
                kp.x = (float) p.x + j * wCell;
                kp.y = (float) p.y + i * hCell;
                kp.size = 7.f;
                kp.angle = -1;
                kp.response = (float) p.score;
                kp.octave = 0;
                kp.class_id = -1;
                vToDistributeKeys.push_back(kp);
            }
        }
    }
}

// :725-804
void Extractor::ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>> &allKeypoints) const {
    allKeypoints.assign(nlevels, std::vector<KeyPoint>());
    for (int level = 0; level < nlevels; ++level) {
        const Image &img = mvImagePyramid[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = img.w - EDGE_THRESHOLD + 3, maxBorderY = img.h - EDGE_THRESHOLD + 3;
        std::vector<KeyPoint> vToDistributeKeys;
        CellCandidates(level, vToDistributeKeys);
        std::vector<KeyPoint> &keypoints = allKeypoints[level];
        if (vToDistributeKeys.empty()) continue;
        keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                      mnFeaturesPerLevel[level]);
        const int scaledPatchSize = (int) (PATCH_SIZE * mvScaleFactor[level]);
        for (size_t i = 0; i < keypoints.size(); i++) {
            keypoints[i].x += minBorderX;
            keypoints[i].y += minBorderY;
            keypoints[i].octave = level;
            keypoints[i].size = (float) scaledPatchSize;
        }
    }
    for (int level = 0; level < nlevels; ++level)
        for (KeyPoint &kp : allKeypoints[level]) kp.angle = ICAngle(mvImagePyramid[level], kp.x, kp.y);
}

// :970-1028
void Extractor::Extract(const uint8_t *img, int w, int h, int stride, std::vector<KeyPoint> &_keypoints,
                        std::vector<uint8_t> &descriptors) {
    _keypoints.clear();
    descriptors.clear();
    if (!img || w <= 0 || h <= 0) return;
    ComputePyramid(img, w, h, stride);
    std::vector<std::vector<KeyPoint>> allKeypoints;
    ComputeKeyPointsOctTree(allKeypoints);
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int) allKeypoints[level].size();
    descriptors.assign((size_t) nkeypoints * 32, 0);
    _keypoints.reserve(nkeypoints);
    int offset = 0;
    for (int level = 0; level < nlevels; ++level) {
        std::vector<KeyPoint> &keypoints = allKeypoints[level];
        int nkeypointsLevel = (int) keypoints.size();
        if (nkeypointsLevel == 0) continue;
        Image workingMat;
        gaussian_blur7_s2_u8(mvImagePyramid[level], workingMat);
        for (int i = 0; i < nkeypointsLevel; i++)
            ComputeDescriptor(keypoints[i], workingMat, &descriptors[(size_t) (offset + i) * 32]);
        offset += nkeypointsLevel;
        if (level != 0) {
            float scale = mvScaleFactor[level];
            for (KeyPoint &kp : keypoints) {
                kp.x *= scale;
                kp.y *= scale;
            }
        }
        _keypoints.insert(_keypoints.end(), keypoints.begin(), keypoints.end());
    }
}

}  // namespace ygzo
