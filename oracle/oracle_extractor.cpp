// oracle_extractor.cpp -- CPU ORACLE (test infrastructure): restatement of ygz::ORBextractor,
// reference src/ORBextractor.cc (line numbers cited per function).  Built with -ffp-contract=off.
// PARITY: PINNED to the reference's own src/ORBextractor.cc for everything in this file (tests/test_ref_extractor.py runs that file,
// compiled where it lies against oracle/ref_shim/, next to this one and demands identical keypoints and descriptors); the OpenCV
// primitives it calls (oracle_cvprims.cpp) remain unpinned, and the octree's heap-pointer tie-break is defined as creation order.
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


// ---- FAST-10 grid path (DSO_KEYPOINT) ---------------------------------------------------------------------------------
extern "C" int yo_fast10_detect(const uint8_t *img, int w, int h, int stride, int barrier, short *xy, int cap);  // oracle_fast10.cpp

// :1152-1187
float Extractor::ShiTomasiScore(const Image &img, int u, int v) const {
    float dXX = 0.0, dYY = 0.0, dXY = 0.0;
    const int halfbox_size = 4;
    const int box_size = 2 * halfbox_size;
    const int box_area = box_size * box_size;
    const int x_min = u - halfbox_size, x_max = u + halfbox_size, y_min = v - halfbox_size, y_max = v + halfbox_size;
    if (x_min < 1 || x_max >= img.w - 1 || y_min < 1 || y_max >= img.h - 1) return 0.0;
    const int stride = img.w;
    for (int y = y_min; y < y_max; ++y) {
        const uint8_t *ptr_left = &img.d[(size_t) stride * y + x_min - 1];
        const uint8_t *ptr_right = &img.d[(size_t) stride * y + x_min + 1];
        const uint8_t *ptr_top = &img.d[(size_t) stride * (y - 1) + x_min];
        const uint8_t *ptr_bottom = &img.d[(size_t) stride * (y + 1) + x_min];
        for (int x = 0; x < box_size; ++x, ++ptr_left, ++ptr_right, ++ptr_top, ++ptr_bottom) {
            float dx = (float) (*ptr_right - *ptr_left);
            float dy = (float) (*ptr_bottom - *ptr_top);
            dXX += dx * dx;
            dYY += dy * dy;
            dXY += dx * dy;
        }
    }
    dXX = (float) (dXX / (2.0 * box_area));
    dYY = (float) (dYY / (2.0 * box_area));
    dXY = (float) (dXY / (2.0 * box_area));
    return (float) (0.5 * (dXX + dYY - std::sqrt((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY))));
}

// :1275-1386
void Extractor::ComputeKeyPointsDSOSingleLevel(std::vector<KeyPoint> &allKeypoints, std::vector<KeyPoint> &exist_kps) {
    const Image &img = mvImagePyramid[0];
    std::vector<uint8_t> occ((size_t) img.w * img.h, 0);
    for (KeyPoint &kp : exist_kps) occ[(size_t) cv_round(kp.y) * img.w + cv_round(kp.x)] = 255;
    const int h = img.h, w = img.w, n = nfeatures;
    if (mnGridSize < 0) mnGridSize = static_cast<int>(std::sqrt(1.0 * h * w / n));
    int cnt = 0;
    std::vector<short> xy((size_t) 2 * 64 * 64 + 16);
    while (cnt < n) {
        if (cnt > 0) {
            mnGridSize -= 5;
            if (mnGridSize < 7) {
                mnGridSize = 7;
                break;
            }
        }
        allKeypoints.clear();
        const int grid_n_rows = h / mnGridSize, grid_n_cols = w / mnGridSize;
        if ((size_t) mnGridSize * mnGridSize * 2 > xy.size()) xy.resize((size_t) mnGridSize * mnGridSize * 2);
        cnt = 0;
        for (int k = 0; k < grid_n_rows * grid_n_cols; k++) {
            const int nn = k / grid_n_cols;
            if (nn == 0 || nn == grid_n_rows - 1 || (k % grid_n_cols) == 0 || (k + 1) % grid_n_cols == 0) continue;
            const int x_start = (k - nn * grid_n_cols) * mnGridSize, y_start = nn * mnGridSize;
            const uint8_t *data = &img.d[(size_t) y_start * w + x_start];
            int nc = yo_fast10_detect(data, mnGridSize, mnGridSize, w, 20, xy.data(), mnGridSize * mnGridSize);
            if (nc == 0) nc = yo_fast10_detect(data, mnGridSize, mnGridSize, w, 5, xy.data(), mnGridSize * mnGridSize);  // :1337 hard-coded 5
            if (nc == 0) continue;
            std::vector<std::pair<std::pair<int, int>, float>> corner_score;
            for (int c = 0; c < nc; c++) {
                const int x = xy[2 * c] + x_start, y = xy[2 * c + 1] + y_start;
                if (x < 20 || y < 20 || x >= img.w - 20 || y >= img.h - 20) continue;
                if (occ[(size_t) y * w + x] == 255) continue;
                corner_score.push_back(std::make_pair(std::make_pair(x, y), ShiTomasiScore(img, x, y)));
            }
            std::stable_sort(corner_score.begin(), corner_score.end(),
                             [](const std::pair<std::pair<int, int>, float> &a, const std::pair<std::pair<int, int>, float> &b) {
                                 return a.second > b.second;
                             });
            const int take = corner_score.size() > 3 ? 3 : (int) corner_score.size();
            for (int i = 0; i < take; i++) {
                KeyPoint kp;
                kp.x = (float) corner_score[i].first.first;
                kp.y = (float) corner_score[i].first.second;
                kp.size = 7;
                kp.response = 0;
                kp.octave = 0;
                kp.class_id = -1;
                kp.angle = ICAngle(img, kp.x, kp.y);
                allKeypoints.push_back(kp);
                cnt++;
            }
        }
        if (cnt == 0) break;  // the reference never leaves this loop on a frame without a single corner; defined: no new keys
    }
    if (cnt > n) mnGridSize += 5;
    for (KeyPoint &kp : exist_kps) kp.angle = ICAngle(mvImagePyramid[kp.octave], kp.x * mvInvScaleFactor[kp.octave], kp.y * mvInvScaleFactor[kp.octave]);
}

// :1031-1127 with method == DSO_KEYPOINT, leftEye == true
void Extractor::ExtractDSO(const uint8_t *img, int w, int h, int stride, std::vector<KeyPoint> &keys, std::vector<uint8_t> &desc) {
    ComputePyramid(img, w, h, stride);   // Frame ctor: ComputeImagePyramid; the overload then uses frame->mvImagePyramid
    const int N = (int) keys.size();
    std::vector<KeyPoint> kps;
    ComputeKeyPointsDSOSingleLevel(kps, keys);
    const int nkeypoints = (int) kps.size() + N;
    desc.assign((size_t) nkeypoints * 32, 0);
    std::vector<Image> blurred(nlevels);
    for (int i = 0; i < nlevels; i++) gaussian_blur7_s2_u8(mvImagePyramid[i], blurred[i]);
    for (int i = 0; i < N; i++) {
        KeyPoint tmp = keys[i];
        tmp.x *= mvInvScaleFactor[tmp.octave];
        tmp.y *= mvInvScaleFactor[tmp.octave];
        ComputeDescriptor(tmp, blurred[tmp.octave], &desc[(size_t) i * 32]);
    }
    for (size_t i = 0; i < kps.size(); i++) ComputeDescriptor(kps[i], blurred[0], &desc[(size_t) (N + i) * 32]);
    keys.insert(keys.end(), kps.begin(), kps.end());
}

// ---- FAST_KEYPOINT: ComputeKeyPointsFast :1189-1273 ---------------------------------------------------------------------
// ComputeKeyPointsFast strips a 20-px margin at the top and the left only, so its corners come as close as 3 px to the right and bottom
// borders, where IC_Angle's 15-px disc and the descriptor's rotated pattern (reach 19 px) read outside the level image: in the reference
// those reads land in the next row or past the buffer (undefined; its author's warning at :1191).  DEFINED here: an out-of-image read sees
// BORDER_REFLECT_101 of the image (what the extractor's own bordered pyramid, :1141-1146, holds around every level) -- identical to the
// reference wherever the reference is defined (patch inside the image).
static inline int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}
static inline int px_reflect(const Image &im, int x, int y) { return im.d[(size_t) reflect101(y, im.h) * im.w + reflect101(x, im.w)]; }
static float ic_angle_reflect(const Extractor &E, const Image &image, float ptx, float pty) {
    int m_01 = 0, m_10 = 0;
    const int cx = cv_round(ptx), cy = cv_round(pty);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * px_reflect(image, cx + u, cy);
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        const int d = E.umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = px_reflect(image, cx + u, cy + v), val_minus = px_reflect(image, cx + u, cy - v);
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2_deg((float) m_01, (float) m_10);
}
static void descriptor_reflect(const KeyPoint &kpt, const Image &img, uint8_t *desc) {
    float a, b;
    sincos_deg(kpt.angle, &a, &b);
    const int cx = cv_round(kpt.x), cy = cv_round(kpt.y);
    const int8_t *pattern = kOracleBriefPattern;
#define GET_VALUE(idx)                                                                                             \
    px_reflect(img, cx + cv_round((float) pattern[2 * (idx)] * a - (float) pattern[2 * (idx) + 1] * b),            \
               cy + cv_round((float) pattern[2 * (idx)] * b + (float) pattern[2 * (idx) + 1] * a))
    for (int i = 0; i < 32; ++i, pattern += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            const int t0 = GET_VALUE(2 * j), t1 = GET_VALUE(2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t) val;
    }
#undef GET_VALUE
}

extern "C" void yo_fast10_score(const uint8_t *img, int stride, const short *xy, int n, int *scores);            // oracle_fast10.cpp
extern "C" int yo_fast_nonmax_3x3(const short *xy, const int *scores, int n, int *out_idx, int cap);

void Extractor::ComputeKeyPointsFast(std::vector<std::vector<KeyPoint>> &allKeypoints, std::vector<KeyPoint> &exist_kps) const {
    const int mnCellSize = 5;
    const int mnGridRows = mvImagePyramid[0].h / mnCellSize, mnGridCols = mvImagePyramid[0].w / mnCellSize;   // ceil() of an integer quotient (:1194-1195)
    const long long nCells = (long long) mnGridCols * mnGridRows;
    std::vector<uint8_t> occ((size_t) nCells, 0);
    for (const KeyPoint &kp : exist_kps) {
        const int gy = static_cast<int>((kp.y) / mnCellSize), gx = static_cast<int>((kp.x) / mnCellSize);
        const long long k = (long long) gy * mnGridCols + gx;
        if (k >= 0 && k < nCells) occ[(size_t) k] = 1;   // (the reference asserts / writes out of bounds otherwise)
    }
    std::vector<KeyPoint> gridFeatures((size_t) nCells);
    for (KeyPoint &g : gridFeatures) { g.x = g.y = 0; g.size = 0; g.angle = -1; g.response = 0; g.octave = 0; g.class_id = -1; }
    allKeypoints.assign(nlevels, std::vector<KeyPoint>());
    for (int level = 0; level < nlevels; level++) {
        const Image &img = mvImagePyramid[level];
        const int scaledPatchSize = (int) (31 * mvScaleFactor[level]);
        const float scale = mvScaleFactor[level];
        const int boarder = 20;
        if (img.w < 42 || img.h < 27) continue;   // DEFINED (header)
        const uint8_t *data_start = &img.d[(size_t) boarder * img.w + boarder];
        const int ww = img.w - boarder, wh = img.h - boarder;
        std::vector<short> xy((size_t) 2 * ww * wh);
        const int nc = yo_fast10_detect(data_start, ww, wh, img.w, iniThFAST, xy.data(), ww * wh);
        std::vector<int> scores(std::max(nc, 1)), nm(std::max(nc, 1));
        yo_fast10_score(data_start, img.w, xy.data(), nc, scores.data());
        const int nn = yo_fast_nonmax_3x3(xy.data(), scores.data(), nc, nm.data(), nc);
        for (int i = 0; i < nn; i++) {
            const short x = (short) (xy[2 * nm[i]] + boarder), y = (short) (xy[2 * nm[i] + 1] + boarder);
            const int gy = static_cast<int>((y * scale) / mnCellSize), gx = static_cast<int>((x * scale) / mnCellSize);
            const long long k = (long long) gy * mnGridCols + gx;
            if (k < 0 || k >= nCells) continue;   // DEFINED (header)
            if (occ[(size_t) k]) continue;
            const float s = ShiTomasiScore(img, x, y);
            if (s > gridFeatures[(size_t) k].response) {
                KeyPoint kp;
                kp.x = x; kp.y = y; kp.size = (float) scaledPatchSize; kp.angle = -1; kp.response = s; kp.octave = level; kp.class_id = -1;
                gridFeatures[(size_t) k] = kp;
            }
        }
    }
    for (KeyPoint &kp : gridFeatures)
        if (kp.size > 0) {
            kp.angle = ic_angle_reflect(*this, mvImagePyramid[kp.octave], kp.x, kp.y);
            allKeypoints[kp.octave].push_back(kp);
        }
    for (KeyPoint &kp : exist_kps)
        kp.angle = ic_angle_reflect(*this, mvImagePyramid[kp.octave], kp.x * mvInvScaleFactor[kp.octave], kp.y * mvInvScaleFactor[kp.octave]);
}

// the part of the Frame overload after the detector (:1063-1126): descriptors of the N existing keys at pt * invScale[octave] first, then
// level after level the new keys (descriptor on the blurred level, pt *= scale for levels > 0)
static void describe_and_concat(const Extractor &E, std::vector<std::vector<KeyPoint>> &all, std::vector<KeyPoint> &keys, std::vector<uint8_t> &desc) {
    const int N = (int) keys.size();
    size_t nk = (size_t) N;
    for (auto &v : all) nk += v.size();
    desc.assign(nk * 32, 0);
    std::vector<Image> blurred(E.nlevels);
    for (int i = 0; i < E.nlevels; i++) gaussian_blur7_s2_u8(E.mvImagePyramid[i], blurred[i]);
    for (int i = 0; i < N; i++) {
        KeyPoint tmp = keys[i];
        tmp.x *= E.mvInvScaleFactor[tmp.octave];
        tmp.y *= E.mvInvScaleFactor[tmp.octave];
        descriptor_reflect(tmp, blurred[tmp.octave], &desc[(size_t) i * 32]);
    }
    size_t off = (size_t) N;
    for (int level = 0; level < E.nlevels; level++) {
        for (KeyPoint &kp : all[level]) {
            descriptor_reflect(kp, blurred[level], &desc[off * 32]);
            off++;
        }
        if (level != 0)
            for (KeyPoint &kp : all[level]) { kp.x *= E.mvScaleFactor[level]; kp.y *= E.mvScaleFactor[level]; }
        keys.insert(keys.end(), all[level].begin(), all[level].end());
    }
}

void Extractor::ExtractFast(const uint8_t *img, int w, int h, int stride, std::vector<KeyPoint> &keys, std::vector<uint8_t> &desc) {
    ComputePyramid(img, w, h, stride);
    std::vector<std::vector<KeyPoint>> all;
    ComputeKeyPointsFast(all, keys);
    describe_and_concat(*this, all, keys, desc);
}

// ---- multi-level DSO grid detector: ComputeKeyPointsDSO :1388-1507 -----------------------------------------------------------
void Extractor::ComputeKeyPointsDSO(std::vector<std::vector<KeyPoint>> &allKeypoints, std::vector<KeyPoint> &exist_kps) {
    const int w0 = mvImagePyramid[0].w, h0 = mvImagePyramid[0].h;
    std::vector<uint8_t> occ((size_t) w0 * h0, 0);
    for (KeyPoint &kp : exist_kps) {
        const int ox = cv_round(kp.x), oy = cv_round(kp.y);
        if (ox >= 0 && oy >= 0 && ox < w0 && oy < h0) occ[(size_t) oy * w0 + ox] = 255;
    }
    allKeypoints.assign(nlevels, std::vector<KeyPoint>());
    std::vector<short> xy((size_t) 2 * 64 * 64 + 16);
    for (int level = 0; level < nlevels; level++) {
        const Image &img = mvImagePyramid[level];
        const int h = img.h, w = img.w, n = mnFeaturesPerLevel[level];
        if (n <= 0) continue;   // (sqrt of a division by zero in the reference)
        mnGridSize = static_cast<int>(std::sqrt(1.0 * h * w / n));
        if (mnGridSize < 1) mnGridSize = 1;
        int cnt = 0;
        while (cnt < n) {
            if (cnt > 0) {
                mnGridSize -= 5;
                if (mnGridSize < 7) {
                    mnGridSize = 7;
                    break;
                }
            }
            allKeypoints[level].clear();
            const int grid_n_rows = h / mnGridSize, grid_n_cols = w / mnGridSize;
            if ((size_t) mnGridSize * mnGridSize * 2 > xy.size()) xy.resize((size_t) mnGridSize * mnGridSize * 2);
            cnt = 0;
            for (int k = 0; k < grid_n_rows * grid_n_cols; k++) {
                const int nn = k / grid_n_cols;
                if (nn == 0 || nn == grid_n_rows - 1 || (k % grid_n_cols) == 0 || (k + 1) % grid_n_cols == 0) continue;
                const int x_start = (k - nn * grid_n_cols) * mnGridSize, y_start = nn * mnGridSize;
                const uint8_t *data = &img.d[(size_t) y_start * w + x_start];
                int nc = yo_fast10_detect(data, mnGridSize, mnGridSize, w, iniThFAST, xy.data(), mnGridSize * mnGridSize);
                if (nc == 0) nc = yo_fast10_detect(data, mnGridSize, mnGridSize, w, minThFAST, xy.data(), mnGridSize * mnGridSize);
                if (nc == 0) continue;
                std::vector<std::pair<std::pair<int, int>, float>> corner_score;
                for (int c = 0; c < nc; c++) {
                    const int x = xy[2 * c] + x_start, y = xy[2 * c + 1] + y_start;
                    if (x < 20 || y < 20 || x >= w - 20 || y >= h - 20) continue;
                    if (occ[(size_t) y * w0 + x] == 255) continue;   // level coordinates on the level-0 map (:1466)
                    corner_score.push_back(std::make_pair(std::make_pair(x, y), ShiTomasiScore(img, x, y)));
                }
                std::stable_sort(corner_score.begin(), corner_score.end(),
                                 [](const std::pair<std::pair<int, int>, float> &a, const std::pair<std::pair<int, int>, float> &b) { return a.second > b.second; });
                const int take = corner_score.size() > 2 ? 2 : (int) corner_score.size();
                for (int i = 0; i < take; i++) {
                    KeyPoint kp;
                    kp.x = (float) corner_score[i].first.first;
                    kp.y = (float) corner_score[i].first.second;
                    kp.size = 7; kp.response = 0; kp.octave = level; kp.class_id = -1;
                    kp.angle = ICAngle(img, kp.x, kp.y);
                    allKeypoints[level].push_back(kp);
                    occ[(size_t) corner_score[i].first.second * w0 + corner_score[i].first.first] = 255;
                    cnt++;
                }
            }
            if (cnt == 0) break;   // DEFINED (header)
        }
    }
    for (KeyPoint &kp : exist_kps) kp.angle = ICAngle(mvImagePyramid[kp.octave], kp.x * mvInvScaleFactor[kp.octave], kp.y * mvInvScaleFactor[kp.octave]);
}

void Extractor::ExtractDSOMultiLevel(const uint8_t *img, int w, int h, int stride, std::vector<KeyPoint> &keys, std::vector<uint8_t> &desc) {
    ComputePyramid(img, w, h, stride);
    std::vector<std::vector<KeyPoint>> all;
    ComputeKeyPointsDSO(all, keys);
    describe_and_concat(*this, all, keys, desc);
}

}  // namespace ygzo
