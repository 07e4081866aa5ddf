// oracle_matcher.cpp -- CPU ORACLE (test infrastructure): restatement of the Tracking-called parts of
// ygz::ORBmatcher (reference src/ORBmatcher.cc) and of the Frame feature grid (src/Frame.cc).
// MapPoint / Frame objects are replaced by the plain arrays the functions actually read (see ygz_oracle.h).
// Built with -ffp-contract=off: float expressions are evaluated in source order without FMA.
// PARITY: the five search functions, DescriptorDistance and ComputeThreeMaxima are PINNED to the reference's own src/ORBmatcher.cc
// (tests/test_ref_matcher.py runs that file, compiled where it lies over oracle/ref_shim/, on identical inputs); PredictScale and
// ComputeDistinctiveDescriptors to its src/MapPoint.cc (tests/test_ref_mappoint.py); the Frame grid and isInFrustum to its src/Frame.cc
// (tests/test_ref_frame.py).
#include <climits>
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ygz_oracle.h"

namespace ygzo {

static const int TH_HIGH = 100;      // src/ORBmatcher.cc:36
static const int TH_LOW = 50;        // :37
static const int HISTO_LENGTH = 30;  // :38

// :1507-1523 -- popcount of the xor over 8 x 32-bit words
int descriptor_distance(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// src/Frame.cc:314-330 (AssignFeaturesToGrid) + :483-493 (PosInGrid: note round(), not floor())
void Grid::Assign(const FrameView &f) {
    for (int i = 0; i < COLS; i++)
        for (int j = 0; j < ROWS; j++) cell[i][j].clear();
    for (int i = 0; i < f.N; i++) {
        const KeyPoint &kp = f.keys[i];
        int posX = (int) std::round((kp.x - f.minX) * f.gridInvW);
        int posY = (int) std::round((kp.y - f.minY) * f.gridInvH);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        cell[posX][posY].push_back(i);
    }
}

// src/Frame.cc:424-481
void Grid::FeaturesInArea(const FrameView &f, float x, float y, float r, int minLevel, int maxLevel,
                          std::vector<int> &vIndices) const {
    vIndices.clear();
    const int nMinCellX = std::max(0, (int) std::floor((x - f.minX - r) * f.gridInvW));
    if (nMinCellX >= COLS) return;
    const int nMaxCellX = std::min((int) COLS - 1, (int) std::ceil((x - f.minX + r) * f.gridInvW));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int) std::floor((y - f.minY - r) * f.gridInvH));
    if (nMinCellY >= ROWS) return;
    const int nMaxCellY = std::min((int) ROWS - 1, (int) std::ceil((y - f.minY + r) * f.gridInvH));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<int> &vCell = cell[ix][iy];
            for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
                const KeyPoint &kpUn = f.keys[vCell[j]];
                if (bCheckLevels) {
                    if (kpUn.octave < minLevel) continue;
                    if (maxLevel >= 0)
                        if (kpUn.octave > maxLevel) continue;
                }
                const float distx = kpUn.x - x;
                const float disty = kpUn.y - y;
                if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
            }
        }
    }
}

// :1471-1502
static void compute_three_maxima(const std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int) histo[i].size();
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float) max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float) max1) {
        ind3 = -1;
    }
}

static inline void mat3_mul_vec(const float R[9], const float v[3], float o[3]) {
    // Eigen 3x3 * 3x1 product, coefficient order: o_i = R_i0*v0 + R_i1*v1 + R_i2*v2
    for (int i = 0; i < 3; i++) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}

// :1218-1350
int search_by_projection_last(const FrameView &cur, const Grid &grid, const ProjLastInput &last, float th, bool bMono,
                              bool checkLevel, bool checkOrientation, uint8_t *cur_owner, int *cur_match) {
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    const float *Rcw = last.Rcw, *tcw = last.tcw, *Rlw = last.Rlw, *tlw = last.tlw;
    // twc = -1 * Rcw^T * tcw ; tlc = Rlw * twc + tlw
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) twc[i] = -1 * (Rcw[i] * tcw[0] + Rcw[3 + i] * tcw[1] + Rcw[6 + i] * tcw[2]);
    mat3_mul_vec(Rlw, twc, tlc);
    for (int i = 0; i < 3; i++) tlc[i] = tlc[i] + tlw[i];
    const bool bForward = tlc[2] > cur.mb && !bMono;
    const bool bBackward = -tlc[2] > cur.mb && !bMono;
    std::vector<int> vIndices2;
    for (int i = 0; i < last.N; i++) {
        if (!last.mp_valid[i]) continue;
        if (last.outlier[i]) continue;
        float x3Dc[3];
        mat3_mul_vec(Rcw, &last.mp_world[3 * i], x3Dc);
        for (int k = 0; k < 3; k++) x3Dc[k] = x3Dc[k] + tcw[k];
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float) (1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        float u = cur.fx * xc * invzc + cur.cx;
        float v = cur.fy * yc * invzc + cur.cy;
        if (u < cur.minX || u > cur.maxX) continue;
        if (v < cur.minY || v > cur.maxY) continue;
        int nLastOctave = last.keys[i].octave;
        float radius = th * cur.scaleFactors[nLastOctave];
        if (checkLevel == false) grid.FeaturesInArea(cur, u, v, radius, -1, -1, vIndices2);
        else if (bForward) grid.FeaturesInArea(cur, u, v, radius, nLastOctave, -1, vIndices2);
        else if (bBackward) grid.FeaturesInArea(cur, u, v, radius, 0, nLastOctave, vIndices2);
        else grid.FeaturesInArea(cur, u, v, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = &last.mp_desc[32 * (size_t) i];
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (cur_owner[i2] == 2) continue;  // mvpMapPoints[i2] && Observations() > 0
            if (cur.uRight && cur.uRight[i2] > 0) {
                const float ur = u - cur.mbf * invzc;
                const float er = std::fabs(ur - cur.uRight[i2]);
                if (er > radius) continue;
            }
            const int dist = descriptor_distance(dMP, &cur.desc[32 * (size_t) i2]);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= TH_HIGH) {
            cur_owner[bestIdx2] = last.mp_has_obs[i] ? 2 : 1;
            cur_match[bestIdx2] = i;
            nmatches++;
            if (checkOrientation) {
                float rot = last.keys[i].angle - cur.keys[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int) std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    cur_owner[rotHist[i][j]] = 0;
                    cur_match[rotHist[i][j]] = -2;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// :43-126
int search_by_projection_mappoints(const FrameView &F, const Grid &grid, const ProjMapPointsInput &in, float th,
                                   bool checkLevel, float nnratio, uint8_t *owner, int *match) {
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < in.M; iMP++) {
        if (!in.track_in_view[iMP]) continue;
        if (in.bad[iMP]) continue;
        const int nPredictedLevel = in.scaleLevel[iMP];
        float r = in.viewCos[iMP] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos :128-133
        if (bFactor) r *= th;
        if (checkLevel)
            grid.FeaturesInArea(F, in.projX[iMP], in.projY[iMP], r * F.scaleFactors[nPredictedLevel], nPredictedLevel - 1,
                                nPredictedLevel, vIndices);
        else
            grid.FeaturesInArea(F, in.projX[iMP], in.projY[iMP], r * F.scaleFactors[nPredictedLevel], -1, -1, vIndices);
        if (vIndices.empty()) continue;
        const uint8_t *MPdescriptor = &in.mp_desc[32 * (size_t) iMP];
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (owner[idx] == 2) continue;
            if (F.uRight && F.uRight[idx] > 0) {
                const float er = std::fabs(in.projXR[iMP] - F.uRight[idx]);
                if (er > r * F.scaleFactors[nPredictedLevel]) continue;
            }
            const int dist = descriptor_distance(MPdescriptor, &F.desc[32 * (size_t) idx]);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = F.keys[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F.keys[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            owner[bestIdx] = in.mp_has_obs[iMP] ? 2 : 1;
            match[bestIdx] = iMP;
            nmatches++;
        }
    }
    return nmatches;
}

// :1352-1469
int search_by_projection_kf(const FrameView &cur, const Grid &grid, const ProjKFInput &in, float th, int ORBdist,
                            bool checkOrientation, uint8_t *cur_owner, int *cur_match, uint8_t *out_valid, float *out_u,
                            float *out_v, int *out_level) {
    int nmatches = 0;
    const float *Rcw = in.Rcw, *tcw = in.tcw;
    float Ow[3];  // -1 * Rcw^T * tcw
    for (int i = 0; i < 3; i++) Ow[i] = -1 * (Rcw[i] * tcw[0] + Rcw[3 + i] * tcw[1] + Rcw[6 + i] * tcw[2]);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vIndices2;
    for (int i = 0; i < in.M; i++) {
        if (out_valid) out_valid[i] = 0;
        if (!in.usable[i]) continue;
        const float *x3Dw = &in.world[3 * i];
        float x3Dc[3];
        mat3_mul_vec(Rcw, x3Dw, x3Dc);
        for (int k = 0; k < 3; k++) x3Dc[k] = x3Dc[k] + tcw[k];
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float) (1.0 / x3Dc[2]);
        const float u = cur.fx * xc * invzc + cur.cx;
        const float v = cur.fy * yc * invzc + cur.cy;
        if (u < cur.minX || u > cur.maxX) continue;
        if (v < cur.minY || v > cur.maxY) continue;
        const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
        const float dist3D = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);  // Eigen norm(): sqrt of the in-order sum
        if (dist3D < in.minDistInv[i] || dist3D > in.maxDistInv[i]) continue;
        // MapPoint::PredictScale(dist3D, &CurrentFrame)  src/MapPoint.cc:359-373 (std::log / std::ceil float overloads)
        const float ratio = in.mfMaxDistance[i] / dist3D;
        int nPredictedLevel = (int) std::ceil(std::log(ratio) / in.logScaleFactor);
        if (nPredictedLevel < 0) nPredictedLevel = 0;
        else if (nPredictedLevel >= in.nScaleLevels) nPredictedLevel = in.nScaleLevels - 1;
        if (out_valid) {
            out_valid[i] = 1;
            out_u[i] = u;
            out_v[i] = v;
            out_level[i] = nPredictedLevel;
        }
        const float radius = th * cur.scaleFactors[nPredictedLevel];
        grid.FeaturesInArea(cur, u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = &in.mp_desc[32 * (size_t) i];
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (cur_owner[i2]) continue;
            const int dist = descriptor_distance(dMP, &cur.desc[32 * (size_t) i2]);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        // (:1431 dereferences bestIdx2 == -1 when every candidate is occupied and ORBdist >= 256; callers pass 100 / 64)
        if (bestDist <= ORBdist && bestIdx2 >= 0) {
            cur_owner[bestIdx2] = 2;
            cur_match[bestIdx2] = i;
            nmatches++;
            if (checkOrientation) {
                float rot = in.kf_angle[i] - cur.keys[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int) std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    cur_owner[rotHist[i][j]] = 0;
                    cur_match[rotHist[i][j]] = -2;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// :155-263
int search_by_bow(int nNodes, const int *kf_off, const int *kf_idx, const int *f_off, const int *f_idx, const uint8_t *kf_valid,
                  const KeyPoint *kf_keys, const uint8_t *kf_desc, int nF, const KeyPoint *f_keys, const uint8_t *f_desc, float nnratio,
                  bool checkOrientation, int *match) {
    for (int i = 0; i < nF; i++) match[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int k = 0; k < nNodes; k++) {
        for (int a = kf_off[k]; a < kf_off[k + 1]; a++) {
            const int realIdxKF = kf_idx[a];
            if (!kf_valid[realIdxKF]) continue;
            const uint8_t *dKF = &kf_desc[32 * (size_t) realIdxKF];
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            for (int b = f_off[k]; b < f_off[k + 1]; b++) {
                const int realIdxF = f_idx[b];
                if (match[realIdxF] >= 0) continue;  // vpMapPointMatches[realIdxF] already set
                const int dist = descriptor_distance(dKF, &f_desc[32 * (size_t) realIdxF]);
                if (dist < bestDist1) {
                    bestDist2 = bestDist1;
                    bestDist1 = dist;
                    bestIdxF = realIdxF;
                } else if (dist < bestDist2) {
                    bestDist2 = dist;
                }
            }
            if (bestDist1 <= TH_LOW) {
                if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    match[bestIdxF] = realIdxKF;
                    if (checkOrientation) {
                        float rot = kf_keys[realIdxKF].angle - f_keys[bestIdxF].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int) std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(bestIdxF);
                    }
                    nmatches++;
                }
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                match[rotHist[i][j]] = -2;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// CheckDistEpipolarLine :136-153.  F12 row-major: F12(r, c) = F[3 * r + c].
static inline bool check_dist_epipolar_line(const KeyPoint &kp1, const KeyPoint &kp2, const float *F, const float *levelSigma2) {
    const float a = kp1.x * F[0] + kp1.y * F[3] + F[6];
    const float b = kp1.x * F[1] + kp1.y * F[4] + F[7];
    const float c = kp1.x * F[2] + kp1.y * F[5] + F[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * levelSigma2[kp2.octave];   // double comparison, as in the reference
}

// :596-741
int search_for_triangulation(const TriangulationInput &in, bool onlyStereo, bool checkOrientation, int *match12) {
    // epipole in the second image (:601-608)
    float C2[3];
    mat3_mul_vec(in.R2w, in.Cw1, C2);
    for (int k = 0; k < 3; k++) C2[k] = C2[k] + in.t2w[k];
    const float invz = 1.0f / C2[2];
    const float ex = in.fx2 * C2[0] * invz + in.cx2;
    const float ey = in.fy2 * C2[1] * invz + in.cy2;

    for (int i = 0; i < in.n1; i++) match12[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int k = 0; k < in.nNodes; k++) {
        for (int a = in.off1[k]; a < in.off1[k + 1]; a++) {
            const int idx1 = in.idx1[a];
            if (in.has_mp1[idx1]) continue;
            const bool bStereo1 = in.uRight1 && in.uRight1[idx1] >= 0;
            if (onlyStereo && !bStereo1) continue;
            const KeyPoint &kp1 = in.keys1[idx1];
            const uint8_t *d1 = &in.desc1[32 * (size_t) idx1];
            int bestDist = TH_LOW, bestIdx2 = -1;
            for (int b = in.off2[k]; b < in.off2[k + 1]; b++) {
                const int idx2 = in.idx2[b];
                if (in.has_mp2[idx2]) continue;   // vbMatched2 is never set by the reference (:616, :655): only the MapPoint test acts
                const bool bStereo2 = in.uRight2 && in.uRight2[idx2] >= 0;
                if (onlyStereo && !bStereo2) continue;
                const int dist = descriptor_distance(d1, &in.desc2[32 * (size_t) idx2]);
                if (dist > TH_LOW || dist > bestDist) continue;
                const KeyPoint &kp2 = in.keys2[idx2];
                if (!bStereo1 && !bStereo2) {
                    const float distex = ex - kp2.x;
                    const float distey = ey - kp2.y;
                    if (distex * distex + distey * distey < 100 * in.scaleFactors2[kp2.octave]) continue;
                }
                if (check_dist_epipolar_line(kp1, kp2, in.F12, in.levelSigma2_2)) {
                    bestIdx2 = idx2;
                    bestDist = dist;
                }
            }
            if (bestIdx2 >= 0) {
                match12[idx1] = bestIdx2;
                nmatches++;
                if (checkOrientation) {
                    float rot = kp1.angle - in.keys2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int) std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(idx1);
                }
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                match12[rotHist[i][j]] = -2;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// src/Frame.cc:363-422
void is_in_frustum(const FrameView &F, const FrustumInput &in, float viewingCosLimit, uint8_t *in_view, float *projX, float *projY,
                   float *projXR, int *level, float *viewCos) {
    for (int i = 0; i < in.M; i++) {
        in_view[i] = 0;
        const float *P = &in.world[3 * i];
        float Pc[3];
        mat3_mul_vec(in.Rcw, P, Pc);
        for (int k = 0; k < 3; k++) Pc[k] = Pc[k] + in.tcw[k];
        const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
        if (PcZ < 0.0f) continue;
        const float invz = 1.0f / PcZ;
        const float u = F.fx * PcX * invz + F.cx;
        const float v = F.fy * PcY * invz + F.cy;
        if (u < F.minX || u > F.maxX) continue;
        if (v < F.minY || v > F.maxY) continue;
        const float PO[3] = {P[0] - in.Ow[0], P[1] - in.Ow[1], P[2] - in.Ow[2]};
        const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
        if (dist < in.minDistInv[i] || dist > in.maxDistInv[i]) continue;
        const float *Pn = &in.normal[3 * i];
        const float vc = (PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / dist;
        if (vc < viewingCosLimit) continue;
        const float ratio = in.mfMaxDistance[i] / dist;   // MapPoint::PredictScale
        int nScale = (int) std::ceil(std::log(ratio) / in.logScaleFactor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= in.nScaleLevels) nScale = in.nScaleLevels - 1;
        in_view[i] = 1;
        projX[i] = u;
        projXR[i] = u - F.mbf * invz;
        projY[i] = v;
        level[i] = nScale;
        viewCos[i] = vc;
    }
}

// src/MapPoint.cc:211-271
int distinctive_descriptor(const uint8_t *desc, int N) {
    std::vector<float> D((size_t) N * N);
    for (int i = 0; i < N; i++) {
        D[(size_t) i * N + i] = 0;
        for (int j = i + 1; j < N; j++) {
            const int distij = descriptor_distance(&desc[32 * (size_t) i], &desc[32 * (size_t) j]);
            D[(size_t) i * N + j] = (float) distij;
            D[(size_t) j * N + i] = (float) distij;
        }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (int i = 0; i < N; i++) {
        std::vector<int> vDists(D.begin() + (size_t) i * N, D.begin() + (size_t) (i + 1) * N);
        std::sort(vDists.begin(), vDists.end());
        const int median = vDists[(size_t) (0.5 * (N - 1))];
        if (median < BestMedian) {
            BestMedian = median;
            BestIdx = i;
        }
    }
    return BestIdx;
}

// :375-478
int search_for_initialization(const FrameView &F1, const FrameView &F2, const Grid &grid2, float *prevMatchedXY,
                              int windowSize, float nnratio, bool checkOrientation, int *vnMatches12) {
    int nmatches = 0;
    for (int i = 0; i < F1.N; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(F2.N, INT_MAX);
    std::vector<int> vnMatches21(F2.N, -1);
    std::vector<int> vIndices2;
    for (int i1 = 0; i1 < F1.N; i1++) {
        const KeyPoint &kp1 = F1.keys[i1];
        int level1 = kp1.octave;
        if (level1 > 0) continue;
        grid2.FeaturesInArea(F2, prevMatchedXY[2 * i1], prevMatchedXY[2 * i1 + 1], (float) windowSize, level1, level1,
                             vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t *d1 = &F1.desc[32 * (size_t) i1];
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            int dist = descriptor_distance(d1, &F2.desc[32 * (size_t) i2]);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float) bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOrientation) {
                    float rot = F1.keys[i1].angle - F2.keys[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int) std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    for (int i1 = 0; i1 < F1.N; i1++)
        if (vnMatches12[i1] >= 0) {
            prevMatchedXY[2 * i1] = F2.keys[vnMatches12[i1]].x;
            prevMatchedXY[2 * i1 + 1] = F2.keys[vnMatches12[i1]].y;
        }
    return nmatches;
}

}  // namespace ygzo
