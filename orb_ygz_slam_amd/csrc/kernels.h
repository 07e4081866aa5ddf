// kernels.h -- launcher declarations shared by the HIP translation units of libygzf (product code).
#ifndef YGZF_KERNELS_H
#define YGZF_KERNELS_H
#include "ygzf_internal.h"

namespace ygzf {

hipError_t upload_constants(const int *umax16);

void launch_pyr_resize(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, const LevelGeom &g, int level, int nFrames,
                       const int *xofs, const short *xalpha, const int *yofs, const short *ybeta);
void launch_fast_cells(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, int nlevels, int iniTh, int minTh,
                       unsigned short *cellCnt, unsigned *slots, int totalCells, long long totalSlots, int nFrames);
size_t octree_lds_bytes(int maxCellsPerLevel, int cap);
hipError_t octree_prepare(size_t ldsBytes);
void launch_octree(hipStream_t st, const LevelGeom *dGeom, int nlevels, const unsigned short *cellCnt, const unsigned *slots,
                   int totalCells, long long totalSlots, unsigned *k0, unsigned *v0, unsigned *k1, unsigned *v1, unsigned *xy,
                   long long candStride, unsigned *lvlKpXY, unsigned char *lvlKpScore, int *lvlKpCnt, int *lvlCandCnt,
                   int kpStride, int cap, size_t ldsBytes, int nFrames);
void launch_describe(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, int nlevels, const unsigned *lvlKpXY,
                     const unsigned char *lvlKpScore, const int *lvlKpCnt, int kpStride, ygzf_kp *outKp, uint8_t *outDesc,
                     int *outCnt, int outStride, int nFrames);
void launch_hamming_pairs(hipStream_t st, const void *a, const void *b, int n, int *out);

}  // namespace ygzf
#endif
