// tools/micro/valu_peak.hip -- calibrates the vector-ALU issue ceiling of gfx950 for the byte-parallel integer instructions the FAST /
// describe / pyramid kernels are built from (is a wave64 VALU instruction 4 cycles -- SIMD16 -- or 2 -- SIMD32?).
// Every kernel runs kIter x 64 dependent-free instructions of one kind per wave (8 independent accumulator chains per lane, so the
// 4-8 cycle result latency never stalls issue) at 1, 2, 4 and 8 waves per SIMD on all 256 CUs and reports
//     cycles per wave-instruction per SIMD = (SIMDs x shader-clock cycles of the launch) / (waves x instructions per wave)
// with the shader clock taken from s_memtime deltas inside the kernel (clock64) and from wall time x 2.4 GHz beside it.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <vector>

constexpr int kIter = 2000, kUnroll = 8, kChains = 8;   // 2000 x 8 x 8 = 128000 instructions per wave

#define CHAIN8(OP)                                                                       \
    for (int it = 0; it < kIter; it++) {                                                 \
        _Pragma("unroll") for (int u = 0; u < kUnroll; u++) {                            \
            OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)                      \
        }                                                                                \
    }

#define OP_LERP(x) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_BITOP3(x) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(b), "v"(c));
#define OP_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_ALIGN(x) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(x) : "v"(b));
#define OP_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_DOT4(x) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c));
#define OP_SAD(x) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c));
#define OP_PKMIN(x) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_PKADD(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_OR3(x) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x##w) : "v"(bw), "v"(cw));
#define OP_CMP(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define OP_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define OP_DPP(x) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
#define OP_MBCNT(x) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x) : "v"(b));
#define OP2(NAME, INS) 
#define OP_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_OR(x) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_LSHL(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
#define OP_LSHR(x) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x));
#define OP_MINU(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_SUB(x) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
#define OP_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
#define OP_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_BFE(x) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(x));
#define OP_BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(b));
#define OP_CNDS(x) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(msk));
#define OP_CMPS(x) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(sm) : "v"(x), "v"(b));
#define OP_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_MIN3(x) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_MED3(x) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_PKMAX(x) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_PKSUB(x) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_PKLSHL(x) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(x));
#define OP_PKMUL(x) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MULF(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_ADDF(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_CVTF(x) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
#define OP_CVTUB(x) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x));
#define OP_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_XAD(x) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_SAD16(x) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c));
#define OP_ADDLSHL(x) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(x) : "v"(b));
#define OP_NOT(x) asm volatile("v_not_b32 %0, %0" : "+v"(x));
#define OP_BCNT(x) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x) : "v"(b));
#define OP_SDWA(x) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(x) : "v"(b));
#define OP_DPPROR(x) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
#define OP_BITOP2(x) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xfe" : "+v"(x) : "v"(b), "v"(c));

#define KERNEL(NAME, OP)                                                                                         \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, long long *clk, unsigned seed) {                  \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        unsigned b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x5bd1e995u;                                   \
        unsigned long long msk = 0x5555aaaa3333ccccull * seed, sm = 0;                                           \
        unsigned z = a0 * 23, z2 = a0 * 29, z3 = a0 * 31;                                                        \
        const long long t0 = clock64(), w0 = wall_clock64();                                                     \
        CHAIN8(OP)                                                                                               \
        const long long t1 = clock64(), w1 = wall_clock64();                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned) sm ^ z ^ z2 ^ z3; \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { clk[0] = t1 - t0; clk[1] = w1 - w0; }             \
    }

KERNEL(k_lerp, OP_LERP)
KERNEL(k_bitop3, OP_BITOP3)
KERNEL(k_add, OP_ADD)
KERNEL(k_align, OP_ALIGN)
KERNEL(k_perm, OP_PERM)
KERNEL(k_dot4, OP_DOT4)
KERNEL(k_sad, OP_SAD)
KERNEL(k_pkmin, OP_PKMIN)
KERNEL(k_pkadd, OP_PKADD)
KERNEL(k_mul24, OP_MUL24)
KERNEL(k_mullo, OP_MULLO)
KERNEL(k_or3, OP_OR3)
KERNEL(k_fma, OP_FMA)
KERNEL(k_cmp, OP_CMP)
KERNEL(k_cndmask, OP_CNDMASK)
KERNEL(k_dpp, OP_DPP)
KERNEL(k_mbcnt, OP_MBCNT)
KERNEL(k_and, OP_AND) KERNEL(k_or, OP_OR) KERNEL(k_xor, OP_XOR) KERNEL(k_lshl, OP_LSHL) KERNEL(k_lshr, OP_LSHR) KERNEL(k_minu, OP_MINU)
KERNEL(k_sub, OP_SUB) KERNEL(k_andor, OP_ANDOR) KERNEL(k_lshlor, OP_LSHLOR) KERNEL(k_lshladd, OP_LSHLADD) KERNEL(k_add3, OP_ADD3)
KERNEL(k_bfe, OP_BFE) KERNEL(k_bfi, OP_BFI) KERNEL(k_mov, OP_MOV) KERNEL(k_cnds, OP_CNDS) KERNEL(k_cmps, OP_CMPS) KERNEL(k_mad24, OP_MAD24)
KERNEL(k_min3, OP_MIN3) KERNEL(k_med3, OP_MED3) KERNEL(k_pkmax, OP_PKMAX) KERNEL(k_pksub, OP_PKSUB) KERNEL(k_pklshl, OP_PKLSHL)
KERNEL(k_pkmul, OP_PKMUL) KERNEL(k_mulf, OP_MULF) KERNEL(k_addf, OP_ADDF) KERNEL(k_cvtf, OP_CVTF) KERNEL(k_cvtub, OP_CVTUB)
KERNEL(k_mulhi, OP_MULHI) KERNEL(k_xad, OP_XAD) KERNEL(k_sad16, OP_SAD16) KERNEL(k_addlshl, OP_ADDLSHL) KERNEL(k_not, OP_NOT)
KERNEL(k_bcnt, OP_BCNT) KERNEL(k_sdwa, OP_SDWA) KERNEL(k_dppror, OP_DPPROR) KERNEL(k_bitop2, OP_BITOP2)

__global__ __launch_bounds__(256) void k_pkfma(unsigned *out, long long *clk, unsigned seed) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 a0w = {1.f + threadIdx.x, 2.f}, a1w = a0w * 3.f, a2w = a0w * 5.f, a3w = a0w * 7.f, a4w = a0w * 0.5f, a5w = a0w * 0.25f, a6w = a0w * 9.f, a7w = a0w * 1.5f;
    v2 bw = {0.999f, 1.001f}, cw = {1e-3f * seed, 2e-3f};
    const long long t0 = clock64(), w0 = wall_clock64();
    CHAIN8(OP_PKFMA)
    const long long t1 = clock64(), w1 = wall_clock64();
    const v2 s = a0w + a1w + a2w + a3w + a4w + a5w + a6w + a7w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s.x + s.y);
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// double-precision forms (k_describe's sin / cos) and mixes of one vector with one or two scalar instructions of the same wave (does a
// scalar instruction cost the wave a vector issue slot?)
#define KERNEL_F64(NAME, INS)                                                                                    \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, long long *clk, unsigned seed) {                  \
        double a0 = 1.0 + threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 0.5, a5 = a0 * 0.25, a6 = a0 * 9, a7 = a0 * 1.5; \
        double b = 0.999999 + 1e-9 * seed, c = 1e-6 * seed;                                                      \
        const long long t0 = clock64(), w0 = wall_clock64();                                                     \
        CHAIN8(INS)                                                                                              \
        const long long t1 = clock64(), w1 = wall_clock64();                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned) (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);         \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { clk[0] = t1 - t0; clk[1] = w1 - w0; }             \
    }
#define OP_MULF64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_ADDF64(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_FMAF64(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
KERNEL_F64(k_mulf64, OP_MULF64) KERNEL_F64(k_addf64, OP_ADDF64) KERNEL_F64(k_fmaf64, OP_FMAF64)
#define OP_ADD_S1(x) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 1" : "+v"(x), "+s"(sm) : "v"(b));
#define OP_ADD_S2(x) asm volatile("v_add_u32 %0, %0, %3\n\ts_add_u32 %1, %1, 1\n\ts_xor_b32 %2, %2, %1" : "+v"(x), "+s"(sm), "+s"(msk) : "v"(b));
#define OP_LERP_S1(x) asm volatile("v_lerp_u8 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 1" : "+v"(x), "+s"(sm) : "v"(b), "v"(c));
#define KERNEL_S(NAME, OP)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, long long *clk, unsigned seed) {                  \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        unsigned b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x5bd1e995u;                                   \
        unsigned msk = __builtin_amdgcn_readfirstlane(0x3333ccccu * seed), sm = __builtin_amdgcn_readfirstlane(seed);                                                           \
        const long long t0 = clock64(), w0 = wall_clock64();                                                     \
        CHAIN8(OP)                                                                                               \
        const long long t1 = clock64(), w1 = wall_clock64();                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ sm ^ msk;           \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { clk[0] = t1 - t0; clk[1] = w1 - w0; }             \
    }
KERNEL_S(k_add_s1, OP_ADD_S1) KERNEL_S(k_add_s2, OP_ADD_S2) KERNEL_S(k_lerp_s1, OP_LERP_S1)

// mixes of a quarter-rate-class (4-cycle) and a half-rate-class (2-cycle) instruction in ONE wave's stream: does the 2-cycle instruction keep its
// rate between 4-cycle neighbours?  (the FAST pass 1 is 32 v_lerp_u8 + 84 v_bitop3 + 18 v_alignbyte per 4 pixels)
#define OP_MIX11(x) asm volatile("v_lerp_u8 %0, %0, %1, %2\n\tv_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(b), "v"(c));
#define OP_MIX13(x) asm volatile("v_lerp_u8 %0, %0, %1, %2\n\tv_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\tv_bitop3_b32 %0, %0, %2, %1 bitop3:0x96\n\tv_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(x) : "v"(b), "v"(c));
#define OP_MIXPA(x) asm volatile("v_perm_b32 %0, %0, %1, %2\n\tv_add_u32 %0, %0, %1" : "+v"(x) : "v"(b), "v"(c));
#define OP_MIXMA(x) asm volatile("v_mul_hi_u32 %0, %0, %1\n\tv_and_b32 %0, %0, %2" : "+v"(x) : "v"(b), "v"(c));
// the same mixes with the half-rate instructions on accumulators of their own (no instruction waits for its neighbour's result)
#define OP_MIXI11(x) asm volatile("v_lerp_u8 %0, %0, %2, %3\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96" : "+v"(x), "+v"(z) : "v"(b), "v"(c));
#define OP_MIXI13(x) asm volatile("v_lerp_u8 %0, %0, %4, %5\n\tv_bitop3_b32 %1, %1, %4, %5 bitop3:0x96\n\tv_bitop3_b32 %2, %2, %5, %4 bitop3:0x96\n\tv_bitop3_b32 %3, %3, %4, %5 bitop3:0xe8" : "+v"(x), "+v"(z), "+v"(z2), "+v"(z3) : "v"(b), "v"(c));
#define OP_MIXIPA(x) asm volatile("v_perm_b32 %0, %0, %2, %3\n\tv_add_u32 %1, %1, %2" : "+v"(x), "+v"(z) : "v"(b), "v"(c));
#define OP_MIXI31(x) asm volatile("v_lerp_u8 %0, %0, %2, %3\n\tv_perm_b32 %0, %0, %2, %3\n\tv_alignbyte_b32 %0, %0, %2, 1\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96" : "+v"(x), "+v"(z) : "v"(b), "v"(c));
KERNEL(k_mixi11, OP_MIXI11) KERNEL(k_mixi13, OP_MIXI13) KERNEL(k_mixipa, OP_MIXIPA) KERNEL(k_mixi31, OP_MIXI31)
KERNEL(k_mix11, OP_MIX11) KERNEL(k_mix13, OP_MIX13) KERNEL(k_mixpa, OP_MIXPA) KERNEL(k_mixma, OP_MIXMA)

typedef void (*kern_t)(unsigned *, long long *, unsigned);

int main(int argc, char **argv) {   // optional arguments: substrings of the instruction names to run
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, simds = cus * 4;
    printf("device %s: %d CUs, clockRate %d kHz\n", p.gcnArchName, cus, p.clockRate);
    unsigned *out;
    long long *clk;
    hipMalloc(&out, (size_t) cus * 8 * 256 * 4 + 1024);
    hipMalloc(&clk, 64);
    struct { const char *name; kern_t k; int mult; } ks[] = {{"indep lerp+bitop3 (per pair)", k_mixi11, 1}, {"indep lerp+3 bitop3 (per 4)", k_mixi13, 1}, {"indep perm+add (per pair)", k_mixipa, 1}, {"indep 3 quarter + bitop3 (per 4)", k_mixi31, 1}, {"lerp+bitop3 (per pair)", k_mix11, 1}, {"lerp+3 bitop3 (per 4)", k_mix13, 1}, {"perm+add (per pair)", k_mixpa, 1}, {"mul_hi+and (per pair)", k_mixma, 1}, {"v_lerp_u8", k_lerp}, {"v_bitop3_b32", k_bitop3}, {"v_or3_b32", k_or3}, {"v_add_u32", k_add}, {"v_alignbyte_b32", k_align},
                                                   {"v_perm_b32", k_perm}, {"v_dot4_u32_u8", k_dot4}, {"v_sad_u8", k_sad}, {"v_pk_min_u16", k_pkmin}, {"v_pk_add_u16", k_pkadd},
                                                   {"v_mul_u32_u24", k_mul24}, {"v_mul_lo_u32", k_mullo}, {"v_cmp_lt_u32", k_cmp}, {"v_cndmask_b32", k_cndmask},
                                                   {"v_add_u32_dpp", k_dpp}, {"v_mbcnt_lo", k_mbcnt}, {"v_fma_f32", k_fma}, {"v_pk_fma_f32", k_pkfma},
                                                   {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_lshl}, {"v_lshrrev_b32", k_lshr},
                                                   {"v_min_u32", k_minu}, {"v_sub_u32", k_sub}, {"v_and_or_b32", k_andor}, {"v_lshl_or_b32", k_lshlor},
                                                   {"v_lshl_add_u32", k_lshladd}, {"v_add3_u32", k_add3}, {"v_bfe_u32", k_bfe}, {"v_bfi_b32", k_bfi}, {"v_mov_b32", k_mov},
                                                   {"v_cndmask (sgpr)", k_cnds}, {"v_cmp -> sgpr", k_cmps}, {"v_mad_u32_u24", k_mad24}, {"v_min3_u32", k_min3},
                                                   {"v_med3_u32", k_med3}, {"v_pk_max_u16", k_pkmax}, {"v_pk_sub_u16", k_pksub}, {"v_pk_lshlrev_b16", k_pklshl},
                                                   {"v_pk_mul_lo_u16", k_pkmul}, {"v_mul_f32", k_mulf}, {"v_add_f32", k_addf}, {"v_cvt_f32_u32", k_cvtf},
                                                   {"v_cvt_f32_ubyte0", k_cvtub}, {"v_mul_hi_u32", k_mulhi}, {"v_xad_u32", k_xad}, {"v_sad_u16", k_sad16},
                                                   {"v_add_lshl_u32", k_addlshl}, {"v_not_b32", k_not}, {"v_bcnt_u32_b32", k_bcnt}, {"v_add_u32_sdwa", k_sdwa},
                                                   {"v_mov_b32_dpp ror", k_dppror}, {"v_bitop3 (or3)", k_bitop2},
                                                   {"v_mul_f64", k_mulf64}, {"v_add_f64", k_addf64}, {"v_fma_f64", k_fmaf64},
                                                   {"v_add + 1 s_add", k_add_s1}, {"v_add + 2 salu", k_add_s2}, {"v_lerp + 1 s_add", k_lerp_s1}};
    const double instr = (double) kIter * kUnroll * kChains;
    printf("%-18s %s\n", "instruction", "waves/SIMD: shader-clock cycles per wave-instruction per SIMD @ measured GHz (s_memtime / s_memrealtime)");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &k : ks) {
        bool want = argc < 2;
        for (int i = 1; i < argc; i++) want = want || strstr(k.name, argv[i]) != nullptr;
        if (!want) continue;
        printf("%-18s", k.name);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;   // 256-thread blocks: one wave per SIMD each
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, clk, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, clk, 2u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long c[2];
            hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            // shader clock while this kernel ran: s_memtime ticks per s_memrealtime tick (100 MHz) over one wave's run
            const double ghz = c[1] > 0 ? (double) c[0] / (double) c[1] * 0.1 : 0;
            const double cyc = ms * 1e-3 * ghz * 1e9 / (instr * wps);
            printf("  %d: %5.2f @%4.2f", wps, cyc, ghz);
        }
        printf("\n");
    }
    return 0;
}
