// gemm_common.h -- device helpers shared by the GEMM translation units (gemm.hip, gemm6.hip): vector types, the exact-erf
// GELU, the XCD-aware block->tile map and the vector-store epilogue of the 256-row kernels.
#pragma once
#include "kernels.h"

namespace cgpt {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {


constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GROUP_M = 8;

// Exact-erf GELU (nn.GELU default, eva_vit.py:50,61) = x * Phi(x).  The fc1 epilogue runs 128 of these per lane with the matrix
// pipe idle (both waves of a SIMD reach the epilogue together and share its VALU: in-kernel stamps, 11-15 k of a tile's 73 k
// cycles), so the VALU op count is what matters.  With t = |x|:
//     gelu(x) = max(x, 0) - t * (1 - Phi(t)),        1 - Phi(t) = 2^Q(t),   Q = log2 of the upper normal tail,
// Q is smooth (-1 - 1.15 t - 0.46 t^2 ... -> -t^2/2 log2 e), so a degree-7 polynomial fitted for t * |error of 2^Q| -- the error
// of GELU itself -- is enough: <= 2.7e-7 evaluated in fp32 (fp32 resolution at |x| = 4 is 2.4e-7), 0.05 % relative in the tail
// -4.2 < x < -3 (fp16 output resolution is 5e-4 relative), and Q(t) <= -29.9 for every t > 6 up to overflow (negative leading
// coefficient), so large inputs give exactly max(x, 0).  Per pair of values: 2 v_and (|x|), 7 v_pk_fma_f32, 2 v_exp_f32,
// 2 v_max_f32, 1 v_pk_fma_f32 = 14 instructions; the form used before -- 1 - Phi = 0.5 (1 + c1 u + ... + c7 u^7)^-16, u = -t/2, i.e.
// the same degree plus a reciprocal AND four squarings (Abramowitz-Stegun 7.1.28) -- took 18 at the same accuracy (3.6e-7 / 0.1 %).
// gelu_erf2 evaluates two values with v_pk_fma_f32 (IEEE per component); every epilogue path goes through it, so a value depends
// on nothing but the element.
#define CGPT_GELU_Q0 -9.999910593e-01f
#define CGPT_GELU_Q1 -1.151225209e+00f
#define CGPT_GELU_Q2 -4.586824775e-01f
#define CGPT_GELU_Q3 -5.356145650e-02f
#define CGPT_GELU_Q4 8.207941428e-03f
#define CGPT_GELU_Q5 -8.253850974e-04f
#define CGPT_GELU_Q6 4.516868648e-05f
#define CGPT_GELU_Q7 -9.805353329e-07f
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 t = {fabsf(x[0]), fabsf(x[1])};
    auto k = [](float v) { return f32x2{v, v}; };
    f32x2 q = __builtin_elementwise_fma(t, k(CGPT_GELU_Q7), k(CGPT_GELU_Q6));
    q = __builtin_elementwise_fma(q, t, k(CGPT_GELU_Q5));
    q = __builtin_elementwise_fma(q, t, k(CGPT_GELU_Q4));
    q = __builtin_elementwise_fma(q, t, k(CGPT_GELU_Q3));
    q = __builtin_elementwise_fma(q, t, k(CGPT_GELU_Q2));
    q = __builtin_elementwise_fma(q, t, k(CGPT_GELU_Q1));
    q = __builtin_elementwise_fma(q, t, k(CGPT_GELU_Q0));
    const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    return __builtin_elementwise_fma(-t, e, f32x2{fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)});
}
// One value: the same arithmetic, result rounded to fp32 before anything else happens to it.  A plain fmaf version is not bit-identical
// once its result is converted to fp16: the compiler fuses the last fma with the conversion (v_fma_mix: one rounding), the packed
// path rounds to fp32 first and converts afterwards -- 6 of 192 000 outputs differed by one fp16 ulp.
__device__ __forceinline__ float gelu_erf(float x) {
    float r = gelu_erf2(f32x2{x, x})[0];
    asm volatile("" : "+v"(r));                          // the fp32 result is materialised before any conversion
    return r;
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// Every EPI_F16_GELU path evaluates GELU on the fp32 value acc + bias (the reference's autocast rounds the linear output to fp16 first,
// eva_vit.py:59-61 with base_model.py:141-142: this is at least as precise) and converts the fp32 result to fp16 -- the same
// arithmetic in every kernel and tile path, so a value depends on nothing but the element (tests/test_gpu_kernels.py).  Rounding
// through fp16 first (tried in round 3 for a deferred GELU, profiles/r03/gemm_deferred_gelu.txt) costs one more VALU instruction per
// value in an epilogue that is VALU-bound.
__device__ __forceinline__ f16x4 gelu_f16x4(f32x4 v) {
    f32x2 a = gelu_erf2(f32x2{v[0], v[1]}), b = gelu_erf2(f32x2{v[2], v[3]});
    asm volatile("" : "+v"(a), "+v"(b));                 // fp32 results materialised before the conversions (no fused fma + cvt)
    return f16x4{(half_t)a[0], (half_t)a[1], (half_t)b[0], (half_t)b[1]};
}
template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int m, int n, float v) {
    if constexpr (EPI == EPI_F16) {
        reinterpret_cast<half_t*>(p.out)[(int64_t)m * p.ldo + n] = (half_t)v;
    } else if constexpr (EPI == EPI_F16_GELU) {
        reinterpret_cast<half_t*>(p.out)[(int64_t)m * p.ldo + n] = (half_t)gelu_erf(v);
    } else if constexpr (EPI == EPI_F32) {
        reinterpret_cast<float*>(p.out)[(int64_t)m * p.ldo + n] = v;
    } else if constexpr (EPI == EPI_RESID) {
        v += p.aux[(int64_t)m * p.ldaux + n];
        reinterpret_cast<float*>(p.out)[(int64_t)m * p.ldo + n] = v;
    } else {  // EPI_PATCH
        const int b = m / p.patches, pp = m - b * p.patches;
        v += p.aux[(int64_t)(1 + pp) * p.ldaux + n];
        reinterpret_cast<float*>(p.out)[((int64_t)b * (p.patches + 1) + 1 + pp) * p.ldo + n] = v;
    }
}

// Block -> tile map shared by both kernels (speed only): XCD-contiguous chunks, then groups of GROUP_M tile-rows
// walked column-major.
__device__ __forceinline__ void tile_of_block(int tiles_m, int tiles_n, int& tm, int& tn) {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = GROUP_M * tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_grp = t - grp * per_group;
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
}

// Same map for a VIRTUAL block id t in [0, nwg) (persistent kernel: physical block b walks t = b, b + grid, ...; with a grid
// that is a multiple of 8, t and b sit on the same XCD).
__device__ __forceinline__ void tile_of_virtual_block(int vb, int nwg, int tiles_m, int tiles_n, int& tm, int& tn, int group_m = GROUP_M) {
    const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, idx = vb >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = group_m * tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_grp = t - grp * per_group;
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
}

// Epilogue shared by the 256-row kernels: the lane owns rows m0 + i*16 and columns n0 + j*16 .. +3 (transposed
// accumulator tiles), so every (i, j) is one 8- or 16-byte store.  `full` = the tile lies completely inside C: then there
// are no guards and no loads, and the stores issue back to back (a load here would make every later use wait for the
// previous STORE: vmcnt retires in order).
// FENCED: a scheduling fence after every row of tiles (keeps the 256-accumulator kernel from hoisting all AGPR reads).
template <int EPI, int TM, int TN, bool FENCED = false>
__device__ __forceinline__ void gemm_epilogue_256(const GemmParams& p, f32x4 (&acc)[TM][TN], f32x4 (&bias4)[TN], int m0, int n0,
                                                  bool full) {
    auto emit = [&](int i, int j, int m, int n) {
        f32x4 v = acc[i][j] + bias4[j];
        int64_t orow = (int64_t)m * p.ldo;
        if constexpr (EPI == EPI_PATCH) {
            const int b = m / p.patches, pp = m - b * p.patches;
            orow = ((int64_t)b * (p.patches + 1) + 1 + pp) * p.ldo;
            v += *reinterpret_cast<const f32x4*>(p.aux + (int64_t)(1 + pp) * p.ldaux + n);
        }
        if constexpr (EPI == EPI_RESID) v += *reinterpret_cast<const f32x4*>(p.aux + (int64_t)m * p.ldaux + n);
        if constexpr (EPI == EPI_F16 || EPI == EPI_F16_GELU) {
            f16x4 hv;
            if constexpr (EPI == EPI_F16_GELU) hv = gelu_f16x4(v);
            else hv = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<f16x4*>(reinterpret_cast<half_t*>(p.out) + orow + n) = hv;
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + orow + n) = v;
        }
    };
    if (full) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) emit(i, j, m0 + i * 16, n0 + j * 16);
            if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + i * 16;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + j * 16;
                if (n >= p.N) continue;
                if ((p.ablate & 2) && acc[i][j][0] != 123.456f) continue;
                emit(i, j, m, n);
            }
            if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

}  // namespace
}  // namespace cgpt
