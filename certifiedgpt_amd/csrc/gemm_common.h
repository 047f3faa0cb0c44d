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

// Exact-erf GELU (nn.GELU default, eva_vit.py:50,61) = x * Phi(x), with erf from Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the fp16 rounding of the output): Phi(|x|) = 1 - 0.5 * P(t) * exp(-x^2/2),
// t = 1/(1 + 0.3275911 |x|/sqrt2).  13 VALU ops (2 transcendental) instead of ocml erff's ~40: the fc1 epilogue runs 128
// of these per lane and is NOT hidden behind MFMAs at one workgroup per CU.
__device__ __forceinline__ float gelu_erf(float x) {
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);       // exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
    float pl = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);                   // 0.5 * A&S 7.1.26 polynomial
    pl = fmaf(t, pl, 0.5f * 1.421413741f);
    pl = fmaf(t, pl, 0.5f * -0.284496736f);
    pl = fmaf(t, pl, 0.5f * 0.254829592f);
    const float h = pl * t * e;                                                      // 0.5 * erfc(|x|/sqrt2) = 1 - Phi(|x|)
    return fmaf(-fabsf(x), h, fmaxf(x, 0.0f));                                       // x>=0: x - x h ; x<0: x h
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
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
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
            if constexpr (EPI == EPI_F16_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
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
