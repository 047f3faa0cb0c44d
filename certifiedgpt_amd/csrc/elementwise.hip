// elementwise.hip -- the HBM-bound kernels of the hot path: Gaussian perturbation fused with the patch-embed
// im2col, LayerNorm, CLS/pos rows, query-token broadcast, mean pool, argmax + vote histogram, weight fills/casts.
//
// Reference ops replaced: smoothing.py:95-98,101-105 (repeat + randn*sigma; argmax(1); _count_arr),
// eva_vit.py:162,168 + base_model.py:281-287 (LayerNorm), eva_vit.py:337-340 (CLS concat + pos_embed),
// minigpt4.py:132 (query_tokens.expand), Qformer.py:106 (embeddings LayerNorm).
#include "kernels.h"
#include "philox.h"
#include "stats.h"

namespace cgpt {

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------ LayerNorm
// One wave per row; the row lives in registers (NCH float2 per lane) between the mean, the centred variance
// and the normalise pass, so x is read from HBM exactly once.  Requires D even and D <= NCH*128.
// With `delta` (fp16) the kernel first applies the pending residual update x += delta and writes x back: the
// reference's `x = x + attn(...)` / `x = x + mlp(...)` (eva_vit.py:180-181), where under autocast the branch output is
// an fp16 tensor added to the fp32 stream -- done here instead of as a read-modify-write in the GEMM epilogue.
// Two pending updates (delta, then delta2, added in that order) and keep_x (normalise x + delta [+ delta2] but leave x as it
// is) let a transformer block write the fp32 stream ONCE: LN1 reads x + (previous fc2 output) without writing, LN2 reads
// x + (previous fc2 output) + (this block's proj output) and writes the sum -- same additions in the same order.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, int64_t ldx,
                                                        const half_t* __restrict__ delta, int64_t ldd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        half_t* __restrict__ y16, int64_t ldy16,
                                                        float* __restrict__ y32, int64_t ldy32, int64_t rows, int D,
                                                        const half_t* __restrict__ delta2, int64_t ldd2, int keep_x) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + row * ldx;
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    float2 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 128 + lane * 2;
        v[c] = (col < D) ? *reinterpret_cast<const float2*>(xr + col) : make_float2(0.f, 0.f);
        if (delta && col < D) {
            const f16x2 d = *reinterpret_cast<const f16x2*>(delta + row * ldd + col);
            v[c].x += (float)d[0];
            v[c].y += (float)d[1];
        }
        if (delta2 && col < D) {
            const f16x2 d = *reinterpret_cast<const f16x2*>(delta2 + row * ldd2 + col);
            v[c].x += (float)d[0];
            v[c].y += (float)d[1];
        }
        if ((delta || delta2) && !keep_x && col < D) *reinterpret_cast<float2*>(xr + col) = v[c];
        s += v[c].x + v[c].y;
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 128 + lane * 2;
        if (col < D) {
            const float dx = v[c].x - mean, dy = v[c].y - mean;
            q += dx * dx + dy * dy;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 128 + lane * 2;
        if (col < D) {
            const float2 gm = *reinterpret_cast<const float2*>(gamma + col);
            const float2 bt = *reinterpret_cast<const float2*>(beta + col);
            const float a = (v[c].x - mean) * rstd * gm.x + bt.x;
            const float b = (v[c].y - mean) * rstd * gm.y + bt.y;
            if (y16) {
                *reinterpret_cast<f16x2*>(y16 + row * ldy16 + col) = f16x2{(half_t)a, (half_t)b};
            }
            if (y32) *reinterpret_cast<float2*>(y32 + row * ldy32 + col) = make_float2(a, b);
        }
    }
}


// 16-byte variant for D % 128 == 0: one row per HALF-wave (32 lanes x NCH float4), two rows per wave; the row still lives
// in registers between the three passes.  Wider accesses than the float2 kernel (16 B per lane instead of 8 B).
template <int NCH>
__device__ __forceinline__ void layernorm4_body(float* __restrict__ x, int64_t ldx,
                                                         const half_t* __restrict__ delta, int64_t ldd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, half_t* __restrict__ y16, int64_t ldy16,
                                                         float* __restrict__ y32, int64_t ldy32, int64_t rows, int D,
                                                         const half_t* __restrict__ delta2, int64_t ldd2, int keep_x) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int l32 = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = row < rows;                        // keep the whole wave in the shuffles
    const int64_t rr = live ? row : rows - 1;
    float* xr = x + rr * ldx;
    float4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 128 + l32 * 4;
        {
            typedef float f32x4nt __attribute__((ext_vector_type(4)));
            const f32x4nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(xr + col));
            v[c] = make_float4(t[0], t[1], t[2], t[3]);
        }
        if (delta) {
            const f16x4 d = __builtin_nontemporal_load(reinterpret_cast<const f16x4*>(delta + rr * ldd + col));
            v[c].x += (float)d[0]; v[c].y += (float)d[1]; v[c].z += (float)d[2]; v[c].w += (float)d[3];
        }
        if (delta2) {
            const f16x4 d = __builtin_nontemporal_load(reinterpret_cast<const f16x4*>(delta2 + rr * ldd2 + col));
            v[c].x += (float)d[0]; v[c].y += (float)d[1]; v[c].z += (float)d[2]; v[c].w += (float)d[3];
        }
        if ((delta || delta2) && !keep_x && live) {
            typedef float f32x4nt __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(f32x4nt{v[c].x, v[c].y, v[c].z, v[c].w}, reinterpret_cast<f32x4nt*>(xr + col));
        }
        s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)D + eps);
    if (!live) return;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 128 + l32 * 4;
        const float4 gm = *reinterpret_cast<const float4*>(gamma + col);
        const float4 bt = *reinterpret_cast<const float4*>(beta + col);
        const float o0 = (v[c].x - mean) * rstd * gm.x + bt.x, o1 = (v[c].y - mean) * rstd * gm.y + bt.y;
        const float o2 = (v[c].z - mean) * rstd * gm.z + bt.z, o3 = (v[c].w - mean) * rstd * gm.w + bt.w;
        if (y16) *reinterpret_cast<f16x4*>(y16 + row * ldy16 + col) = f16x4{(half_t)o0, (half_t)o1, (half_t)o2, (half_t)o3};
        if (y32) *reinterpret_cast<float4*>(y32 + row * ldy32 + col) = make_float4(o0, o1, o2, o3);
    }
}

#define CGPT_LN4_ARGS float* __restrict__ x, int64_t ldx, const half_t* __restrict__ delta, int64_t ldd, const float* __restrict__ gamma, \
                      const float* __restrict__ beta, float eps, half_t* __restrict__ y16, int64_t ldy16, float* __restrict__ y32, \
                      int64_t ldy32, int64_t rows, int D, const half_t* __restrict__ delta2, int64_t ldd2, int keep_x
#define CGPT_LN4_PASS x, ldx, delta, ldd, gamma, beta, eps, y16, ldy16, y32, ldy32, rows, D, delta2, ldd2, keep_x
template <int NCH>
__global__ __launch_bounds__(256) void layernorm4_kernel(CGPT_LN4_ARGS) { layernorm4_body<NCH>(CGPT_LN4_PASS); }
// The ViT-G width (D = 1408, NCH = 11) with amdgpu_waves_per_eu(6, 8): 80 instead of 86 VGPRs, six waves per SIMD instead of five:
// 187 -> 181 us per launch in the model (7: 72 VGPRs with 2 spilled, no further gain; 8: 64 VGPRs, 6 spilled, slower).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void layernorm4_vitg_kernel(CGPT_LN4_ARGS) {
    layernorm4_body<11>(CGPT_LN4_PASS);
}

// x[row, :] += delta[row, :]  (the last block's pending residual update, when no LayerNorm follows on those rows)
__global__ __launch_bounds__(256) void add_delta_kernel(float* __restrict__ x, int64_t ldx, const half_t* __restrict__ delta,
                                                        int64_t ldd, int64_t rows, int D, int skip_mod) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one float2 per thread
    const int half_d = D >> 1;
    if (i >= rows * half_d) return;
    const int64_t r = i / half_d;
    if (skip_mod > 0 && r % skip_mod == 0) return;                           // rows already updated (the CLS rows)
    const int col = (int)(i - r * half_d) * 2;
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    float2 v = *reinterpret_cast<const float2*>(x + r * ldx + col);
    const f16x2 d = *reinterpret_cast<const f16x2*>(delta + r * ldd + col);
    v.x += (float)d[0];
    v.y += (float)d[1];
    *reinterpret_cast<float2*>(x + r * ldx + col) = v;
}

// ------------------------------------------------------------------ noise (+ im2col of the patch embedding)
// Destination of pixel (c, y, x) in the im2col matrix: row = b*P + (y/ps)*(img/ps) + x/ps,
// column = c*ps*ps + (y%ps)*ps + x%ps   (= the flattening of Conv2d weight [D,3,ps,ps], eva_vit.py:202).
// Sample index of batch row b: first_sample + b for b < na, else first_b + (b - na) (two index ranges in one batch:
// the selection and estimation draws of Smooth.certify, smoothing.py:44,48).
// Several images (per > 0): the rows of ALL images form one image-major sequence, image i owning rows [i*per, (i+1)*per)
// with the same two-range split; it reads the clean image src + i*3*img*img and shifts both ranges by i*img_stride sample
// indices.  A batch is any window of that sequence: batch row b is sequence row row0 + b, so batches need not start or end
// at image boundaries (the batch size can then be chosen for the GEMMs' tile quantisation, not for the image's draw count).
// One workgroup per (batch row b, patch row py): it reads the 3 x ps image rows of that patch row with coalesced float4
// loads, adds the noise, and stages the fp16 values of the img/ps patches in LDS as [patch][c*ps*ps + iy*ps + ix]; every
// patch row of A (3*ps*ps contiguous halfs = 1176 B at ps = 14) then leaves in 8-byte vector stores, one row after the other
// (the direct form wrote 2-byte stores scattered over img/ps rows per pixel group).  The noise of pixel group `grp` depends
// only on (seed, sample, grp), so the values are the ones the direct form produced.
template <bool NOISE>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ src, int img, int ps,
                                                     int64_t first_sample, int na, int64_t first_b, int nb, float sigma,
                                                     uint64_t seed, half_t* __restrict__ A, int64_t lda, int per,
                                                     int64_t img_stride, int64_t row0) {
    extern __shared__ __attribute__((aligned(16))) half_t patch_lds[];   // [pw][kp]
    const int pw = img / ps, kp = 3 * ps * ps;
    const int b = blockIdx.x / pw, py = blockIdx.x - b * pw;
    // NOISE: src is the clean image(s) x[.,3,img,img]; else src is the batch [nb,3,img,img]
    const int64_t rs = row0 + b;                                 // row in the sequence of all images
    const int im = (NOISE && per > 0) ? (int)(rs / per) : 0, j = (NOISE && per > 0) ? (int)(rs - (int64_t)im * per) : b;
    const float* image = src + (NOISE ? (int64_t)im : (int64_t)b) * 3 * img * img;
    const int64_t sample = NOISE ? (j < na ? first_sample + j : first_b + (j - na)) + im * img_stride : 0;
    const int xg_per_row = img >> 2, groups = 3 * ps * xg_per_row;        // float4 groups of this patch row
    for (int g = threadIdx.x; g < groups; g += 256) {
        const int c = g / (ps * xg_per_row), r = g - c * ps * xg_per_row;
        const int iy = r / xg_per_row, x0 = (r - iy * xg_per_row) << 2;
        const int e = (c * img + py * ps + iy) * img + x0;                // element index in the image; e/4 = noise group
        float4 px = *reinterpret_cast<const float4*>(image + e);
        if (NOISE) {
            const float4 z = normal4(seed, (uint64_t)sample, (uint32_t)(e >> 2), 0u);
            // explicit fma: the fused path and cgpt_noise_batch must round identically (bit-identical votes)
            px.x = __fmaf_rn(sigma, z.x, px.x); px.y = __fmaf_rn(sigma, z.y, px.y);
            px.z = __fmaf_rn(sigma, z.z, px.z); px.w = __fmaf_rn(sigma, z.w, px.w);
            // keep the fp32 sum as its own value: otherwise hipcc fuses fma + f16 convert into v_fma_mixlo_f16 (ONE
            // rounding), while the reference rounds batch + noise to fp32 first and autocast then casts to fp16.
            asm volatile("" : "+v"(px.x), "+v"(px.y), "+v"(px.z), "+v"(px.w));
        }
        const float vals[4] = {px.x, px.y, px.z, px.w};
        const int col0 = c * ps * ps + iy * ps;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + k;
            const int pxi = x / ps, ix = x - pxi * ps;
            patch_lds[pxi * kp + col0 + ix] = (half_t)vals[k];
        }
    }
    __syncthreads();
    half_t* out = A + ((int64_t)b * pw * pw + (int64_t)py * pw) * lda;   // first of this block's pw rows of A
    if ((kp & 3) == 0 && (lda & 3) == 0) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const int q4 = kp >> 2;
        for (int i = threadIdx.x; i < pw * q4; i += 256) {
            const int pxi = i / q4, q = i - pxi * q4;
            *reinterpret_cast<f16x4*>(out + (int64_t)pxi * lda + q * 4) = *reinterpret_cast<const f16x4*>(patch_lds + pxi * kp + q * 4);
        }
    } else {
        for (int i = threadIdx.x; i < pw * kp; i += 256) {
            const int pxi = i / kp, q = i - pxi * kp;
            out[(int64_t)pxi * lda + q] = patch_lds[pxi * kp + q];
        }
    }
}

__global__ __launch_bounds__(256) void noise_batch_kernel(const float* __restrict__ x, int64_t chw,
                                                          int64_t first_sample, int64_t num, float sigma,
                                                          uint64_t seed, float* __restrict__ out) {
    const int64_t groups = (chw + 3) / 4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= groups * num) return;
    const int64_t b = gid / groups, grp = gid - b * groups;
    const float4 z = normal4(seed, (uint64_t)(first_sample + b), (uint32_t)grp, 0u);
    const float zz[4] = {z.x, z.y, z.z, z.w};
    const int64_t e = grp * 4;
    if (e + 3 < chw && (chw & 3) == 0) {
        const float4 px = *reinterpret_cast<const float4*>(x + e);
        *reinterpret_cast<float4*>(out + b * chw + e) =
            make_float4(__fmaf_rn(sigma, zz[0], px.x), __fmaf_rn(sigma, zz[1], px.y), __fmaf_rn(sigma, zz[2], px.z),
                        __fmaf_rn(sigma, zz[3], px.w));
    } else {
        for (int k = 0; k < 4 && e + k < chw; ++k) out[b * chw + e + k] = __fmaf_rn(sigma, zz[k], x[e + k]);
    }
}

// One step of the random-gradient-free (RGF) black-box attack against the smoothed classifier (BASELINE configs[4]; the
// reference describes it in prose only, README.md:62-64,108-120 -- this update rule is build-side):
//   g[e]   = sum_i coeff[i] * u_{first_dir+i}[e]         (u_s = the N(0,1) draw of sample index s in the noise stream,
//                                                         i.e. exactly the direction cgpt_noise_batch(x, s, 1, delta) added)
//   out[e] = clamp(x_adv[e] + lr * sign(g[e]),  x_clean[e] - eps,  x_clean[e] + eps)
// Separate multiply and add (no contraction), directions in index order: the CPU oracle reproduces the sum bit for bit.
struct RgfCoeffs { float c[CGPT_RGF_MAX_DIRS]; };
__global__ __launch_bounds__(256) void rgf_step_kernel(const float* __restrict__ x_adv, const float* __restrict__ x_clean,
                                                       int64_t chw, int64_t first_dir, int q, RgfCoeffs coeffs, float lr,
                                                       float eps, uint64_t seed, float* __restrict__ out) {
    const int64_t grp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (grp >= (chw + 3) / 4) return;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < q; ++i) {
        const float4 z = normal4(seed, (uint64_t)(first_dir + i), (uint32_t)grp, 0u);
        const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = __fadd_rn(g[k], __fmul_rn(coeffs.c[i], zz[k]));
    }
    const int64_t e = grp * 4;
    for (int k = 0; k < 4 && e + k < chw; ++k) {
        const float sg = g[k] > 0.f ? 1.f : (g[k] < 0.f ? -1.f : 0.f);
        const float c = x_clean[e + k];
        const float v = __fadd_rn(x_adv[e + k], __fmul_rn(lr, sg));
        out[e + k] = fminf(fmaxf(v, c - eps), c + eps);
    }
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                float* __restrict__ resid, int64_t ld, int T, int nb, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb * D) return;
    const int b = i / D, n = i - b * D;
    resid[(int64_t)b * T * ld + n] = cls[n] + pos[n];
}

__global__ void broadcast_rows_kernel(const float* __restrict__ src, int rows, int D, int nb,
                                      float* __restrict__ dst32, int64_t ld32, half_t* __restrict__ dst16, int64_t ld16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nb * rows * D) return;
    const int n = (int)(i % D);
    const int64_t br = i / D;                     // b*rows + r
    const float v = src[(br % rows) * D + n];
    dst32[br * ld32 + n] = v;
    dst16[br * ld16 + n] = (half_t)v;
}

__global__ void mean_rows_kernel(const float* __restrict__ src, int64_t lds, int rows, int D, int nb,
                                 half_t* __restrict__ dst16, int64_t ld16) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb * D) return;
    const int b = i / D, n = i - b * D;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += src[((int64_t)b * rows + r) * lds + n];
    dst16[(int64_t)b * ld16 + n] = (half_t)(s / (float)rows);
}

// ------------------------------------------------------------------------------- argmax + vote histogram
// One wave per sample.  Each lane scans classes lane, lane+64, ... keeping the first maximum (strict >),
// then a 6-step xor-shuffle butterfly keeps (larger value, then smaller index): ndarray/tensor argmax semantics
// "first maximal index" (smoothing.py:97).  Lane 0 adds the vote with one 64-bit atomic (smoothing.py:98,101-105).
// per > 0: image-major batch of several images (see im2col_kernel): image i = s / per votes into counts + i*2K (first
// range) or counts_b + i*2K (second range): a [images, 2, K] table when counts_b = counts + K.
__global__ __launch_bounds__(256) void vote_kernel(const float* __restrict__ logits, int64_t ld, int64_t num, int K,
                                                   unsigned long long* __restrict__ counts, int64_t na,
                                                   unsigned long long* __restrict__ counts_b, int per, int64_t row0) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= num) return;
    const float* row = logits + s * ld;
    // torch.argmax / numpy argmax semantics (smoothing.py:97): a NaN ranks above every number, ties go to the lower index
    auto beats = [](float v, int i, float bv, int bi) {
        const bool vn = v != v, bn = bv != bv;
        if (vn != bn) return vn;
        if (vn) return i < bi;
        return v > bv || (v == bv && i < bi);
    };
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
        const float v = row[k];
        if (bi == 0x7fffffff || beats(v, k, best, bi)) { best = v; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (beats(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    const int64_t rs = row0 + s;
    const int64_t im = per > 0 ? rs / per : 0, j = per > 0 ? rs - im * per : s;
    if (lane == 0) atomicAdd((j < na ? counts : counts_b) + im * 2 * K + bi, 1ull);   // rows >= na belong to the second range
}


// ------------------------------------------------------------------- device-side finalisation of certify / predict
// One wave.  Smooth.certify lines 46-56 (smoothing.py): cAHat = first maximal index of the selection histogram
// (64 lanes scan K/64 classes each, then a 6-step xor-shuffle butterfly keeps (larger count, then smaller index)),
// nA = estimation[cAHat], Clopper-Pearson lower bound and sigma * Phi^-1 in float64 by lane 0 (stats.h, the same
// code the host path runs).  Smooth.predict lines 73-79: top-2 counts by two butterflies, two-sided binomial test.
// out[0] = label (as double), out[1] = radius (certify) or p-value (predict).
__global__ __launch_bounds__(64) void finalize_kernel(const long long* __restrict__ csel, const long long* __restrict__ cest,
                                                      int K, long long n, double alpha, double sigma, int predict,
                                                      double* __restrict__ out) {
    const int lane = threadIdx.x;
    auto wave_argmax = [&](const long long* c, int skip) {
        long long best = -1;
        int bi = 0x7fffffff;
        for (int k = lane; k < K; k += 64) {
            if (k == skip) continue;
            const long long v = c[k];
            if (v > best) { best = v; bi = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        return bi;
    };
    if (!predict) {
        const int cAHat = wave_argmax(csel, -1);
        if (lane == 0) {
            const long long nA = cest[cAHat];
            const double pABar = cgpt_stats::cp_lower_bound(nA, n, alpha);
            if (pABar < 0.5) { out[0] = -1.0; out[1] = 0.0; }
            else { out[0] = (double)cAHat; out[1] = sigma * cgpt_stats::norm_ppf(pABar); }
        }
    } else {
        const int i1 = wave_argmax(cest, -1);
        const int i2 = wave_argmax(cest, i1);
        if (lane == 0) {
            const long long c1 = cest[i1], c2 = cest[i2];
            const double pv = cgpt_stats::binom_test_two_sided(c1, c1 + c2, 0.5);
            out[0] = pv > alpha ? -1.0 : (double)i1;
            out[1] = pv;
        }
    }
}

// ------------------------------------------------------------------------------------- fills and casts
__global__ void fill_normal_kernel(void* dst, int is_f16, int64_t rows, int64_t cols, int64_t ld, float mean,
                                   float stdv, uint64_t seed, uint64_t tensor_id) {
    const int64_t groups = (rows * cols + 3) / 4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= groups) return;
    const float4 z = normal4(seed, tensor_id, (uint32_t)gid, 1u + (uint32_t)(gid >> 32));
    const float zz[4] = {z.x, z.y, z.z, z.w};
    for (int k = 0; k < 4; ++k) {
        const int64_t e = gid * 4 + k;
        if (e >= rows * cols) break;
        const int64_t r = e / cols, c = e - r * cols;
        const float v = mean + stdv * zz[k];
        if (is_f16) reinterpret_cast<half_t*>(dst)[r * ld + c] = (half_t)v;
        else reinterpret_cast<float*>(dst)[r * ld + c] = v;
    }
}

__global__ void fill_const_kernel(float* dst, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

__global__ void f32_to_f16_kernel(const float* src, int64_t lds, half_t* dst, int64_t ldd, int64_t rows, int64_t cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols, c = i - r * cols;
    dst[r * ldd + c] = (half_t)src[r * lds + c];
}

__global__ void f16_to_f32_kernel(const half_t* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols, c = i - r * cols;
    dst[r * ldd + c] = (float)src[r * lds + c];
}

inline unsigned blocks_for(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

hipError_t launch_add_delta(float* x, int64_t ldx, const half_t* delta, int64_t ldd, int64_t rows, int D,
                            hipStream_t stream, int skip_mod) {
    if (rows <= 0) return hipSuccess;
    if ((D & 1) || (ldx & 1) || (ldd & 1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_delta_kernel, dim3(blocks_for(rows * (D >> 1))), dim3(256), 0, stream, x, ldx, delta, ldd, rows, D, skip_mod);
    return hipGetLastError();
}

hipError_t launch_layernorm(float* x, int64_t ldx, const half_t* delta, int64_t ldd, const float* gamma,
                            const float* beta, float eps, half_t* y16, int64_t ldy16, float* y32, int64_t ldy32,
                            int64_t rows, int D, hipStream_t stream, const half_t* delta2, int64_t ldd2, int keep_x) {
    if (rows <= 0) return hipSuccess;
    if (D <= 0 || (D & 1) || D > 32 * 128 || (ldx & 1) || (ldy16 & 1) || (ldy32 & 1) || (ldd & 1) || (ldd2 & 1)) return hipErrorInvalidValue;
    // 16-byte path: D a multiple of 128 (1408 = 11 x 128, 768 = 6 x 128, 4096) and 16-byte aligned rows
    if ((D % 128) == 0 && D / 128 <= 32 && (ldx % 4) == 0 && (ldd % 4) == 0 && (ldd2 % 4) == 0 && (ldy16 % 4) == 0 && (ldy32 % 4) == 0) {
        dim3 grid4((unsigned)((rows + 7) / 8)), block4(256);
#define CGPT_LN4(NCH) hipLaunchKernelGGL(layernorm4_kernel<NCH>, grid4, block4, 0, stream, x, ldx, delta, ldd, gamma, beta, eps, \
                                         y16, ldy16, y32, ldy32, rows, D, delta2, ldd2, keep_x)
        const int n = D / 128;
        if (n <= 1) CGPT_LN4(1);
        else if (n <= 6) { if (n == 6) CGPT_LN4(6); else goto generic; }
        else if (n == 11) hipLaunchKernelGGL(layernorm4_vitg_kernel, grid4, block4, 0, stream, x, ldx, delta, ldd, gamma, beta, eps,
                                             y16, ldy16, y32, ldy32, rows, D, delta2, ldd2, keep_x);
        else if (n == 32) CGPT_LN4(32);
        else goto generic;
#undef CGPT_LN4
        return hipGetLastError();
    }
generic:
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define CGPT_LN(NCH) hipLaunchKernelGGL(layernorm_kernel<NCH>, grid, block, 0, stream, x, ldx, delta, ldd, gamma, beta, eps, \
                                        y16, ldy16, y32, ldy32, rows, D, delta2, ldd2, keep_x)
    if (D <= 2 * 128) CGPT_LN(2);
    else if (D <= 6 * 128) CGPT_LN(6);
    else if (D <= 11 * 128) CGPT_LN(11);
    else if (D <= 16 * 128) CGPT_LN(16);
    else CGPT_LN(32);
#undef CGPT_LN
    return hipGetLastError();
}

hipError_t launch_noise_im2col(const float* x, int img, int ps, int64_t first_sample, int na, int64_t first_b, int nb,
                               float sigma, uint64_t seed, half_t* A, int64_t lda, hipStream_t stream, int per,
                               int64_t img_stride, int64_t row0) {
    if (nb <= 0) return hipSuccess;
    if ((img & 3) || (img % ps)) return hipErrorInvalidValue;
    const int pw = img / ps;
    hipLaunchKernelGGL(im2col_kernel<true>, dim3((unsigned)(nb * pw)), dim3(256), (size_t)pw * 3 * ps * ps * sizeof(half_t), stream, x,
                       img, ps, first_sample, na, first_b, nb, sigma, seed, A, lda, per, img_stride, row0);
    return hipGetLastError();
}

hipError_t launch_im2col(const float* images, int img, int ps, int nb, half_t* A, int64_t lda, hipStream_t stream) {
    if (nb <= 0) return hipSuccess;
    if ((img & 3) || (img % ps)) return hipErrorInvalidValue;
    const int pw = img / ps;
    hipLaunchKernelGGL(im2col_kernel<false>, dim3((unsigned)(nb * pw)), dim3(256), (size_t)pw * 3 * ps * ps * sizeof(half_t), stream,
                       images, img, ps, (int64_t)0, nb, (int64_t)0, nb, 0.0f, (uint64_t)0, A, lda, 0, (int64_t)0, (int64_t)0);
    return hipGetLastError();
}

hipError_t launch_noise_batch(const float* x, int64_t chw, int64_t first_sample, int64_t num, float sigma,
                              uint64_t seed, float* out, hipStream_t stream) {
    if (num <= 0) return hipSuccess;
    const int64_t n = ((chw + 3) / 4) * num;
    hipLaunchKernelGGL(noise_batch_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, x, chw, first_sample, num, sigma,
                       seed, out);
    return hipGetLastError();
}

hipError_t launch_rgf_step(const float* x_adv, const float* x_clean, int64_t chw, int64_t first_dir, int q, const float* coeffs,
                           float lr, float eps, uint64_t seed, float* out, hipStream_t stream) {
    if (q < 0 || q > CGPT_RGF_MAX_DIRS) return hipErrorInvalidValue;
    RgfCoeffs c;
    for (int i = 0; i < CGPT_RGF_MAX_DIRS; ++i) c.c[i] = i < q ? coeffs[i] : 0.f;
    hipLaunchKernelGGL(rgf_step_kernel, dim3(blocks_for((chw + 3) / 4)), dim3(256), 0, stream, x_adv, x_clean, chw, first_dir, q,
                       c, lr, eps, seed, out);
    return hipGetLastError();
}

hipError_t launch_cls_rows(const float* cls, const float* pos, float* resid, int64_t ld, int T, int nb, int D,
                           hipStream_t stream) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3(blocks_for((int64_t)nb * D)), dim3(256), 0, stream, cls, pos, resid, ld, T,
                       nb, D);
    return hipGetLastError();
}

hipError_t launch_broadcast_rows(const float* src, int rows, int D, int nb, float* dst32, int64_t ld32, half_t* dst16,
                                 int64_t ld16, hipStream_t stream) {
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3(blocks_for((int64_t)nb * rows * D)), dim3(256), 0, stream, src, rows,
                       D, nb, dst32, ld32, dst16, ld16);
    return hipGetLastError();
}

hipError_t launch_mean_rows(const float* src, int64_t lds, int rows, int D, int nb, half_t* dst16, int64_t ld16,
                            hipStream_t stream) {
    hipLaunchKernelGGL(mean_rows_kernel, dim3(blocks_for((int64_t)nb * D)), dim3(256), 0, stream, src, lds, rows, D, nb,
                       dst16, ld16);
    return hipGetLastError();
}

hipError_t launch_vote(const float* logits, int64_t ld, int64_t num, int K, int64_t* counts, int64_t na, int64_t* counts_b,
                       hipStream_t stream, int per, int64_t row0) {
    if (num <= 0) return hipSuccess;
    hipLaunchKernelGGL(vote_kernel, dim3((unsigned)((num + 3) / 4)), dim3(256), 0, stream, logits, ld, num, K,
                       reinterpret_cast<unsigned long long*>(counts), na, reinterpret_cast<unsigned long long*>(counts_b), per, row0);
    return hipGetLastError();
}

hipError_t launch_finalize(const int64_t* csel, const int64_t* cest, int K, int64_t n, double alpha, double sigma, int predict,
                           double* out, hipStream_t stream) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const long long*>(csel),
                       reinterpret_cast<const long long*>(cest), K, (long long)n, alpha, sigma, predict, out);
    return hipGetLastError();
}

hipError_t launch_fill_normal(void* dst, int is_f16, int64_t rows, int64_t cols, int64_t ld, float mean, float stdv,
                              uint64_t seed, uint64_t tensor_id, hipStream_t stream) {
    const int64_t groups = (rows * cols + 3) / 4;
    if (groups <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_normal_kernel, dim3(blocks_for(groups)), dim3(256), 0, stream, dst, is_f16, rows, cols, ld,
                       mean, stdv, seed, tensor_id);
    return hipGetLastError();
}

hipError_t launch_fill_const(float* dst, int64_t n, float v, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_const_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, dst, n, v);
    return hipGetLastError();
}

hipError_t launch_f32_to_f16(const float* src, int64_t lds, half_t* dst, int64_t ldd, int64_t rows, int64_t cols,
                             hipStream_t stream) {
    if (rows * cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(blocks_for(rows * cols)), dim3(256), 0, stream, src, lds, dst, ldd, rows,
                       cols);
    return hipGetLastError();
}

hipError_t launch_f16_to_f32(const half_t* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols,
                             hipStream_t stream) {
    if (rows * cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3(blocks_for(rows * cols)), dim3(256), 0, stream, src, lds, dst, ldd, rows,
                       cols);
    return hipGetLastError();
}

}  // namespace cgpt
