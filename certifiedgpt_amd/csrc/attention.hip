// attention.hip -- fused softmax(scale * Q K^T) V for short sequences (T = 257 ViT tokens, 32 Q-Former queries).
//
// Reference ops replaced: Attention.forward eva_vit.py:133-150 (16 heads x head_dim 88, no mask, no rel-pos bias for
// ViT-G), BertSelfAttention.forward Qformer.py:231-266 (12 heads x 64; self 32x32 and cross 32x257; the additive
// masks are identically zero, Qformer.py:798-801).
//
// One workgroup per (sample, head).  The whole K and V of the head live in LDS (T <= 288 rows), so there is no
// online softmax: a wave computes S^T = K . Q^T for ALL keys of a 16-query tile with v_mfma_f32_16x16x32_f16
// (operands swapped so that a query's scores sit in one lane's registers), takes the row max / sum with two
// cross-lane shuffles, and feeds exp() of the scores -- still in registers -- as the A operand of the P.V MFMAs:
//   S^T tile kt (keys 16kt..16kt+15):  lane l, reg r  =  score(query l&15, key 16kt + 4*(l>>4) + r)
//   P.V A-fragment for the key pair (2u, 2u+1): element j<4 <- tile 2u reg j, j>=4 <- tile 2u+1 reg j-4,
//   so MFMA k-slot 8g+j stands for key 32u + 4g + j (j<4) or 32u + 16 + 4g + (j-4); the matching V B-fragment
//   (rows = those keys, col = d) is fetched from the ROW-major LDS image of V by two ds_read_b64_tr_b16
//   (4 rows x 16 columns per 16-lane group, transposed in hardware).
// LDS row strides: K 208 B (13 x 16 B: the 16 rows of a ds_read_b128 fragment land on distinct 16-B slots),
// V 224 B / 160 B (the 8 rows a 32-lane half reads by ds_read_b64_tr_b16 land on distinct 32-B bank ranges).
#include "kernels.h"

namespace cgpt {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

namespace {

__device__ __forceinline__ f16x4 lds_read_tr16(const half_t* p) {
    fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(p));
    return __builtin_bit_cast(f16x4, v);
}

template <int DPAD> struct AttnLayout {
    static constexpr int KSTR = DPAD + 8;                   // halfs
    static constexpr int VSTR = (DPAD == 96) ? 112 : 80;    // halfs
};

// HD: head_dim (88 | 64); DPAD: HD rounded up to 32; NKT: number of 16-key tiles held (keys padded to NKT*16).
template <int HD, int DPAD, int NKT>
__global__ __launch_bounds__(512) void attention_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int KSTR = AttnLayout<DPAD>::KSTR, VSTR = AttnLayout<DPAD>::VSTR;
    constexpr int TKP = NKT * 16;
    constexpr int CH = DPAD / 8;
    half_t* Ks = reinterpret_cast<half_t*>(smem_raw);
    half_t* Vs = Ks + TKP * KSTR;

    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int r15 = lane & 15, g = lane >> 4;

    // ---- stage K and V of this (sample, head): zero-fill d >= HD and keys >= Tk
    const half_t* Kg = p.K + (int64_t)b * p.kv_batch_stride + h * HD;
    const half_t* Vg = p.V + (int64_t)b * p.kv_batch_stride + h * HD;
    for (int idx = tid; idx < TKP * CH; idx += blockDim.x) {
        const int row = idx / CH, ch = idx - row * CH;
        f16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < p.Tk && ch * 8 < HD) {
            kv = *reinterpret_cast<const f16x8*>(Kg + (int64_t)row * p.ldk + ch * 8);
            vv = *reinterpret_cast<const f16x8*>(Vg + (int64_t)row * p.ldv + ch * 8);
        }
        *reinterpret_cast<f16x8*>(Ks + row * KSTR + ch * 8) = kv;
        *reinterpret_cast<f16x8*>(Vs + row * VSTR + ch * 8) = vv;
    }
    __syncthreads();

    const float sl2 = p.scale * 1.44269504088896340736f;   // softmax(scale*s) = 2^((s - max) * scale * log2 e) / sum
    const int nqt = (p.Tq + 15) >> 4;
    const half_t* Qb = p.Q + (int64_t)b * p.q_batch_stride + h * HD;
    half_t* Ob = p.O + (int64_t)b * p.o_batch_stride + h * HD;

    for (int qt = wave; qt < nqt; qt += nwaves) {
        // K/V fragments do not depend on the query tile: without this opaque zero the compiler hoists all
        // 54 + 108 LDS reads out of the loop and spills.  (cdna guide section 5.7 item 3)
        int opq = 0;
        asm volatile("" : "+v"(opq));
        const half_t* Kq = Ks + opq;
        const half_t* Vq = Vs + opq;
        // Q^T B-fragments: lane holds Q[query r15][d = 32*ds + 8*g .. +7]
        const int qrow = min(qt * 16 + r15, p.Tq - 1);
        f16x8 qf[DPAD / 32];
#pragma unroll
        for (int ds = 0; ds < DPAD / 32; ++ds) {
            const int d = ds * 32 + g * 8;
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (d < HD) v = *reinterpret_cast<const f16x8*>(Qb + (int64_t)qrow * p.ldq + d);
            qf[ds] = v;
        }
        // S^T = K . Q^T
        f32x4 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < DPAD / 32; ++ds) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(Kq + (kt * 16 + r15) * KSTR + ds * 32 + g * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ds], acc, 0, 0, 0);
            }
            s[kt] = acc;
        }
        // softmax over keys (all keys of a query: this lane's registers x the 4 lanes sharing r15)
        float mx = -1e30f;
        const int klim = p.Tk - 4 * g + opq;          // key (kt*16 + 4g + r) is padding iff kt*16 + r >= klim
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt * 16 + 16 > p.Tk) {                // wave-uniform: only the boundary tiles pay for the mask
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + r >= klim) s[kt][r] = -1e30f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp2f((s[kt][r] - mx) * sl2);
                s[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);

        // O = P . V
        f32x4 o[DPAD / 16];
#pragma unroll
        for (int dt = 0; dt < DPAD / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const half_t* vbase = Vq + (4 * g + (r15 >> 2)) * VSTR + 4 * (r15 & 3);
#pragma unroll
        for (int u = 0; u < NKT / 2; ++u) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf[j] = (half_t)s[2 * u][j];
                pf[4 + j] = (half_t)s[2 * u + 1][j];
            }
#pragma unroll
            for (int dt = 0; dt < DPAD / 16; ++dt) {
                const half_t* a1 = vbase + (32 * u) * VSTR + dt * 16;
                const f16x4 v1 = lds_read_tr16(a1);
                const f16x4 v2 = lds_read_tr16(a1 + 16 * VSTR);
                const f16x8 vf = {v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vf, o[dt], 0, 0, 0);
            }
        }
        // O tile: col d = lane&15, row (query) = 4g + r.  1/sum of query q sits in lanes with r15 == q.
        const float inv = 1.0f / sum;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float invr = __shfl(inv, 4 * g + r);
            const int q = qt * 16 + 4 * g + r;
            if (q < p.Tq) {
#pragma unroll
                for (int dt = 0; dt < DPAD / 16; ++dt) {
                    const int d = dt * 16 + r15;
                    if (d < HD) Ob[(int64_t)q * p.ldo + d] = (half_t)(o[dt][r] * invr);
                }
            }
        }
    }
}

template <int HD, int DPAD, int NKT>
hipError_t launch_one(const AttnParams& p, hipStream_t stream) {
    constexpr int lds_bytes = NKT * 16 * (AttnLayout<DPAD>::KSTR + AttnLayout<DPAD>::VSTR) * (int)sizeof(half_t);
    static bool configured = false;   // per instantiation; the attribute is idempotent
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<HD, DPAD, NKT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int nqt = (p.Tq + 15) / 16;
    const int waves = nqt >= 8 ? 8 : (nqt >= 4 ? 4 : (nqt >= 2 ? 2 : 1));
    dim3 grid(p.heads, p.B), block(64 * waves);
    hipLaunchKernelGGL((attention_kernel<HD, DPAD, NKT>), grid, block, lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_attention(const AttnParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.heads <= 0 || p.Tq <= 0 || p.Tk <= 0) return hipErrorInvalidValue;
    if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8)) return hipErrorInvalidValue;   // 16-byte row alignment
    const bool small = p.Tk <= 32;
    if (p.Tk > 288) return hipErrorInvalidValue;   // whole-K/V-in-LDS design: T <= 288 (224^2 images; 448^2 is "next")
    if (p.head_dim == 88) return small ? launch_one<88, 96, 2>(p, stream) : launch_one<88, 96, 18>(p, stream);
    if (p.head_dim == 64) return small ? launch_one<64, 64, 2>(p, stream) : launch_one<64, 64, 18>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace cgpt
