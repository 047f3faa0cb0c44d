// gemm.hip -- fp16 MFMA GEMM with fused epilogues for the ViT / Q-Former linears.
//
// Reference ops replaced (graphs/models/minigpt4/models/): PatchEmbed conv-as-GEMM eva_vit.py:202,209;
// Attention.qkv/proj eva_vit.py:129,151; Mlp.fc1/GELU/fc2 eva_vit.py:59-66; the residual adds eva_vit.py:180-181;
// BertSelfAttention/BertSelfOutput/BertIntermediate/BertOutput denses Qformer.py:185-188,285-289,349-375;
// llama_proj minigpt4.py:76-78,141.
//
// Layout: C[M,N] = A[M,K] * W[N,K]^T.  Both operands are K-contiguous ("B^T input"), which is exactly how
// nn.Linear stores its weight, so the MFMA A and B fragments are both plain 16-byte row reads.
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 tiles of v_mfma_f32_16x16x32_f16.
// LDS tiles are [128 rows][64 halfs] with the 16-byte chunk index XOR-swizzled by (row>>1)&7, which makes the
// ds_read_b128 fragment reads conflict-free for the gfx950 lane groups (MI355X guide, LDS section) and keeps
// each row's 128 bytes in place so the ds_write_b128 staging writes stay conflict-free too.
// Pipeline: register-staged double buffer -- global loads of tile k+1 are issued before the MFMAs of tile k
// and written to the other LDS buffer after them; one barrier per K-tile.
// Block -> tile map: XCD-contiguous chunks (blocks b and b+8 share an XCD) and, inside a chunk, groups of 8
// tile-rows walked column-major so that the 64 tiles resident on an XCD share 8 A-strips and 8 W-strips in L2.
#include "kernels.h"

namespace cgpt {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GROUP_M = 8;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) half_t smem[2 * 2 * BM * BK];  // [buf][A|W][128*64] = 64 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r15 = lane & 15, g = lane >> 4;

    // ---- block -> tile (speed only; any map is correct)
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.N + BN - 1) / BN;
    int t;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_group = GROUP_M * tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_grp = t - grp * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;

    // ---- staging: thread -> (row tid/8 + 32*i, 16-byte chunk tid%8)
    const int ld_row = tid >> 3, ld_chunk = tid & 7;
    const half_t* a_src = p.A + (int64_t)(tm * BM + ld_row) * p.lda + ld_chunk * 8;
    const half_t* w_src = p.W + (int64_t)(tn * BN + ld_row) * p.ldw + ld_chunk * 8;
    const int64_t a_step = 32 * p.lda, w_step = 32 * p.ldw;
    const int wr_off = ld_row * BK + ((ld_chunk ^ ((ld_row >> 1) & 7)) << 3);  // +32 rows keeps the swizzle term

    // ---- fragment reads: row r15 of sub-tile i, chunk ks*4+g  (the swizzle term does not depend on i)
    const int sw = (r15 >> 1) & 7;
    const int k_off0 = ((g ^ sw) << 3);           // ks = 0
    const int k_off1 = (((4 + g) ^ sw) << 3);     // ks = 1
    const int a_rd = (wm * 64 + r15) * BK;
    const int b_rd = (wn * 64 + r15) * BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f16x8 ra[4], rb[4];
    const int nk = p.K / BK;

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const f16x8*>(a_src + i * a_step);
        rb[i] = *reinterpret_cast<const f16x8*>(w_src + i * w_step);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f16x8*>(smem + wr_off + i * 32 * BK) = ra[i];
        *reinterpret_cast<f16x8*>(smem + BM * BK + wr_off + i * 32 * BK) = rb[i];
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const half_t* sa = smem + (kt & 1) * (2 * BM * BK);
        const half_t* sb = sa + BM * BK;
        const bool more = (kt + 1 < nk);
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const f16x8*>(a_src + i * a_step + (kt + 1) * BK);
                rb[i] = *reinterpret_cast<const f16x8*>(w_src + i * w_step + (kt + 1) * BK);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ko = ks ? k_off1 : k_off0;
            f16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f16x8*>(sa + a_rd + i * 16 * BK + ko);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(sb + b_rd + j * 16 * BK + ko);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            half_t* da = smem + ((kt + 1) & 1) * (2 * BM * BK);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f16x8*>(da + wr_off + i * 32 * BK) = ra[i];
                *reinterpret_cast<f16x8*>(da + BM * BK + wr_off + i * 32 * BK) = rb[i];
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D map of v_mfma_f32_16x16x32: col = lane&15, row = 4*(lane>>4) + reg.
    const int m_base = tm * BM + wm * 64 + 4 * g;
    const int n_base = tn * BN + wn * 64 + r15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n_base + j * 16;
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + i * 16 + r;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bias;
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
        }
    }
}

}  // namespace

hipError_t launch_gemm(int epilogue, const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0 || (p.lda % 8) != 0 || (p.ldw % 8) != 0)
        return hipErrorInvalidValue;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    dim3 grid(tiles), block(256);
    switch (epilogue) {
        case EPI_F16: hipLaunchKernelGGL(gemm_f16_kernel<EPI_F16>, grid, block, 0, stream, p); break;
        case EPI_F16_GELU: hipLaunchKernelGGL(gemm_f16_kernel<EPI_F16_GELU>, grid, block, 0, stream, p); break;
        case EPI_F32: hipLaunchKernelGGL(gemm_f16_kernel<EPI_F32>, grid, block, 0, stream, p); break;
        case EPI_RESID: hipLaunchKernelGGL(gemm_f16_kernel<EPI_RESID>, grid, block, 0, stream, p); break;
        case EPI_PATCH: hipLaunchKernelGGL(gemm_f16_kernel<EPI_PATCH>, grid, block, 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace cgpt
