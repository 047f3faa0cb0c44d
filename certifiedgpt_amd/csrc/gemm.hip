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
#include <stdlib.h>

#include "gemm_common.h"

namespace cgpt {

namespace {

// Per-device launch state (a process may drive several GPUs: one handle per device): CU count and "dynamic LDS size
// configured" flags are cached per device ordinal, not per process.
constexpr int kMaxDevices = 64;
struct DeviceInfo { int dev; int num_cus; };
inline hipError_t device_info(DeviceInfo& out) {
    static int cus[kMaxDevices] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    if (cus[dev] == 0) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        cus[dev] = n > 0 ? n : 256;
    }
    out.dev = dev; out.num_cus = cus[dev];
    return hipSuccess;
}
template <typename KernelT>
inline hipError_t configure_lds(KernelT kernel, int lds_bytes, bool (&done)[kMaxDevices], int dev) {
    if (done[dev]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

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
    int tm, tn;
    tile_of_block(tiles_m, tiles_n, tm, tn);

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
                epilogue_store<EPI>(p, m, n, acc[i][j][r] + bias);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------ v2
// 256 x BN x 64 tile, 8 waves, operands staged straight into LDS by global_load_lds_dwordx4 (no VGPR round
// trip, no ds_write: the register-staged v1 kernel is bound by the ~79 B/clk ds_write_b128 path).
// One wave-instruction writes 1 KiB = 8 tile rows x 128 B, lane l -> row l>>3, 16-byte slot l&7; the XOR swizzle
// therefore goes on the per-lane SOURCE address (slot c' of row r receives global chunk c' ^ ((r>>1)&7)) and the
// fragment reads apply the same involution (cdna guide section 5.4 rule 21).
// Two LDS stages; the loads of K-tile t+1 are issued before the MFMAs of K-tile t and retired by the
// vmcnt(0) + barrier that ends the iteration.
//   BN = 256: waves 2(M) x 4(N), wave tile 128x64 (acc 128 VGPRs), 12 ds_read_b128 per 32 MFMAs, LDS 128 KiB
//   BN = 128: waves 4(M) x 2(N), wave tile  64x64 (acc  64 VGPRs), 16 ds_read_b128 per 32 MFMAs, LDS  96 KiB
template <int EPI, int BN_>
__global__ __launch_bounds__(512, 2) void gemm2_f16_kernel(GemmParams p) {
    constexpr int BM2 = 256;
    constexpr int WN = (BN_ == 256) ? 4 : 2, WM = 8 / WN;
    constexpr int TM = BM2 / WM / 16, TN = BN_ / WN / 16;
    constexpr int A_ELEMS = BM2 * BK, B_ELEMS = BN_ * BK, STAGE = A_ELEMS + B_ELEMS;
    constexpr int A_INSTR = BM2 / 64, B_INSTR = BN_ / 64;          // 1-KiB glds pieces per wave per stage
    extern __shared__ __attribute__((aligned(16))) half_t smem2[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int r15 = lane & 15, g = lane >> 4;

    const int tiles_m = (p.M + BM2 - 1) / BM2;
    const int tiles_n = (p.N + BN_ - 1) / BN_;
    const int ntiles = tiles_m * tiles_n;

    // ---- staging sources: piece i of this wave covers tile rows wave*(rows/8) + i*8 .. +7
    const int lr = lane >> 3, cpos = lane & 7;
    const half_t* a_src[A_INSTR];
    const half_t* b_src[B_INSTR];
    int tm = 0, tn = 0;
    auto set_tile = [&](int t) {
        tile_of_virtual_block(t, ntiles, tiles_m, tiles_n, tm, tn);
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int r = wave * (BM2 / 8) + i * 8 + lr;
            a_src[i] = p.A + (int64_t)(tm * BM2 + r) * p.lda + ((cpos ^ ((r >> 1) & 7)) << 3);
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int r = wave * (BN_ / 8) + i * 8 + lr;
            b_src[i] = p.W + (int64_t)(tn * BN_ + r) * p.ldw + ((cpos ^ ((r >> 1) & 7)) << 3);
        }
    };
    auto stage_load = [&](int stage, int kt) {
        half_t* sa = smem2 + stage * STAGE + wave * (BM2 / 8) * BK;
        half_t* sb = smem2 + stage * STAGE + A_ELEMS + wave * (BN_ / 8) * BK;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(sa + i * 8 * BK), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(sb + i * 8 * BK), 16, 0, 0);
    };

    const int sw = (r15 >> 1) & 7;
    const int k_off0 = ((g ^ sw) << 3), k_off1 = (((4 + g) ^ sw) << 3);
    const int a_rd = (wr * (BM2 / WM) + r15) * BK;
    const int b_rd = A_ELEMS + (wc * (BN_ / WN) + r15) * BK;

    // acc[i][j] holds the TRANSPOSED 16x16 tile (operands swapped: W fragment first), so that
    // acc[i][j][r] = C[m = i*16 + (lane&15)][n = j*16 + 4*(lane>>4) + r]: a lane owns 4 CONSECUTIVE columns of one row
    // and the epilogue issues 8/16-byte vector stores instead of four 2/4-byte ones.
    f32x4 acc[TM][TN];
    constexpr int HM = TM / 2;   // the wave tile is walked in two M-halves so that fragment reads run one group ahead
    f16x8 a0[HM], a1[HM], b0[TN], b1[TN];
#define CGPT_LDA(dst, st_, ko_, h_)                                                                        \
    _Pragma("unroll") for (int i = 0; i < HM; ++i)                                                          \
        dst[i] = *reinterpret_cast<const f16x8*>((st_) + a_rd + ((h_) * HM + i) * 16 * BK + (ko_));
#define CGPT_LDB(dst, st_, ko_)                                                                             \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                          \
        dst[j] = *reinterpret_cast<const f16x8*>((st_) + b_rd + j * 16 * BK + (ko_));
#ifdef CGPT_SETPRIO
#define CGPT_PRIO(x) __builtin_amdgcn_s_setprio(x);
#else
#define CGPT_PRIO(x)
#endif
#define CGPT_MM_ROWS(af_, bf_, h_, i0_, i1_)                                                                \
    CGPT_PRIO(1)                                                                                            \
    _Pragma("unroll") for (int i = (i0_); i < (i1_); ++i)                                                   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                      \
            acc[(h_) * HM + i][j] =                                                                         \
                __builtin_amdgcn_mfma_f32_16x16x32_f16(bf_[j], af_[i], acc[(h_) * HM + i][j], 0, 0, 0);      \
    CGPT_PRIO(0)
#define CGPT_FENCE __builtin_amdgcn_sched_barrier(0);

    // Schedule of one K-tile (4 groups of HM*TN MFMAs).  Each group starts with one row of MFMAs, THEN issues the
    // fragment reads of the next group, then runs its remaining MFMAs: the compiler-inserted lgkmcnt wait in front of a
    // group is therefore exact (no younger read outstanding) and the reads get >= (HM-1)*TN MFMAs of cover.
    // Persistent: this workgroup walks tiles t = blockIdx.x, blockIdx.x + gridDim.x, ...  `c` counts K-tiles across the
    // whole walk; K-tile c lives in LDS stage c & 1, so the first K-tile of the NEXT output tile can be requested before
    // this tile's epilogue (its stage was last read two K-tiles ago) and lands while the epilogue stores drain.
    const int nk = p.K / BK;
    int c = 0;
    int t = blockIdx.x;
#ifdef CGPT_STAMPS
    unsigned long long stamp_vm = 0, stamp_bar = 0, stamp_epi = 0, stamp_ld = 0;
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memtime();
#endif
    if (t < ntiles) { set_tile(t); stage_load(0, 0); }
    for (; t < ntiles; t += gridDim.x) {
    f32x4 bias4[TN];                                    // this lane's 4 x TN bias values; loaded now, used in the epilogue
    {
        const int nb0 = tn * BN_ + wc * (BN_ / WN) + 4 * g;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && nb0 + j * 16 < p.N) bias4[j] = *reinterpret_cast<const f32x4*>(p.bias + nb0 + j * 16);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    CGPT_LDA(a0, smem2 + (c & 1) * STAGE, k_off0, 0)
    CGPT_LDB(b0, smem2 + (c & 1) * STAGE, k_off0)
    for (int kt = 0; kt < nk; ++kt, ++c) {
        const half_t* st = smem2 + (c & 1) * STAGE;
        const half_t* nx = smem2 + ((c + 1) & 1) * STAGE;
        const bool more = kt + 1 < nk;
        // K-tile kt+1 -> the other stage: every wave finished reading it before the barrier that ended iteration kt-1
#ifndef CGPT_STAGGER
#define CGPT_STAGGER 0
#endif
        // LDS-DMA issue is expensive (60-185 cycles per 1-KiB piece, MI355X guide): the two waves of a SIMD (w, w+4)
        // issue their 8 pieces at different points of the K-tile so that one of them always feeds the matrix pipe.
#ifdef CGPT_STAMPS
        const unsigned long long tl0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        if (more && !(p.ablate & 1) && (CGPT_STAGGER == 0 || wave < 4)) stage_load((c + 1) & 1, kt + 1);
#ifdef CGPT_STAMPS
        stamp_ld += __builtin_amdgcn_s_memtime() - tl0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        CGPT_FENCE
        CGPT_MM_ROWS(a0, b0, 0, 0, 1)            // G1: k-step 0, M-half 0
        CGPT_FENCE
        CGPT_LDA(a1, st, k_off0, 1)
        CGPT_FENCE
        CGPT_MM_ROWS(a0, b0, 0, 1, HM)
        CGPT_FENCE
        if (CGPT_STAGGER == 1 && more && !(p.ablate & 1) && wave >= 4) stage_load((c + 1) & 1, kt + 1);
        CGPT_FENCE
        CGPT_MM_ROWS(a1, b0, 1, 0, 1)            // G2: k-step 0, M-half 1
        CGPT_FENCE
        CGPT_LDA(a0, st, k_off1, 0)
        CGPT_LDB(b1, st, k_off1)
        CGPT_FENCE
        CGPT_MM_ROWS(a1, b0, 1, 1, HM)
        CGPT_FENCE
        if (CGPT_STAGGER == 2 && more && !(p.ablate & 1) && wave >= 4) stage_load((c + 1) & 1, kt + 1);
        CGPT_FENCE
        CGPT_MM_ROWS(a0, b1, 0, 0, 1)            // G3: k-step 1, M-half 0
        CGPT_FENCE
        CGPT_LDA(a1, st, k_off1, 1)
        CGPT_FENCE
        CGPT_MM_ROWS(a0, b1, 0, 1, HM)
        CGPT_FENCE
        // every read of `st` by this wave has been issued; vmcnt(0) + lgkmcnt(0) + s_barrier: K-tile kt+1 has landed for
        // every wave and nobody still reads `st`.  G4 runs AFTER the barrier, covering the next tile's first reads.
#ifdef CGPT_STAMPS
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp_vm += ts1 - ts0; stamp_bar += ts2 - ts1;
#else
        __syncthreads();
#endif
        if (more) {
            CGPT_LDA(a0, nx, k_off0, 0)
            CGPT_LDB(b0, nx, k_off0)
        }
        CGPT_FENCE
        CGPT_MM_ROWS(a1, b1, 1, 0, HM)           // G4: k-step 1, M-half 1
        CGPT_FENCE
    }
    // ---- epilogue of tile (etm, etn).  Order matters for the in-order vmcnt counter: (1) pin the bias registers (their
    // loads were issued at tile start and retired by the K-loop barriers), (2) request K-tile 0 of the NEXT tile, (3) compute
    // and store.  On the full-tile fast path the stores then stream back to back with no s_waitcnt between them; a load in
    // this phase would make every later use wait for the previous store to complete.
    const int etm = tm, etn = tn;
#ifdef CGPT_STAMPS
    const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bias4[j]));
    if (t + (int)gridDim.x < ntiles) { set_tile(t + gridDim.x); stage_load(c & 1, 0); }

    const int m0 = etm * BM2 + wr * (BM2 / WM) + r15;      // lane owns rows m0 + i*16, columns n0 + j*16 .. +3
    const int n0 = etn * BN_ + wc * (BN_ / WN) + 4 * g;
    const bool full = (etm + 1) * BM2 <= p.M && (etn + 1) * BN_ <= p.N && !(p.ablate & 2);
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
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) emit(i, j, m0 + i * 16, n0 + j * 16);
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
        }
    }
#ifdef CGPT_STAMPS
    stamp_epi += __builtin_amdgcn_s_memtime() - te0;
#endif
    }   // persistent tile loop
#ifdef CGPT_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - stamp_begin; d[1] = stamp_ld; d[2] = stamp_bar; d[3] = stamp_epi;
    }
#endif
#undef CGPT_MM_ROWS
#undef CGPT_FENCE
}

template <int EPI, int BN_>
hipError_t launch_v2(const GemmParams& p, hipStream_t stream) {
    constexpr int lds_bytes = 2 * (256 + BN_) * BK * (int)sizeof(half_t);
    DeviceInfo di;
    if (hipError_t e = device_info(di); e != hipSuccess) return e;
    static bool configured[kMaxDevices] = {false};
    if (hipError_t e = configure_lds(&gemm2_f16_kernel<EPI, BN_>, lds_bytes, configured, di.dev); e != hipSuccess) return e;
    const int tiles = ((p.M + 255) / 256) * ((p.N + BN_ - 1) / BN_);
    const int num_cus = di.num_cus;
    const int grid = tiles < num_cus ? tiles : num_cus;     // one 512-thread workgroup per CU (LDS-limited), persistent
    hipLaunchKernelGGL((gemm2_f16_kernel<EPI, BN_>), dim3(grid), dim3(512), lds_bytes, stream, p);
    return hipGetLastError();
}

template <int BN_>
hipError_t launch_v2_epi(int epilogue, const GemmParams& p, hipStream_t stream) {
    switch (epilogue) {
        case EPI_F16: return launch_v2<EPI_F16, BN_>(p, stream);
        case EPI_F16_GELU: return launch_v2<EPI_F16_GELU, BN_>(p, stream);
        case EPI_F32: return launch_v2<EPI_F32, BN_>(p, stream);
        case EPI_RESID: return launch_v2<EPI_RESID, BN_>(p, stream);
        case EPI_PATCH: return launch_v2<EPI_PATCH, BN_>(p, stream);
        default: return hipErrorInvalidValue;
    }
}


// ------------------------------------------------------------------------------------------------ v3
// Same tile, staging and swizzle as v2<256>, different time structure (the "phase" structure of the MI355X guide):
// measured on v2 (in-kernel s_memtime stamps, profiles/r01/gemm_variants.txt) the OLDER wave of every SIMD wins MFMA
// arbitration, finishes its K-tile early and parks 25 % of its time at the barrier, after which the younger wave runs
// alone with its 8 LDS-DMA issues (40-70 cycles each) and 24 fragment reads exposed.  Here the two waves of a SIMD
// (w and w+4 = the top / bottom 128 rows of the tile) execute the same program ONE SLOT apart:
//     slot:      0     1     2     3     4     5     6     7   | 8 ...
//     waves 0-3  L0    M0    L1    M1    L2    M2    L3    M3  | L0'          L = fragment reads (+ LDS-DMA issue, L0/L1)
//     waves 4-7  (M3)  L0    M0    L1    M1    L2    M2    L3  | M3           M = 16 MFMAs (one quadrant of the wave tile)
// with an s_barrier between slots, so in every slot one wave of each SIMD feeds the matrix pipe while its partner
// does everything else.  Quadrant order (k-step, M-half) = (0,0) (0,1) (1,1) (1,0): B fragments are re-read only twice.
// Hazards: every L segment ends with lgkmcnt(0) BEFORE its barrier, so a stage is never overwritten while a read of it
// is outstanding (K-tile t+1 goes to the other stage, requested in L0/L1 of K-tile t); every wave retires its own
// LDS-DMA (vmcnt(0)) in the segment that occupies slot 7, so after that slot's barrier K-tile t+1 is visible to all.
// NQ = phases per K-tile: 4 (16-MFMA quadrants) or 2 (32 MFMAs = one k-step of the whole wave tile per phase).
template <int V> struct IntTag { static constexpr int value = V; };

// 16-byte stores of the fp16 epilogues: nontemporal (the outputs are streamed: written once, read by the next kernel, far larger
// than L2) -- in the model -1.9 % GEMM time against plain stores (-DCGPT_PLAIN_STORES builds those for the A/B).
#ifdef CGPT_PLAIN_STORES
#define CGPT_STORE16(v, ptr) (*(ptr) = (v))
#else
#define CGPT_STORE16(v, ptr) __builtin_nontemporal_store((v), (ptr))
#endif

template <int EPI, int NQ>
__global__ __launch_bounds__(512, 2) void gemm3_f16_kernel(GemmParams p) {
    constexpr int BM2 = 256, BN_ = 256, WN = 4, WM = 2;
    constexpr int TM = 8, TN = 4, HM = 4;
    constexpr int A_ELEMS = BM2 * BK, B_ELEMS = BN_ * BK, STAGE = A_ELEMS + B_ELEMS;
    constexpr int A_INSTR = 4, B_INSTR = 4;
    extern __shared__ __attribute__((aligned(16))) half_t smem3[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const bool late = wave >= 4;                       // the half that runs one slot behind
    const int r15 = lane & 15, g = lane >> 4;

    const int tiles_m = (p.M + BM2 - 1) / BM2;
    const int tiles_n = (p.N + BN_ - 1) / BN_;
    const int ntiles = tiles_m * tiles_n;

    const int lr = lane >> 3, cpos = lane & 7;
    const half_t* a_src[A_INSTR];
    const half_t* b_src[B_INSTR];
    // N = 256 k + 128 (ViT-G: proj / fc2 1408, qkv 4224): instead of a last 256-column tile that is half padding, the last
    // 384 columns are covered by TWO 192-column tiles (wave tile 128 x 48: TN = 3, 12 MFMAs per phase instead of 16, three
    // B pieces per wave instead of four).  With the block -> tile order used here every workgroup gets the same number of narrow
    // tiles when the batch fills 256 tile rows (t = b + 256 k walks tn in steps of 2 mod 6: {0,2,4} or {1,3,5} for N = 1408).
    const bool split_n = (p.N % 256) == 128 && p.N >= 384 && !(p.ablate & 16384);
    int tm = 0, tn = 0, ncol0 = 0;                     // tile coordinates and first column of the tile set_tile() last selected
    bool narrow = false;                               // ... and whether it is a 192-column tile
    auto set_tile = [&](int t) {
        tile_of_virtual_block(t, ntiles, tiles_m, tiles_n, tm, tn, p.group_m);
        narrow = split_n && tn >= tiles_n - 2;
        ncol0 = narrow ? (tiles_n - 2) * BN_ + (tn - (tiles_n - 2)) * 192 : tn * BN_;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int r = wave * (BM2 / 8) + i * 8 + lr;
            a_src[i] = p.A + (int64_t)(tm * BM2 + r) * p.lda + ((cpos ^ ((r >> 1) & 7)) << 3);
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int r = wave * (narrow ? 24 : BN_ / 8) + i * 8 + lr;      // LDS row of the B image (narrow: 192 rows, 24 per wave)
            b_src[i] = p.W + (int64_t)(ncol0 + r) * p.ldw + ((cpos ^ ((r >> 1) & 7)) << 3);
        }
    };
    auto load_a = [&](int stage, int kt) {
        half_t* sa = smem3 + stage * STAGE + wave * (BM2 / 8) * BK;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(sa + i * 8 * BK), 16, 0, 0);
    };
    auto load_b = [&](int stage, int kt) {             // for the tile set_tile() last selected
        half_t* sb = smem3 + stage * STAGE + A_ELEMS + wave * (narrow ? 24 : BN_ / 8) * BK;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i)
            if (i < 3 || !narrow)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + kt * BK),
                                                 (__attribute__((address_space(3))) void*)(sb + i * 8 * BK), 16, 0, 0);
    };

    const int sw = (r15 >> 1) & 7;
    const int k_off0 = ((g ^ sw) << 3), k_off1 = (((4 + g) ^ sw) << 3);
    const int a_rd = (wr * (BM2 / WM) + r15) * BK;
    const int b_rd_wide = A_ELEMS + (wc * (BN_ / WN) + r15) * BK, b_rd_narrow = A_ELEMS + (wc * 48 + r15) * BK;

    f32x4 acc[TM][TN];
    f16x8 af[TM], bf[TN];
#define CGPT_FENCE __builtin_amdgcn_sched_barrier(0);
#define CGPT_SLOT_END CGPT_FENCE __builtin_amdgcn_s_barrier(); CGPT_FENCE

    const int nk = p.K / BK;
    int c = 0;
    int t = blockIdx.x;
    bool lds_stores = false;
#ifdef CGPT_STAMPS
    unsigned long long st_wait = 0, st_loop = 0, st_epi = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    if (t < ntiles) { set_tile(t); load_a(0, 0); load_b(0, 0); }
    // one tile; TNv = column tiles of 16 per wave (4: 256-column tile, 3: 192-column tile)
    auto tile_body = [&](auto tnv_tag) __attribute__((always_inline)) {
        constexpr int TNv = decltype(tnv_tag)::value;
        constexpr bool NARROW = TNv == 3;
        const int b_rd = NARROW ? b_rd_narrow : b_rd_wide;
        const int wcols = NARROW ? 48 : BN_ / WN;       // columns per wave
        const int ecol0 = ncol0;                        // first column of THIS tile (set_tile moves on during the last K-tile)
        // bias of this wave's 64 (48) columns: ONE LDS-DMA request (4 bytes per lane) into a per-wave 256-byte LDS area, read back at
        // the epilogue.  (Register loads here cost four dependent round trips per tile: the compiler guards each conditional
        // load with its own `s_waitcnt vmcnt(0)`, which also drained the previous tile's stores; and the 16 values sat in VGPRs
        // through the whole K loop.)  The request is retired by the K loop's own waits long before the epilogue.
        float* bias_lds = reinterpret_cast<float*>(smem3 + 2 * STAGE) + wave * 64;
        if (p.bias) {
            const int bc = min(ecol0 + wc * wcols + lane, p.N - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + bc),
                                             (__attribute__((address_space(3))) void*)bias_lds, 4, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNv; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef CGPT_STAMPS
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        // K-tile 0 of this tile has landed (requested during the previous tile's last K-tile); both halves aligned.  After an
        // LDS-path epilogue its 8 requests are older than that epilogue's 16 stores: a counted wait leaves the stores in flight
        // (+1: the bias request just issued is the youngest entry of the queue)
        if (lds_stores && !(p.ablate & 1024)) {
            if (p.bias) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CGPT_SLOT_END
        if (late) { CGPT_SLOT_END }                    // the late half enters one slot behind
#ifdef CGPT_STAMPS
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
        const int etm = tm;                            // this tile's coordinates (set_tile below moves on to the next tile)
        const bool next_tile = t + (int)gridDim.x < ntiles;
        const bool early = next_tile && !(p.ablate & 16);   // request the next tile's first K-tile during this tile's last one

        for (int kt = 0; kt < nk; ++kt, ++c) {
            const half_t* st = smem3 + (c & 1) * STAGE;
            const bool more = kt + 1 < nk;
            // last K-tile: the other stage is free, so the NEXT tile's K-tile 0 is requested here, a whole K-tile before the
            // epilogue instead of after it (it has landed by the time the epilogue is over)
            const bool pre = !more && early;
            if (pre) set_tile(t + gridDim.x);
            const int nkt = pre ? 0 : kt + 1;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ks = (NQ == 4) ? (q >> 1) : q;
                const int ko = ks ? k_off1 : k_off0;
                const int h0 = (NQ == 4) ? ((q == 0 || q == 3) ? 0 : 1) : 0;    // first M-half of this phase
                const int nh = (NQ == 4) ? 1 : 2;                                // M-halves per phase
                // ---------------- L(q): fragment reads (+ the LDS-DMA requests of the next K-tile)
                if (NQ == 2 || q == 0 || q == 2) {
#pragma unroll
                    for (int j = 0; j < TNv; ++j) bf[j] = *reinterpret_cast<const f16x8*>(st + b_rd + j * 16 * BK + ko);
                }
#pragma unroll
                for (int i = 0; i < nh * HM; ++i)
                    af[h0 * HM + i] = *reinterpret_cast<const f16x8*>(st + a_rd + (h0 * HM + i) * 16 * BK + ko);
                if ((more || pre) && !(p.ablate & 1)) {
                    if (q == 0) load_a((c + 1) & 1, nkt);
                    if ((NQ == 4 && q == 1) || (NQ == 2 && q == 0)) load_b((c + 1) & 1, nkt);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (q == NQ - 1 && late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CGPT_SLOT_END
                // ---------------- M(q): 16 * nh MFMAs
#pragma unroll
                for (int i = 0; i < nh * HM; ++i)
#pragma unroll
                    for (int j = 0; j < TNv; ++j)
                        acc[h0 * HM + i][j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[h0 * HM + i], acc[h0 * HM + i][j], 0, 0, 0);
                if (q == NQ - 1 && !late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CGPT_SLOT_END
            }
        }
        if (!late) { CGPT_SLOT_END }                   // the early half waits one slot for its partners' last M
#ifdef CGPT_STAMPS
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif

        if (next_tile && !early) { set_tile(t + gridDim.x); load_a(c & 1, 0); load_b(c & 1, 0); }
        f32x4 bias4[TNv];
        {
            int el0;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el0));
#pragma unroll
            for (int j = 0; j < TNv; ++j)
                bias4[j] = p.bias ? *reinterpret_cast<const f32x4*>(bias_lds + j * 16 + 4 * (el0 >> 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const bool full = (etm + 1) * BM2 <= p.M && ecol0 + (NARROW ? 192 : BN_) <= p.N && !(p.ablate & 2);
        constexpr bool F16_OUT = EPI == EPI_F16 || EPI == EPI_F16_GELU;
        if (NARROW && !(F16_OUT && full && (p.ldo & 7) == 0 && !(p.ablate & 512))) {
            // 192-column tile, fp32 / patch-embed / partial: direct epilogue
            if constexpr (NARROW) {
                f32x4 accn[TM][3];
#pragma unroll
                for (int i2 = 0; i2 < TM; ++i2)
#pragma unroll
                    for (int j2 = 0; j2 < 3; ++j2) accn[i2][j2] = acc[i2][j2];
                gemm_epilogue_256<EPI, TM, 3>(p, accn, bias4, etm * BM2 + wr * (BM2 / WM) + r15, ecol0 + wc * 48 + 4 * g, full);
            }
            lds_stores = false;
        } else if (F16_OUT && full && (p.ldo & 7) == 0 && !(p.ablate & 512)) {
            // fp16 output of a full tile: transposed through LDS so that a lane stores 16 contiguous bytes and 8 lanes one
            // 128-byte line.  (Straight from the accumulators a store instruction writes 16 rows x 32 bytes: 4x the L2 write
            // requests, 2x the store instructions; measured in the model: the stores cost 15 of the GEMMs' 97 ms per certify,
            // 8 of them even when the target sits in L2.)  Scratch = the LDS stage of the last K-tile: every wave finished
            // reading it before the barrier above, and the next tile does not request into it before its first barrier.
            // Each wave owns 8 KiB of it and handles its 128 x 64 sub-tile in two passes of 64 rows; rows are 128 B, the
            // 16-byte chunk index is XOR-swizzled with row & 7 (conflict-free ds_write_b64 / ds_read_b128).  A 192-column tile
            // uses 6 of a row's 8 chunks (128 x 48 per wave): the read / store instructions run with 48 of 64 lanes.
                    half_t* scr = smem3 + ((c + 1) & 1) * STAGE + wave * 4096;
            half_t* outp = reinterpret_cast<half_t*>(p.out);
            // lane coordinates recomputed here (volatile: not merged with the copies the K loop uses), so that no epilogue-only
            // value is kept -- or spilled -- across the K loop (a scratch reload would wait behind the LDS-DMA requests in flight)
            int el;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
            const int row_rd = el >> 3, ch_rd = el & 7, r15 = el & 15, g = el >> 4;
#ifndef CGPT_V3_PASSES
#define CGPT_V3_PASSES 2
#endif
            constexpr int NPS = CGPT_V3_PASSES, RT = 8 / NPS;           // passes, and 16-row tiles per pass (experiment knob; product: 2 x 4)
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
#pragma unroll
                for (int ii = 0; ii < RT; ++ii) {
                    const int i = ps * RT + ii;
                    const int row = ii * 16 + r15;
#pragma unroll
                    for (int jj = 0; jj < TNv; ++jj) {
                        const f32x4 v = acc[i][jj] + bias4[jj];
                        f16x4 hv;
                        if constexpr (EPI == EPI_F16_GELU) hv = gelu_f16x4(v);
                        else hv = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        const int ch = (jj * 2 + (g >> 1)) ^ (row & 7);
                        *reinterpret_cast<f16x4*>(scr + row * 64 + ch * 8 + (g & 1) * 4) = hv;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same wave wrote what it now reads
                // one 64-bit row address per pass; the 8 stores step by 8 rows (a uniform offset)
                half_t* dst0 = outp + ((int64_t)etm * BM2 + wr * (BM2 / WM) + ps * (RT * 16) + row_rd) * p.ldo + ecol0 + wc * wcols + ch_rd * 8;
                const int64_t step = 8 * p.ldo;
#pragma unroll
                for (int it = 0; it < RT * 2; ++it) {
                    const int row = it * 8 + row_rd;
                    if (!NARROW || ch_rd < 6) {
                        const f16x8 o = *reinterpret_cast<const f16x8*>(scr + row * 64 + ((ch_rd ^ (row & 7)) * 8));
                        CGPT_STORE16(o, reinterpret_cast<f16x8*>(dst0 + it * step));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads returned before the second pass overwrites
            }
            lds_stores = early;                        // (only then are this tile's stores younger than the next K-tile 0 requests)
        } else {
            lds_stores = false;
            if constexpr (!NARROW)
                gemm_epilogue_256<EPI, TM, TN>(p, acc, bias4, etm * BM2 + wr * (BM2 / WM) + r15, ecol0 + wc * (BN_ / WN) + 4 * g, full);
        }
#ifdef CGPT_STAMPS
        st_wait += ts1 - ts0; st_loop += ts2 - ts1; st_epi += __builtin_amdgcn_s_memtime() - ts2;
#endif
    };
    for (; t < ntiles; t += gridDim.x) {
        if (narrow) tile_body(IntTag<3>{});
        else tile_body(IntTag<4>{});
    }
#ifdef CGPT_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - st_begin; d[1] = st_wait; d[2] = st_loop; d[3] = st_epi;
    }
#endif
#undef CGPT_FENCE
#undef CGPT_SLOT_END
}

template <int EPI, int NQ>
hipError_t launch_v3(const GemmParams& p, hipStream_t stream) {
    constexpr int lds_bytes = 2 * (256 + 256) * BK * (int)sizeof(half_t) + 8 * 64 * (int)sizeof(float);   // two stages + bias
    DeviceInfo di;
    if (hipError_t e = device_info(di); e != hipSuccess) return e;
    static bool configured[kMaxDevices] = {false};
    if (hipError_t e = configure_lds(&gemm3_f16_kernel<EPI, NQ>, lds_bytes, configured, di.dev); e != hipSuccess) return e;
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int num_cus = di.num_cus;
    const int grid = tiles < num_cus ? tiles : num_cus;
    hipLaunchKernelGGL((gemm3_f16_kernel<EPI, NQ>), dim3(grid), dim3(512), lds_bytes, stream, p);
    return hipGetLastError();
}

template <int NQ>
hipError_t launch_v3_epi(int epilogue, const GemmParams& p, hipStream_t stream) {
    switch (epilogue) {
        case EPI_F16: return launch_v3<EPI_F16, NQ>(p, stream);
        case EPI_F16_GELU: return launch_v3<EPI_F16_GELU, NQ>(p, stream);
        case EPI_F32: return launch_v3<EPI_F32, NQ>(p, stream);
        case EPI_RESID: return launch_v3<EPI_RESID, NQ>(p, stream);
        case EPI_PATCH: return launch_v3<EPI_PATCH, NQ>(p, stream);
        default: return hipErrorInvalidValue;
    }
}


#ifdef CGPT_LAB   // lab-only schedules (slower or equal; kept for A/B runs: make LAB=1)
#include "lab/gemm_v4v5.inc"
#endif  // CGPT_LAB

}  // namespace

int g_gemm_kernel = 0;
int g_gemm_ablate = 0;
int g_gemm_grid = 0;
int g_gemm_group_m = 4;   // measured: 4 ~ 8 > 2 > 16 (profiles/r01/gemm_variants.txt)
unsigned long long* g_gemm_dbg = nullptr;

// Result-preserving switches of gemm_ablate (TEST-ONLY: each turns one optimisation off so that a test can check the bits do not
// depend on it): 512 = LDS-transposed fp16 epilogue, 16384 = 192-column last tiles for N = 256k + 128.
// Lab builds additionally honour 16 / 1024 (request / wait placement of the phased kernel) and the WRONG-result timing switches
// 1 / 2 / 4 (skip loads / stores / MFMAs).
[[maybe_unused]] constexpr int kAblateSafeBits = 512 | 16384;

hipError_t launch_gemm(int epilogue, const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
#ifdef CGPT_LAB
    p.ablate = g_gemm_ablate;
    p.group_m = g_gemm_group_m;
    p.dbg = g_gemm_dbg;
#else
    p.ablate = g_gemm_ablate & kAblateSafeBits;
    p.group_m = 4;
    p.dbg = nullptr;
#endif
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0 || (p.lda % 8) != 0 || (p.ldw % 8) != 0)
        return hipErrorInvalidValue;
    // Kernel choice (speed only; every kernel gives the same bits).  The 256-row kernels need A readable for round_up(M,256)
    // rows and W for round_up(N,256) rows: the library's activation and weight buffers are padded to 256 rows.
    const bool vec_ok = (p.N % 4) == 0 && (p.ldo % 4) == 0 && (p.ldaux % 4) == 0;
    const int force = vec_ok ? g_gemm_kernel : 1;   // 0 auto, 1 = 128x128 register-staged, 3 = 256x128 direct-to-LDS, 4 = 256x256 phased
    if (force == 3 && (p.N % 128) == 0) return launch_v2_epi<128>(epilogue, p, stream);
    if (force == 4) return launch_v3_epi<4>(epilogue, p, stream);
    if (force == 14) return launch_v9_epi(epilogue, p, stream);
#ifdef CGPT_LAB
    if (force == 2) return launch_v2_epi<256>(epilogue, p, stream);
    if (force == 5) return launch_v3_epi<2>(epilogue, p, stream);
    if (force == 6) return launch_v4_epi<4>(epilogue, p, stream);
    if (force == 7) return launch_v4_epi<5>(epilogue, p, stream);
    if (force == 8) return launch_v5_epi(epilogue, p, stream);
    if (force == 9) return launch_v6_epi(epilogue, 0, p, stream);
    if (force == 10) return launch_v6_epi(epilogue, 1, p, stream);
    if (force == 11) return launch_v8_epi(epilogue, p, stream);
#endif
    if ((force == 0 || force == 15) && p.M >= 1024) {
        // measured on MI355X (profiles/r01/gemm_variants.txt): the 256x256 direct-to-LDS tile with the phase-alternating
        // schedule wins on every ViT / Q-Former shape, also when N is not a multiple of 256 (weights are allocated with 256-row
        // padding) ... except when 256x256 tiles would leave more than half of the 256 CUs idle (the Q-Former's N = 768 linears
        // at M = 200 x 32 rows: 75 tiles): there the 256x128 tile is 25-35 % faster.
        const int tiles256 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
        if (tiles256 <= 128 && (p.N % 128) == 0) return launch_v2_epi<128>(epilogue, p, stream);
#ifdef CGPT_LAB
        if (force == 15) return launch_v9_epi(epilogue, p, stream);          // lab: the two-phase quadrant kernel wherever the automatic choice is a 256x256 tile
#endif
        // the two-phase quadrant kernel (gemm9.hip) wins on every ViT shape in the model: qkv 650 -> 611, proj 238 -> 229,
        // fc1 + GELU ~1 000 -> 976, fc2 859 -> 823 us per 255-sample launch (profiles/r02/gemm_two_phase_variants.txt)
        return launch_v9_epi(epilogue, p, stream);
    }
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
