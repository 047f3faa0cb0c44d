// gemm6.hip -- 256x256 tile, FOUR waves (2x2), wave tile 128x128, one wave per SIMD with the whole 512-register file.
//
// Why (profiles/r02/gemm_v6.txt): the 8-wave kernels (v2..v5, wave tile 128x64) read (128+64)*64*2 B * 8 waves = 192 KiB of
// fragments from LDS per 256x256x64 K-tile -- the LDS pipe is busy 75 % of the MFMA time and the kernel sits on the power
// limit (1.46 GHz with the operand stream, 1.97 GHz without).  A 128x128 wave tile needs (128+128)*64*2 B * 4 = 128 KiB per
// K-tile (-33 % LDS bytes per MFMA) and half as many barrier participants; the vendor library's kernel for these shapes
// (MT256x256x64, 4 waves, MIWT 8x8) runs 1.25-1.3 PFLOP/s sustained where v3 runs 1.05-1.1.
//
// With ONE wave per SIMD nobody covers a stall, so the K loop is a continuous software pipeline:
//   * LDS is a ring of NS = 4 slots of one K=32 sub-tile each (A 256x32 + W 256x32 halfs = 32 KiB; rows are 64 B, chunk c of
//     row r at position c ^ ((-(r>>2)) & 3): conflict-free ds_read_b128 fragments, see gemm.hip v4).
//   * step c (64 MFMAs = 1024 MFMA cycles) consumes register fragment set c&1 and, behind its MFMAs,
//       - waits for ITS requests of sub-tile c+1 (counted vmcnt: the two younger groups stay in flight), s_barrier,
//       - reads fragment set (c+1)&1 from slot (c+1)%NS (16 ds_read_b128, in the first half of the step),
//       - requests sub-tile c+NS into slot c%NS (8 global_load_lds_dwordx4 per wave, ONE per 8 MFMAs: an LDS-DMA request
//         occupies the issue port for tens of cycles when the queue is backed up and nobody else feeds the matrix pipe).
//     Slot c%NS is free after the barrier of step c: every wave retired its reads of it before its first MFMA of step c.
//     A request therefore has three full steps to land.
//   * the stream of sub-tiles runs across tile boundaries (persistent workgroup): the next tile's first NS sub-tiles are
//     requested during the last NS steps of the current one, and its fragment set 0 is read in the last step.
//   * epilogue: 64 vector stores per lane issue back to back.  vmcnt retires in order, so a counted wait that follows them
//     would wait for the stores; the first NS-1 steps of the next tile wait with vmcnt(63) instead, which is exact there:
//     the group they need is older than >= 63 younger operations (16 requests + 64 stores).
//   * past the end of the block's work the request cursor stays on its last sub-tile (re-requests land in slots nobody
//     consumes) so every step issues exactly 8 requests and the counted wait needs no special cases.
#include <stdlib.h>

#include "gemm_common.h"

namespace cgpt {

namespace {

template <int EPI, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm6_f16_kernel(GemmParams p) {
    constexpr int BM2 = 256, BN_ = 256, BK6 = 32, NS = 4;
    constexpr int TM = 8, TN = 8;
    constexpr bool PAIR = (MODE == 1);   // requests issued as full 128-byte lines: sub-tiles c+3 and c+4 together, every odd step
    constexpr int A_ELEMS = BM2 * BK6, STAGE = (BM2 + BN_) * BK6;     // halfs
    constexpr bool kStoresOnly = (EPI == EPI_F16 || EPI == EPI_F16_GELU || EPI == EPI_F32);
    extern __shared__ __attribute__((aligned(16))) half_t smem6[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r15 = lane & 15, g = lane >> 4;

    const int tiles_m = (p.M + BM2 - 1) / BM2;
    const int tiles_n = (p.N + BN_ - 1) / BN_;
    const int ntiles = tiles_m * tiles_n;
    const int ns = p.K / BK6;

    // ---- request cursor: wave w owns pieces 4w..4w+3 (tile rows 64w .. 64w+63) of A and of W in every sub-tile;
    // one piece = 1 KiB = 16 rows x 64 B, lane l -> row l>>2, position l&3, source chunk (l&3) ^ ((-(l>>4)) & 3).
    const int prow = lane >> 2;
    const int psrc = ((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 3;
    const int64_t a_piece = 16 * p.lda, b_piece = 16 * p.ldw;
    const half_t* a_src = p.A;
    const half_t* b_src = p.W;
    int pt = blockIdx.x, ps = 0;
    auto set_cursor = [&](int t) {
        int ptm, ptn;
        tile_of_virtual_block(t, ntiles, tiles_m, tiles_n, ptm, ptn, p.group_m);
        a_src = p.A + (int64_t)(ptm * BM2 + 64 * wave + prow) * p.lda + psrc;
        b_src = p.W + (int64_t)(ptn * BN_ + 64 * wave + prow) * p.ldw + psrc;
    };
    auto request_one = [&](int slot, int idx, int koff = 0) {           // idx 0..3: A pieces, 4..7: W pieces
        half_t* dst = smem6 + slot * STAGE + (idx >= 4 ? A_ELEMS : 0) + (4 * wave + (idx & 3)) * 16 * BK6;
        const half_t* src = (idx >= 4 ? b_src + (idx & 3) * b_piece : a_src + (idx & 3) * a_piece) + koff;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    constexpr int ADV = PAIR ? 2 : 1;                                   // sub-tiles per request group
    auto advance_cursor = [&]() {
        if (ps + ADV < ns) {
            ps += ADV; a_src += ADV * BK6; b_src += ADV * BK6;
        } else if (pt + (int)gridDim.x < ntiles) {
            pt += gridDim.x; ps = 0; set_cursor(pt);
        }                                                               // else: stay on the last sub-tile (pair): re-requests are harmless
    };

    const int rsw = ((g ^ ((0 - (r15 >> 2)) & 3)) << 3);                // swizzled chunk of this lane's fragment
    const int a_rd = (wr * 128 + r15) * BK6 + rsw;
    const int b_rd = A_ELEMS + (wc * 128 + r15) * BK6 + rsw;

    const int wsel = 2 * wave;
    f32x4 acc[TM][TN];
    f16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
#define CGPT_FENCE __builtin_amdgcn_sched_barrier(0);
// MFMAs are inline asm with the accumulators pinned to AGPRs ("a") and the fragments to VGPRs ("v"): with the builtin the
// register allocator mixed the two files and filled the K loop with v_accvgpr copies and scratch reloads.  Z = first step
// of a tile (srcC = 0 instead of 256 v_accvgpr_write per tile).
#define CGPT_MM_ROW(FA, FB, i_, Z)                                                                               \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                             \
        if (Z) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc[i_][j]) : "v"(FB[j]), "v"(FA[i_])); \
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i_][j]) : "v"(FB[j]), "v"(FA[i_])); \
    }
#define CGPT_RD4(F, base_, q_)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        F[(q_) * 4 + i] = *reinterpret_cast<const f16x8*>((base_) + ((q_) * 4 + i) * 16 * BK6);
    // One step of the ring.  CA/CB: fragment set consumed; NA/NB: set filled for the next step.
#ifdef CGPT_STAMPS
    unsigned long long stamp_vm = 0, stamp_bar = 0, stamp_epi = 0, stamp_pre = 0, stamp_last = 0;
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memtime();
#define CGPT_STAMP(x) const unsigned long long x = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define CGPT_STAMP_ACC stamp_pre += ts0 - stamp_last; stamp_vm += ts1 - ts0; stamp_bar += ts2 - ts1; stamp_last = ts2;
#else
#define CGPT_STAMP(x)
#define CGPT_STAMP_ACC
#endif
#define CGPT_REQ(i_)                                                                                             \
    if (!(p.ablate & 1)) {                                                                                       \
        if constexpr (!PAIR) request_one(c % NS, i_);                                                            \
        else { request_one((c + NS - 1) % NS, i_, 0); request_one(c % NS, i_, BK6); }                            \
    }
    // One row of 8 MFMAs with the other instruction streams threaded through it one instruction at a time (a lone wave
    // issues in order: a burst of reads / requests between MFMA groups leaves the matrix pipe idle for its whole issue time).
    //   RD: after MFMAs 1,3,5,7 one fragment read (quartet q_ of F from base_);  R0_/R1_: request indices issued after MFMA
    //   2*wave (+1): the four waves of the workgroup run in lockstep, so they take turns at the shared address path.
#define CGPT_ROW(CA, CB, i_, Z, RD, F, base_, q_, R0_, R1_)                                                      \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                             \
        if (Z) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc[i_][j]) : "v"(CB[j]), "v"(CA[i_])); \
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i_][j]) : "v"(CB[j]), "v"(CA[i_])); \
        if ((RD) && (j & 1))                                                                                     \
            F[(q_) * 4 + (j >> 1)] = *reinterpret_cast<const f16x8*>((base_) + ((q_) * 4 + (j >> 1)) * 16 * BK6); \
        if constexpr (req) {                                                                                     \
            if ((R0_) >= 0 && j == wsel) { CGPT_REQ(R0_) }                                                       \
            if ((R1_) >= 0 && j == wsel + 1) { CGPT_REQ(R1_) }                                                   \
        }                                                                                                        \
        CGPT_FENCE                                                                                               \
    }
    // ODD: second step of a pair (c odd).  PAIR mode requests only there, two sub-tiles (one 128-byte line per row) at once.
#define CGPT_STEP(CA, CB, NA, NB, W63, Z, ODD)                                                                   \
    {                                                                                                            \
        const half_t* nst = smem6 + ((c + 1) % NS) * STAGE;                                                      \
        constexpr bool req = !PAIR || (ODD);                                                                     \
        CGPT_MM_ROW(CA, CB, 0, Z)                                                                                \
        CGPT_FENCE                                                                                               \
        CGPT_STAMP(ts0)                                                                                          \
        if (W63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");                                               \
        else if (PAIR && (ODD)) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                 \
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        CGPT_STAMP(ts1)                                                                                          \
        __builtin_amdgcn_s_barrier();                                                                            \
        CGPT_STAMP(ts2)                                                                                          \
        CGPT_STAMP_ACC                                                                                           \
        CGPT_FENCE                                                                                               \
        CGPT_ROW(CA, CB, 1, Z, true, NA, nst + a_rd, 0, 0, -1)                                                   \
        CGPT_ROW(CA, CB, 2, Z, true, NA, nst + a_rd, 1, 1, -1)                                                   \
        CGPT_ROW(CA, CB, 3, Z, true, NB, nst + b_rd, 0, 2, -1)                                                   \
        CGPT_ROW(CA, CB, 4, Z, true, NB, nst + b_rd, 1, 3, -1)                                                   \
        CGPT_ROW(CA, CB, 5, Z, false, NA, nst, 0, 4, -1)                                                         \
        CGPT_ROW(CA, CB, 6, Z, false, NA, nst, 0, 5, -1)                                                         \
        CGPT_ROW(CA, CB, 7, Z, false, NA, nst, 0, 6, 7)                                                          \
        if constexpr (req) { advance_cursor(); }                           \
        ++c;                                                                                                     \
    }

    // ---- prologue: fill the ring, park the bias vector behind it (keeps 32 registers out of the K loop and the
    // epilogue's bias reads out of the in-order vmcnt queue), wait for sub-tile 0, read fragment set 0
    set_cursor(pt);
#pragma unroll
    for (int s = 0; s < NS; s += ADV) {
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) {
            request_one(s, idx);
            if constexpr (PAIR) request_one(s + 1, idx, BK6);
        }
        advance_cursor();
    }
    float* sbias = reinterpret_cast<float*>(smem6 + NS * STAGE);
    for (int n = tid * 4; n < tiles_n * BN_; n += 1024) {
        f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) b = *reinterpret_cast<const f32x4*>(p.bias + n);        // N % 4 == 0 (launch check)
        *reinterpret_cast<f32x4*>(sbias + n) = b;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CGPT_RD4(fa0, smem6 + a_rd, 0) CGPT_RD4(fa0, smem6 + a_rd, 1)
    CGPT_RD4(fb0, smem6 + b_rd, 0) CGPT_RD4(fb0, smem6 + b_rd, 1)

    unsigned c = 0;                                    // running sub-tile counter: sub-tile c lives in slot c % NS
    bool drained = true;                               // no epilogue stores of a previous tile can be outstanding
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int tm, tn;
        tile_of_virtual_block(t, ntiles, tiles_m, tiles_n, tm, tn, p.group_m);
        {   // first two steps of the tile: the first one starts the accumulators from zero
            const bool w = !drained;                   // after a full tile's 64 stores: store-tolerant wait (see header)
            CGPT_STEP(fa0, fb0, fa1, fb1, w, true, false)
            CGPT_STEP(fa1, fb1, fa0, fb0, w, false, true)
        }
#pragma nounroll
        for (int s = 2; s < ns; s += 2) {
            const bool w0 = !drained && s < NS - 1, w1 = !drained && s + 1 < NS - 1;
            CGPT_STEP(fa0, fb0, fa1, fb1, w0, false, false)
            CGPT_STEP(fa1, fb1, fa0, fb0, w1, false, true)
        }
#ifdef CGPT_STAMPS
        const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
        // the last MFMAs must have left the pipe before their AGPRs are read (asm MFMAs get no automatic hazard padding)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

        // lane coordinates are recomputed here (volatile asm: not merged with the copies above) so that no epilogue-only
        // value stays live -- or gets spilled -- across the K loop: a scratch reload sits in the vmcnt queue behind the
        // in-flight LDS-DMA requests and would drain them.
        int el;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
        const int er15 = el & 15, eg = el >> 4;
        f32x4 bias4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bias4[j] = *reinterpret_cast<const f32x4*>(sbias + tn * BN_ + wc * 128 + 4 * eg + j * 16);
        const bool full = (tm + 1) * BM2 <= p.M && (tn + 1) * BN_ <= p.N && !(p.ablate & 2);
        gemm_epilogue_256<EPI, TM, TN, true>(p, acc, bias4, tm * BM2 + wr * 128 + er15, tn * BN_ + wc * 128 + 4 * eg, full);
        drained = !(kStoresOnly && full);
#ifdef CGPT_STAMPS
        stamp_epi += __builtin_amdgcn_s_memtime() - te0;
#endif
    }
#ifdef CGPT_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - stamp_begin; d[1] = stamp_vm; d[2] = stamp_bar; d[3] = stamp_epi;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing re-requests must land before the LDS is released
#undef CGPT_FENCE
#undef CGPT_MM_ROW
#undef CGPT_RD4
#undef CGPT_STEP
#undef CGPT_ROW
#undef CGPT_REQ
#undef CGPT_STAMP
#undef CGPT_STAMP_ACC
}

template <int EPI, int MODE>
hipError_t launch_v6(const GemmParams& p, hipStream_t stream) {
    constexpr int ring_bytes = 4 * (256 + 256) * 32 * (int)sizeof(half_t);
    constexpr int max_lds = 160 * 1024;
    const int lds_bytes = ring_bytes + ((p.N + 255) / 256) * 256 * (int)sizeof(float);
    if (lds_bytes > max_lds || (p.N % 4) != 0) return hipErrorInvalidValue;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm6_f16_kernel<EPI, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    static int num_cus = 0;
    if (num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = tiles < num_cus ? tiles : num_cus;
    hipLaunchKernelGGL((gemm6_f16_kernel<EPI, MODE>), dim3(grid), dim3(256), lds_bytes, stream, p);
    return hipGetLastError();
}


template <int MODE>
hipError_t launch_v6_mode(int epilogue, const GemmParams& p, hipStream_t stream) {
    switch (epilogue) {
        case EPI_F16: return launch_v6<EPI_F16, MODE>(p, stream);
        case EPI_F16_GELU: return launch_v6<EPI_F16_GELU, MODE>(p, stream);
        case EPI_F32: return launch_v6<EPI_F32, MODE>(p, stream);
        case EPI_RESID: return launch_v6<EPI_RESID, MODE>(p, stream);
        case EPI_PATCH: return launch_v6<EPI_PATCH, MODE>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

hipError_t launch_v6_epi(int epilogue, int mode, const GemmParams& p, hipStream_t stream) {
    return mode == 1 ? launch_v6_mode<1>(epilogue, p, stream) : launch_v6_mode<0>(epilogue, p, stream);
}

}  // namespace cgpt
