// gemm8.hip -- 128x256 tile, FOUR waves per workgroup (1x4, wave tile 128x64 as in v3), TWO workgroups per CU.
//
// Why: on the K = 1408 shapes (qkv, proj, fc1: 73 % of the GEMM FLOPs) a 256x256 tile is only 22 K-tiles long and all
// eight waves of the v3 workgroup sit in the epilogue at the same time: PMC shows 52 % MFMA-busy there against 62-63 % on
// the K = 6144 shape (profiles/r01/gemm_variants.txt).  Two independent workgroups per CU have their own barriers and
// drift apart, so one's epilogue (and its LDS-DMA issue, and its barrier waits) runs under the other's MFMAs -- each SIMD
// holds one wave of each.
// Price: a 128x256 tile moves (128+256) rows of operands per 128x256 outputs, 1.5x the L2->LDS bytes per FLOP of 256x256.
//   * LDS per workgroup: ring of three K=32 slots of A 128x32 + W 256x32 halfs = 24 KiB each (72 KiB; two workgroups 144 KiB).
//     64-byte rows, chunk c of row r at position c ^ ((-(r>>2)) & 3) (conflict-free ds_read_b128, see v4 / v6).
//   * step c: wait for ITS requests of sub-tile c (vmcnt(6): the six of sub-tile c+1 stay in flight), s_barrier, request
//     sub-tile c+2 into slot (c+2)%3 (it held sub-tile c-1, which every wave finished before this barrier), read the 12
//     fragments of slot c%3, 32 MFMAs.  The sub-tile stream runs across tile boundaries (persistent workgroup); past the
//     end the cursor stays on the last sub-tile so every step issues exactly six requests.
//   * bias: one extra 256-byte LDS-DMA request per wave per tile parks the tile's 256 bias values behind the ring (two
//     buffers, by tile parity); the epilogue reads them with ds_read, so no register stays live across the K loop and no
//     compiler-visible global load sits in the in-order vmcnt queue (its wait would drain the requests in flight).
//   * after a full tile's 32 stores the first two steps of the next tile wait with vmcnt(38) instead of vmcnt(6): the group
//     they need is older than >= 38 younger operations (6 requests + 32 stores), so the stores are not drained.
#include <stdlib.h>

#include "gemm_common.h"

namespace cgpt {

namespace {

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm8_f16_kernel(GemmParams p) {
    constexpr int BM8 = 128, BN8 = 256, BK8 = 32, NS = 3;
    constexpr int TM = 8, TN = 4;
    constexpr int A_ELEMS = BM8 * BK8, STAGE = (BM8 + BN8) * BK8;     // halfs: 12288 = 24 KiB
    extern __shared__ __attribute__((aligned(16))) half_t smem8[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // = wave column: tile columns 64*wave .. +63
    const int r15 = lane & 15, g = lane >> 4;

    const int tiles_m = (p.M + BM8 - 1) / BM8;
    const int tiles_n = (p.N + BN8 - 1) / BN8;
    const int ntiles = tiles_m * tiles_n;
    const int ns = p.K / BK8;

    // request cursor: a sub-tile is 8 A pieces + 16 W pieces of 1 KiB (16 rows x 64 B); wave w requests A pieces 2w, 2w+1 and
    // W pieces 4w .. 4w+3.  Lane l -> row l>>2 of the piece, position l&3, source chunk (l&3) ^ ((-(l>>4)) & 3).
    const int prow = lane >> 2;
    const int psrc = ((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 3;
    const int64_t a_piece = 16 * p.lda, b_piece = 16 * p.ldw;
    const half_t* a_src = p.A;
    const half_t* b_src = p.W;
    int pt = blockIdx.x, ps = 0;
    auto set_cursor = [&](int t) {
        int ptm, ptn;
        tile_of_virtual_block(t, ntiles, tiles_m, tiles_n, ptm, ptn, 2 * p.group_m);
        a_src = p.A + (int64_t)(ptm * BM8 + 32 * wave + prow) * p.lda + psrc;
        b_src = p.W + (int64_t)(ptn * BN8 + 64 * wave + prow) * p.ldw + psrc;
    };
    auto request = [&](int slot) {
        half_t* sa = smem8 + slot * STAGE + (2 * wave) * 16 * BK8;
        half_t* sb = smem8 + slot * STAGE + A_ELEMS + (4 * wave) * 16 * BK8;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src + i * a_piece),
                                             (__attribute__((address_space(3))) void*)(sa + i * 16 * BK8), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src + i * b_piece),
                                             (__attribute__((address_space(3))) void*)(sb + i * 16 * BK8), 16, 0, 0);
        if (ps + 1 < ns) {
            ++ps; a_src += BK8; b_src += BK8;
        } else if (pt + (int)gridDim.x < ntiles) {
            pt += gridDim.x; ps = 0; set_cursor(pt);
        }                                                               // else: stay on the last sub-tile
    };

    const int rsw = ((g ^ ((0 - (r15 >> 2)) & 3)) << 3);
    const int a_rd = r15 * BK8 + rsw;
    const int b_rd = A_ELEMS + (wave * 64 + r15) * BK8 + rsw;

    f32x4 acc[TM][TN];
    f16x8 af[TM], bf[TN];
    constexpr bool kStoresOnly = (EPI == EPI_F16 || EPI == EPI_F16_GELU || EPI == EPI_F32);

    float* sbias = reinterpret_cast<float*>(smem8 + NS * STAGE);       // [2][256]
    if (!p.bias) {
        sbias[tid] = 0.f;
        sbias[256 + tid] = 0.f;
    }
    set_cursor(pt);
    request(0);
    request(1);
    unsigned c = 0;
    bool drained = true;
    int parity = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, parity ^= 1) {
        int tm, tn;
        tile_of_virtual_block(t, ntiles, tiles_m, tiles_n, tm, tn, 2 * p.group_m);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int s = 0; s < ns; ++s, ++c) {
            // sub-tile c: its six requests are older than the six of sub-tile c+1 (and than the stores of an epilogue in between)
            if (!drained && s < 2) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0 && p.bias) {                   // this wave's 64 bias values of the tile -> LDS (256 bytes, one request,
                const int col = min(tn * BN8 + wave * 64 + lane, p.N - 1);   // issued BEFORE this step's group: see the wait below)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + col),
                                                 (__attribute__((address_space(3))) void*)(sbias + parity * 256 + wave * 64), 4, 0, 0);
            }
            if (!(p.ablate & 1)) request((c + 2) % NS);
            const half_t* st = smem8 + (c % NS) * STAGE;
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f16x8*>(st + b_rd + j * 16 * BK8);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f16x8*>(st + a_rd + i * 16 * BK8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
            // every read of slot c%3 must have returned before the NEXT barrier lets somebody refill a slot (the MFMAs above
            // consumed them all, so this costs nothing)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // the bias request is older than the two groups requested in the last two steps: leaving those 12 in flight retires it
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        f32x4 bias4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bias4[j] = *reinterpret_cast<const f32x4*>(sbias + parity * 256 + wave * 64 + 4 * g + j * 16);
        const bool full = (tm + 1) * BM8 <= p.M && (tn + 1) * BN8 <= p.N && !(p.ablate & 2);
        gemm_epilogue_256<EPI, TM, TN>(p, acc, bias4, tm * BM8 + r15, tn * BN8 + wave * 64 + 4 * g, full);
        drained = !(kStoresOnly && full);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI>
hipError_t launch_v8(const GemmParams& p, hipStream_t stream) {
    constexpr int lds_bytes = 3 * (128 + 256) * 32 * (int)sizeof(half_t) + 2 * 256 * (int)sizeof(float);
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8_f16_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int tiles = ((p.M + 127) / 128) * ((p.N + 255) / 256);
    static int num_cus = 0;
    if (num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = tiles < 2 * num_cus ? tiles : 2 * num_cus;
    hipLaunchKernelGGL((gemm8_f16_kernel<EPI>), dim3(grid), dim3(256), lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_v8_epi(int epilogue, const GemmParams& p, hipStream_t stream) {
    switch (epilogue) {
        case EPI_F16: return launch_v8<EPI_F16>(p, stream);
        case EPI_F16_GELU: return launch_v8<EPI_F16_GELU>(p, stream);
        case EPI_F32: return launch_v8<EPI_F32>(p, stream);
        case EPI_RESID: return launch_v8<EPI_RESID>(p, stream);
        case EPI_PATCH: return launch_v8<EPI_PATCH>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace cgpt
