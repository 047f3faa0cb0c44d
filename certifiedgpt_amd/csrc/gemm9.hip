// gemm9.hip -- 256 x 256 x 64 fp16 MFMA GEMM: two 32-MFMA phases per K-tile, operand PARTS requested 1.5 K-tiles ahead by LDS-DMA
// behind counted vmcnt waits.  The library's kernel for every large shape (M >= 1024, more than 128 tiles): all four ViT GEMMs.
//
// Tile, LDS image, swizzle, transposed accumulators, persistent XCD-aware tile walk, 192-column last tiles and LDS-transposed fp16
// epilogue are those of gemm3_f16_kernel (gemm.hip); every kernel gives the same bits (tests/test_gpu_kernels.py).
//
// K loop.  The 128 x 64 wave tile is four quadrants over the whole K = 64 of a K-tile; a stage (64 KiB: A image then W image, 128-byte
// rows) is four PARTS of 16 KiB = two 1-KiB requests per wave:
//     A(m0), A(m1) = first / second 64 rows of the wave's 128;   B(n0), B(n1) = first / second 32 columns of its 64 (2 / 1 tiles of 16
//     columns in a 192-column tile).
// A K-tile is two phases, each an L segment (fragment reads + requests) and an M segment (32 MFMAs and -- see below -- nothing else
// that touches memory); the two halves of the workgroup (waves 0-3 / 4-7 = the two waves of each SIMD) run one barrier-delimited slot
// apart, so one wave of a SIMD computes while its partner reads:
//     L(P0): reads A(m0) [8 fragments], B(n0) [4], B(n1) [4];  requests A(m1) of K-tile c+1
//     M(P0): quadrants (m0,n0), (m0,n1)
//     L(P1): reads A(m1) [8];  requests A(m0), B(n0), B(n1) of K-tile c+2 (all three were last read in L(P0))
//     M(P1): quadrants (m1,n1), (m1,n0)
// Every part is requested 6 slots = 1.5 K-tiles before its first read.  Both L segments end with a COUNTED s_waitcnt vmcnt(N) (never 0
// in steady state: the four younger parts stay in flight), lgkmcnt(0) and the barrier, so a part is retired by EVERY wave at least one
// barrier before any wave reads it, and re-requested at least one barrier after every wave's reads of it have returned.
// The first K-tile of a tile multiplies into a ZERO accumulator operand instead of clearing 128 registers (round 3: -1.5 ... -2 % on the
// K = 1408 shapes, raw harness).
//
// Round 4 (profiles/r04/gemm_kloop.txt): ablation builds (-DCGPT_ABL) put a K-tile at 2 628 cycles against 2 031 with neither requests nor
// fragment reads (2 193 / 2 252 with one of them): the L segments are shorter than their partner's M segment in every build, the loss is what a
// partner's LDS reads and LDS-DMA traffic cost the wave that issues the MFMAs.  Fragment reads UNDER the M segments (A(m1) and the next K-tile's
// B read into registers as their last MFMA is issued, counted waits at the ends of the M segments; bit-identical, race screen clean) were 13 %
// SLOWER on every shape: a ds_read_b128 between MFMAs costs the issuing wave ~19 cycles of matrix-pipe time.  Removed again.
//
// Tried and removed in round 3 (profiles/r03/gemm_deferred_gelu.txt):
//  * finishing fc1's GELU inside the NEXT tile's K loop -- raw fp16 tile written by a plain epilogue, re-read in 1-KiB pieces by LDS-DMA into
//    the idle epilogue scratch behind exactly re-counted waits, GELU in the M segments, stored back.  Bit-identical, but 1 215 us per launch
//    against 1 050 for the fused epilogue: 32 more in-order memory operations per tile in the L segments cost +266 us (an L segment is the
//    critical path of its slot: every request it issues delays the barrier its partner's MFMAs wait for), the arithmetic in the M
//    segments itself next to nothing.
//  * requests as `buffer_load_dwordx4 ... offen lds` (one 32-bit per-lane offset per piece + a wave-uniform soffset: 6 VGPRs instead of
//    eight 64-bit pointers, 247 -> 220 VGPRs): bit-identical, but the K loop is ~2 % slower (fc2 864 vs 846 us raw; 9.81 vs 10.00 img/s
//    in the model on one box).
#include "gemm_common.h"

namespace cgpt {

namespace {

#ifdef CGPT_PLAIN_STORES
#define CGPT9_STORE16(v, ptr) (*(ptr) = (v))
#else
#define CGPT9_STORE16(v, ptr) __builtin_nontemporal_store((v), (ptr))
#endif

constexpr int kMaxDevices9 = 64;
template <int V> struct IntTag9 { static constexpr int value = V; };


template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm9_f16_kernel(GemmParams p) {
    constexpr int BM2 = 256, BN_ = 256;
    constexpr int A_ELEMS = BM2 * BK, STAGE = 2 * A_ELEMS;                 // halfs: A image then W image, 128-byte rows
    constexpr bool GELU = EPI == EPI_F16_GELU;
    extern __shared__ __attribute__((aligned(16))) half_t smem9[];
    half_t* const scratch_all = smem9 + 2 * STAGE;

    // in-kernel clock (cgpt_profile_clock): two scalar counter reads here and at the end, d(s_memtime) / d(s_memrealtime) x 100 MHz
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const bool late = wave >= 4;                                           // the half that runs one slot behind
    const int r15 = lane & 15, g = lane >> 4;

    const int tiles_m = (p.M + BM2 - 1) / BM2;
    const int tiles_n = (p.N + BN_ - 1) / BN_;
    const int ntiles = tiles_m * tiles_n;
    const int nk = p.K / BK;
    const bool split_n = (p.N % 256) == 128 && p.N >= 384 && !(p.ablate & 16384);

    // ------------------------------------------------------------------ request side (runs ~1.5 K-tiles ahead of the MFMAs)
    // piece i (0, 1) of this wave covers part rows 16*(wave&3 | wave&1) + 8*i + (lane>>3); source 16-byte chunk swizzled per row.
    // A(m1) is A(m0) + 64 rows and (wide tiles) B(n1) is B(n0) + 32 rows: the swizzle (row >> 1) & 7 is the same, so they differ by a
    // wave-uniform offset.  All offsets are bytes.
    const int lr = lane >> 3, cpos = lane & 7;
    const half_t* src_a0[2];   // A(m0): tile rows (wave>>2)*128 +  0 + 16*(wave&3) + 8*i + lr
    const half_t* src_a1[2];   // A(m1):                        + 64
    const half_t* src_b0[2];   // B(n0): W rows (wave>>1)*64 + 16*(wave&1) + 8*i + lr          (narrow: (wave>>1)*48 + ...)
    const half_t* src_b1[2];   // B(n1):                    + 32                               (narrow: one piece, (wave>>1)*48 + 32 + 8*(wave&1) + lr)
    int dst_a0, dst_a1, dst_b0[2], dst_b1[2];                               // LDS offsets (halfs) inside a stage, piece 0 (A: piece 1 = + 8 rows)
    int rt = blockIdx.x, rkt = 0, rc = 0;                                   // request cursor: tile, K-tile in it, stream K-tile counter
    bool req_ok = rt < ntiles;
    auto set_req_tile = [&](int t) {
        int tm, tn;
        tile_of_virtual_block(t, ntiles, tiles_m, tiles_n, tm, tn, p.group_m);
        const bool narrow = split_n && tn >= tiles_n - 2;
        const int ncol0 = narrow ? (tiles_n - 2) * BN_ + (tn - (tiles_n - 2)) * 192 : tn * BN_;
        const int ra = (wave >> 2) * 128 + 16 * (wave & 3);
        dst_a0 = ra * BK;
        dst_a1 = (ra + 64) * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r0 = ra + 8 * i + lr, r1 = r0 + 64;
            src_a0[i] = p.A + (int64_t)(tm * BM2 + r0) * p.lda + ((cpos ^ ((r0 >> 1) & 7)) << 3);
            src_a1[i] = p.A + (int64_t)(tm * BM2 + r1) * p.lda + ((cpos ^ ((r1 >> 1) & 7)) << 3);
            // W rows of the two B parts (LDS row == row inside the tile's W image)
            const int b0 = narrow ? (wave >> 1) * 48 + 16 * (wave & 1) + 8 * i : (wave >> 1) * 64 + 16 * (wave & 1) + 8 * i;
            const int b1 = narrow ? (wave >> 1) * 48 + 32 + 8 * (wave & 1) : b0 + 32;   // narrow: both pieces are the same 8 rows (idempotent)
            dst_b0[i] = A_ELEMS + b0 * BK;
            dst_b1[i] = A_ELEMS + b1 * BK;
            src_b0[i] = p.W + (int64_t)(ncol0 + b0 + lr) * p.ldw + ((cpos ^ (((b0 + lr) >> 1) & 7)) << 3);
            src_b1[i] = p.W + (int64_t)(ncol0 + b1 + lr) * p.ldw + ((cpos ^ (((b1 + lr) >> 1) & 7)) << 3);
        }
    };
#ifndef CGPT_ABL
#define CGPT_ABL 0   // timing-only experiment builds (WRONG results): 1 = no LDS-DMA requests, 2 = fragment reads only in a tile's first K-tile
#endif
    auto glds = [&](const half_t* src, int lds_off) {
        if constexpr ((CGPT_ABL & 1) != 0) return;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem9 + lds_off), 16, 0, 0);
    };
    // PART: 0 = A(m0), 1 = B(n0), 2 = B(n1), 3 = A(m1); the four parts of one K-tile are requested consecutively in this order
    auto request = [&](auto part_tag) __attribute__((always_inline)) {
        constexpr int PART = decltype(part_tag)::value;
        if (!req_ok) return;
        const int sbase = (rc & 1) * STAGE;
        const int koff = rkt * BK;
        if constexpr (PART == 0) { glds(src_a0[0] + koff, sbase + dst_a0); glds(src_a0[1] + koff, sbase + dst_a0 + 8 * BK); }
        if constexpr (PART == 1) { glds(src_b0[0] + koff, sbase + dst_b0[0]); glds(src_b0[1] + koff, sbase + dst_b0[1]); }
        if constexpr (PART == 2) { glds(src_b1[0] + koff, sbase + dst_b1[0]); glds(src_b1[1] + koff, sbase + dst_b1[1]); }
        if constexpr (PART == 3) {
            glds(src_a1[0] + koff, sbase + dst_a1); glds(src_a1[1] + koff, sbase + dst_a1 + 8 * BK);
            ++rc;                                                           // K-tile complete: advance the cursor
            if (++rkt == nk) {
                rkt = 0;
                rt += gridDim.x;
                req_ok = rt < ntiles;
                if (req_ok) set_req_tile(rt);
            }
        }
    };

    // ------------------------------------------------------------------ compute side
    const int sw = (r15 >> 1) & 7;
    const int k_off0 = ((g ^ sw) << 3), k_off1 = (((4 + g) ^ sw) << 3);
    const int a_rd = (wr * 128 + r15) * BK;
    const int b_rd_wide = A_ELEMS + (wc * 64 + r15) * BK, b_rd_narrow = A_ELEMS + (wc * 48 + r15) * BK;

    f32x4 acc[8][4];
    f16x8 af[4][2], bf0[2][2], bf1[2][2];
#define CGPT_FENCE __builtin_amdgcn_sched_barrier(0);
#define CGPT_SLOT_END CGPT_FENCE __builtin_amdgcn_s_barrier(); CGPT_FENCE

    int c = 0;                                                              // stream K-tile counter of the compute side
    int t = blockIdx.x;
    if (req_ok) {
        set_req_tile(rt);
        // prime the stream: all of K-tile 0 and the first three parts of K-tile 1 (L(P0) of K-tile 0 then requests A(m1) of 1)
        request(IntTag9<0>{}); request(IntTag9<1>{}); request(IntTag9<2>{}); request(IntTag9<3>{});
        request(IntTag9<0>{}); request(IntTag9<1>{}); request(IntTag9<2>{});
    }
    half_t* const scr = scratch_all + wave * 2048;                          // 4 KiB of epilogue scratch per wave
    float* const bias_lds = reinterpret_cast<float*>(scr);
    // K-tile 0 of the first tile: A(m0), B(n0), B(n1) landed for every wave (the younger parts stay in flight, as in steady state)
    if (!req_ok) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    CGPT_SLOT_END

#ifdef CGPT_STAMPS
    unsigned long long st_first = 0, st_loop = 0, st_epi = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
    // phase stamps of the K loop (cycles summed over all K-tiles): 0 L(P0) incl. its waits, 1 barrier, 2 M(P0), 3 barrier, 4 L(P1), 5 barrier,
    // 6 M(P1), 7 barrier.  An s_memtime is issued in order: the M stamps are taken when the last MFMA has ISSUED, not completed.
    unsigned long long ph9[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph_last = 0;
#define CGPT9_PH_BEGIN ph_last = __builtin_amdgcn_s_memtime();
#define CGPT9_PH(k) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph9[k] += tn_ - ph_last; ph_last = tn_; }
#else
#define CGPT9_PH_BEGIN
#define CGPT9_PH(k)
#endif
// The 8 x JN MFMAs of one half of an M segment: rows R0 + 0..3 of the wave tile against JN column tiles (fragments BF), both k-steps.  An
// accumulator sees k-step 0 before k-step 1 in either order (the K order per element is what bit-identity rests on).
//   CGPT_MFMA_ORDER 0: row-tile-major -- neighbours share the A fragment in pairs, every other transition shares nothing.
//   CGPT_MFMA_ORDER 1: column-tile-major with the rows walked back and forth -- every transition inside a k-step keeps one of the two
//                      fragment operands (a bare MFMA loop sustains 3 % more with one operand kept than with none: profiles/r04/mfma_sustained.txt).
#ifndef CGPT_MFMA_ORDER
#define CGPT_MFMA_ORDER 0
#endif
#if CGPT_MFMA_ORDER == 0
#define CGPT9_MM(R0, BF, JN, C0)                                                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < JN; ++j)   \
        acc[R0 + i][C0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(BF[j][ks], af[i][ks], (Z && ks == 0) ? zero4 : acc[R0 + i][C0 + j], 0, 0, 0);
#else
#define CGPT9_MM(R0, BF, JN, C0)                                                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < JN; ++j) _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) { \
        const int i = (j & 1) ? 3 - ii : ii;                                                                                             \
        acc[R0 + i][C0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(BF[j][ks], af[i][ks], (Z && ks == 0) ? zero4 : acc[R0 + i][C0 + j], 0, 0, 0); }
#endif
    auto tile_body = [&](auto tnv_tag, int tm, int ncol0) __attribute__((always_inline)) {
        constexpr int TNv = decltype(tnv_tag)::value;                       // column tiles of 16 per wave: 4, or 3 (192-column tile)
        constexpr bool NARROW = TNv == 3;
        constexpr int N1 = TNv - 2;                                         // column tiles of the n1 half
        const int b_rd = NARROW ? b_rd_narrow : b_rd_wide;
        const int wcols = NARROW ? 48 : 64;
        if (p.bias) {                                                       // 64 bias values of this wave -> its scratch
            const int bc = min(ncol0 + wc * wcols + lane, p.N - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + bc),
                                             (__attribute__((address_space(3))) void*)bias_lds, 4, 0, 0);
        }
        if (late) { CGPT_SLOT_END }                                         // the late half enters one slot behind
#ifdef CGPT_STAMPS
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
        unsigned long long ts_k1 = 0;
#endif

        // One K-tile; two copies of this body: the first K-tile of a tile (Z: zero accumulator operand) and every other one.
        auto wait_vm = [&]() __attribute__((always_inline)) {               // the four younger parts stay in flight
            if (!req_ok) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        };
        auto ktile = [&](auto zero_tag) __attribute__((always_inline)) {
            constexpr bool Z = decltype(zero_tag)::value != 0;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const half_t* st = smem9 + (c & 1) * STAGE;
            CGPT9_PH_BEGIN
            constexpr bool RD = Z || !(CGPT_ABL & 2);
            // ---------------- P0 = (m0; n0, n1)
            if constexpr (RD) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf0[j][0] = *reinterpret_cast<const f16x8*>(st + b_rd + j * 16 * BK + k_off0);
                bf0[j][1] = *reinterpret_cast<const f16x8*>(st + b_rd + j * 16 * BK + k_off1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *reinterpret_cast<const f16x8*>(st + a_rd + i * 16 * BK + k_off0);
                af[i][1] = *reinterpret_cast<const f16x8*>(st + a_rd + i * 16 * BK + k_off1);
            }
#pragma unroll
            for (int j = 0; j < N1; ++j) {
                bf1[j][0] = *reinterpret_cast<const f16x8*>(st + b_rd + (2 + j) * 16 * BK + k_off0);
                bf1[j][1] = *reinterpret_cast<const f16x8*>(st + b_rd + (2 + j) * 16 * BK + k_off1);
            }
            }
            CGPT_FENCE
            request(IntTag9<3>{});                                          // A(m1) of K-tile c+1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vm();                                                      // A(m1) of this K-tile has landed
            CGPT9_PH(0)
            CGPT_SLOT_END
            CGPT9_PH(1)
            CGPT9_MM(0, bf0, 2, 0)
            CGPT9_MM(0, bf1, N1, 2)
            CGPT9_PH(2)
            CGPT_SLOT_END
            CGPT9_PH(3)
            // ---------------- P1 = (m1; n1, n0)
            if constexpr (RD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *reinterpret_cast<const f16x8*>(st + a_rd + (4 + i) * 16 * BK + k_off0);
                af[i][1] = *reinterpret_cast<const f16x8*>(st + a_rd + (4 + i) * 16 * BK + k_off1);
            }
            }
            CGPT_FENCE
            request(IntTag9<0>{});                                          // A(m0), B(n0), B(n1) of K-tile c+2: all last read in L(P0)
            request(IntTag9<1>{});
            request(IntTag9<2>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vm();                                                      // A(m0), B(n0), B(n1) of K-tile c+1 have landed
            CGPT9_PH(4)
            CGPT_SLOT_END
            CGPT9_PH(5)
            CGPT9_MM(4, bf1, N1, 2)
            CGPT9_MM(4, bf0, 2, 0)
            CGPT9_PH(6)
            CGPT_SLOT_END
            CGPT9_PH(7)
            ++c;
        };

        ktile(IntTag9<1>{});
#ifdef CGPT_STAMPS
        ts_k1 = __builtin_amdgcn_s_memtime();
#endif
        for (int kt = 1; kt < nk; ++kt) ktile(IntTag9<0>{});

#ifdef CGPT_STAMPS
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        // the early half waits one slot for its partners' last M: both waves of a SIMD then run their epilogues TOGETHER (one wave
        // alone issues VALU at half the SIMD's rate)
        if (!late) { CGPT_SLOT_END }
        // ------------------------------------------------------------ epilogue
        int el;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));   // lane id, not kept across the K loop
        const int e15 = el & 15, eg = el >> 4;
        if (nk < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // too few K-tiles for the counted waits to have retired the bias request
        f32x4 bias4[TNv];
#pragma unroll
        for (int j = 0; j < TNv; ++j)
            bias4[j] = p.bias ? *reinterpret_cast<const f32x4*>(bias_lds + j * 16 + 4 * eg) : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // before the first pass overwrites the scratch
        const bool full = (tm + 1) * BM2 <= p.M && ncol0 + (NARROW ? 192 : BN_) <= p.N;
        constexpr bool F16_OUT = EPI == EPI_F16 || EPI == EPI_F16_GELU;
        if (F16_OUT && full && (p.ldo & 7) == 0 && !(p.ablate & 512)) {
            // fp16 output of a full tile, transposed through the wave's 4-KiB scratch: four passes of 32 rows x 64 columns; rows are
            // 128 B with the 16-byte chunk index XOR-swizzled by row & 7; a lane then stores 16 contiguous bytes, 8 lanes one line
            half_t* outp = reinterpret_cast<half_t*>(p.out);
            const int row_rd = el >> 3, ch_rd = el & 7;
            {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    // The two waves of a SIMD share its VALU by age: the older (early) half finishes its epilogue first and the younger
                    // then runs the rest alone, at a single wave's issue rate (in-kernel stamps, fc1: 8.4 vs 11.4 us).  Priority for the
                    // younger half during the FIRST two passes evens them out: fc1 -0.9 %, qkv -0.7 %, +0.35 % end to end (three
                    // interleaved runs on one box; all four passes, one pass or three passes: neutral or worse;
                    // profiles/r03/gemm_deferred_gelu.txt).
                    if (ps == 0) { if (late) __builtin_amdgcn_s_setprio(1); }
                    if (ps == 2) { if (late) __builtin_amdgcn_s_setprio(0); }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int i = ps * 2 + ii;
                        const int row = ii * 16 + e15;
#pragma unroll
                        for (int jj = 0; jj < TNv; ++jj) {
                            const f32x4 v = acc[i][jj] + bias4[jj];
                            f16x4 hv;
                            if constexpr (GELU) hv = gelu_f16x4(v);
                            else hv = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                            const int ch = (jj * 2 + (eg >> 1)) ^ (row & 7);
                            *reinterpret_cast<f16x4*>(scr + row * 64 + ch * 8 + (eg & 1) * 4) = hv;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same wave wrote what it now reads
                    half_t* dst0 = outp + ((int64_t)tm * BM2 + wr * 128 + ps * 32 + row_rd) * p.ldo + ncol0 + wc * wcols + ch_rd * 8;
                    const int64_t step = 8 * p.ldo;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 8 + row_rd;
                        if (!NARROW || ch_rd < 6) {
                            const f16x8 o = *reinterpret_cast<const f16x8*>(scr + row * 64 + ((ch_rd ^ (row & 7)) * 8));
                            CGPT9_STORE16(o, reinterpret_cast<f16x8*>(dst0 + it * step));
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads returned before the next pass overwrites
                }
            }
        } else {
            if constexpr (NARROW) {
                f32x4 accn[8][3];
#pragma unroll
                for (int i2 = 0; i2 < 8; ++i2)
#pragma unroll
                    for (int j2 = 0; j2 < 3; ++j2) accn[i2][j2] = acc[i2][j2];
                gemm_epilogue_256<EPI, 8, 3>(p, accn, bias4, tm * BM2 + wr * 128 + e15, ncol0 + wc * 48 + 4 * eg, full);
            } else {
                gemm_epilogue_256<EPI, 8, 4>(p, acc, bias4, tm * BM2 + wr * 128 + e15, ncol0 + wc * 64 + 4 * eg, full);
            }
        }
#ifdef CGPT_STAMPS
        st_first += ts_k1 - ts0; st_loop += ts2 - ts0; st_epi += __builtin_amdgcn_s_memtime() - ts2;
#endif
    };

    for (; t < ntiles; t += gridDim.x) {
        int tm, tn;
        tile_of_virtual_block(t, ntiles, tiles_m, tiles_n, tm, tn, p.group_m);
        const bool narrow = split_n && tn >= tiles_n - 2;
        const int ncol0 = narrow ? (tiles_n - 2) * BN_ + (tn - (tiles_n - 2)) * 192 : tn * BN_;
        if (narrow) tile_body(IntTag9<3>{}, tm, ncol0);
        else tile_body(IntTag9<4>{}, tm, ncol0);
    }
#ifdef CGPT_STAMPS
    // Three tables of FIXED capacity (256 workgroups: one per CU; the grid varies with cgpt_set_option("gemm_grid") and with shapes of
    // fewer tiles than CUs, the readers -- tools/gemm_wg_balance.py, scratch stamp scripts -- index by this capacity and size their
    // buffer as 256 * 8 * 12 + 512 words): [wg][wave][4], then [wg][wave][8] phase stamps, then [wg][begin, end].
    constexpr size_t STAMP_WGS = 256;
    if (p.dbg && lane == 0 && blockIdx.x < STAMP_WGS) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - st_begin; d[1] = st_first; d[2] = st_loop; d[3] = st_epi;
        unsigned long long* e = p.dbg + STAMP_WGS * 8 * 4 + ((size_t)blockIdx.x * 8 + wave) * 8;   // second table: phase stamps
        for (int k = 0; k < 8; ++k) e[k] = ph9[k];
        if (wave == 0) {   // third table: [workgroup][begin, end] in 100-MHz real-time ticks (how evenly the workgroups finish)
            unsigned long long* r = p.dbg + STAMP_WGS * 8 * 12 + (size_t)blockIdx.x * 2;
            r[0] = clk_r0; r[1] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
    if (p.clk && threadIdx.x == 0) {
        atomicAdd(p.clk, __builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(p.clk + 1, __builtin_amdgcn_s_memrealtime() - clk_r0);
    }
#undef CGPT_FENCE
#undef CGPT_SLOT_END
}

template <int EPI>
hipError_t launch_v9(const GemmParams& p, hipStream_t stream) {
    constexpr int lds_bytes = 2 * (256 + 256) * BK * (int)sizeof(half_t) + 32768;   // two stages + epilogue scratch = 160 KiB
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices9) return hipErrorInvalidDevice;
    static bool configured[kMaxDevices9] = {false};
    static int cus[kMaxDevices9] = {0};
    if (!configured[dev]) {
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm9_f16_kernel<EPI>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); e != hipSuccess) return e;
        int n = 0;
        if (hipError_t e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
        cus[dev] = n > 0 ? n : 256;
        configured[dev] = true;
    }
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    int grid = tiles < cus[dev] ? tiles : cus[dev];                         // one 512-thread workgroup per CU (LDS-limited), persistent
    // cgpt_set_option("gemm_grid", n): at most n workgroups (a multiple of 8 keeps the XCD-aware walk), i.e. CUs left free for the
    // kernels of ANOTHER stream (a workgroup of this kernel fills its CU's register file: nothing co-resides with it).  Any grid gives
    // the same bits: a tile's arithmetic does not depend on which workgroup computes it.
    if (g_gemm_grid > 0 && grid > g_gemm_grid) grid = g_gemm_grid;
    hipLaunchKernelGGL((gemm9_f16_kernel<EPI>), dim3(grid), dim3(512), lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_v9_epi(int epilogue, const GemmParams& p, hipStream_t stream) {
    switch (epilogue) {
        case EPI_F16: return launch_v9<EPI_F16>(p, stream);
        case EPI_F16_GELU: return launch_v9<EPI_F16_GELU>(p, stream);
        case EPI_F32: return launch_v9<EPI_F32>(p, stream);
        case EPI_RESID: return launch_v9<EPI_RESID>(p, stream);
        case EPI_PATCH: return launch_v9<EPI_PATCH>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace cgpt
