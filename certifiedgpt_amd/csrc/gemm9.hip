// gemm9.hip -- 256 x 256 x 64 fp16 MFMA GEMM: two 32-MFMA phases per K-tile, operand PARTS requested 1.5 K-tiles ahead by LDS-DMA
// behind counted vmcnt waits, and (round 3) the GELU of the ViT MLP's fc1 moved out of the epilogue into the NEXT tile's K loop.
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
// Requests are `buffer_load_dwordx4 ... offen lds`: one 32-bit per-lane offset per piece (6 VGPRs in all) + a wave-uniform offset for
// tile, K-tile and part, instead of eight 64-bit per-lane pointers.  The first K-tile of a tile multiplies into a ZERO accumulator
// operand instead of clearing 128 registers.
//
// Deferred GELU (EPI_F16_GELU, nn.GELU after Mlp.fc1, eva_vit.py:59-61).  128 outputs per lane and tile cost the two waves of a SIMD
// ~7 k cycles of VALU issue with the matrix pipe idle (in-kernel stamps, profiles/r02/gemm_stamps.txt: epilogue 10 us of a 55-us tile
// against 2.8-3.6 us for a plain one), and neither registers (247 of 256) nor LDS (160 of 160 KiB) can hold a tile's outputs across
// the next tile.  So a full tile that has a successor writes h = fp16(acc + bias) -- the tensor the reference materialises between
// `fc1` and `act` under autocast -- with the plain epilogue, and during K-tiles 1..17 of the successor each wave brings its 128 x 64
// block back in 16 pieces of 1 KiB (8 rows; 16 bytes per lane, the lane's own bytes of the epilogue's stores) by LDS-DMA into its idle
// epilogue scratch (L2 / Infinity-Cache hits, `sc1`: not through L1), reads a piece back in two halves, applies GELU to a half inside
// an M segment -- where the VALU is otherwise idle -- and stores the piece (nontemporal).  Piece k is requested in L(P0) of K-tile
// c = k + 1, retired by the counted wait that ends L(P1) of c, its first half computed in M(P1) of c, its second in M(P0) of c + 1, and
// stored in L(P1) of c + 1; two 1-KiB slots alternate.  Loads, stores and LDS-DMA retire in issue order, so the counted waits grow by
// exactly the deferred operations younger than what they wait for (derivation at `ktile`):
//     K-tile    kind      L(P0) requests a piece    L(P1) stores a piece    vmcnt at the end of L(P0), L(P1)
//     0         ZEROC              -                         -                       8, 8      (also every K-tile of a tile with
//     1         FIRST              x                         -                       9, 8       nothing pending)
//     2         SECOND             x                         x                       9, 9
//     3..16     STEADY             x                         x                      10, 9
//     17        LAST               -                         x                       9, 9
//     18        AFTER              -                         -                       9, 8
//     19..      NONE               -                         -                       8, 8
// A smaller count than the exact one is always safe (it waits for more), a larger one never.  The last tile of a workgroup, a partial
// tile, a 192-column tile and any launch with fewer than 18 K-tiles or N % 256 != 0 apply GELU in the epilogue as before; both forms
// evaluate the same arithmetic on the fp16-rounded h (gelu_h4), so an element's value does not depend on the path its tile takes
// (tested bitwise against the fused form: gemm_ablate bit 32768).
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
constexpr int kDeferMinK = 18;                                             // K-tiles a successor needs to carry a deferred tile

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm9_f16_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses gfx950 buffer builtins the host pass of hipcc cannot type-check: the host sees only the stub
    constexpr int BM2 = 256, BN_ = 256;
    constexpr int A_ELEMS = BM2 * BK, STAGE = 2 * A_ELEMS;                 // halfs: A image then W image, 128-byte rows
    constexpr bool GELU = EPI == EPI_F16_GELU;
    extern __shared__ __attribute__((aligned(16))) half_t smem9[];
    half_t* const scratch_all = smem9 + 2 * STAGE;

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
    const bool defer_on = GELU && p.defer_gelu && nk >= kDeferMinK;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.W), 0, -1, 0x00020000);

    // ------------------------------------------------------------------ request side (runs ~1.5 K-tiles ahead of the MFMAs)
    // piece i (0, 1) of this wave covers part rows 16*(wave&3 | wave&1) + 8*i + (lane>>3); source 16-byte chunk swizzled per row.
    // A(m1) is A(m0) + 64 rows and (wide tiles) B(n1) is B(n0) + 32 rows: the swizzle (row >> 1) & 7 is the same, so they differ by a
    // wave-uniform offset.  All offsets are bytes.
    const int lr = lane >> 3, cpos = lane & 7;
    const int ra = (wave >> 2) * 128 + 16 * (wave & 3);
    int va[2], vb0[2], vb1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r0 = ra + 8 * i + lr;
        va[i] = (r0 * (int)p.lda + ((cpos ^ ((r0 >> 1) & 7)) << 3)) * 2;
    }
    const int dst_a0 = ra * BK, dst_a1 = (ra + 64) * BK;                    // LDS offsets (halfs) inside a stage, piece 0 (piece 1 = + 8 rows)
    int dst_b0[2], dst_b1[2];
    unsigned a_off = 0, w_off = 0;                                          // wave-uniform byte offsets of the request tile's A rows / W rows
    int rt = blockIdx.x, rkt = 0, rc = 0;                                   // request cursor: tile, K-tile in it, stream K-tile counter
    bool req_ok = rt < ntiles;
    auto set_req_tile = [&](int t) {
        int tm, tn;
        tile_of_virtual_block(t, ntiles, tiles_m, tiles_n, tm, tn, p.group_m);
        const bool narrow = split_n && tn >= tiles_n - 2;
        const int ncol0 = narrow ? (tiles_n - 2) * BN_ + (tn - (tiles_n - 2)) * 192 : tn * BN_;
        a_off = (unsigned)(tm * BM2) * (unsigned)p.lda * 2u;
        w_off = (unsigned)ncol0 * (unsigned)p.ldw * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // W rows of the two B parts (LDS row == row inside the tile's W image)
            const int b0 = narrow ? (wave >> 1) * 48 + 16 * (wave & 1) + 8 * i : (wave >> 1) * 64 + 16 * (wave & 1) + 8 * i;
            const int b1 = narrow ? (wave >> 1) * 48 + 32 + 8 * (wave & 1) : b0 + 32;   // narrow: both pieces are the same 8 rows (idempotent)
            dst_b0[i] = A_ELEMS + b0 * BK;
            dst_b1[i] = A_ELEMS + b1 * BK;
            vb0[i] = ((b0 + lr) * (int)p.ldw + ((cpos ^ (((b0 + lr) >> 1) & 7)) << 3)) * 2;
            vb1[i] = ((b1 + lr) * (int)p.ldw + ((cpos ^ (((b1 + lr) >> 1) & 7)) << 3)) * 2;
        }
    };
    auto glds = [&](const __amdgpu_buffer_rsrc_t& rs, int voff, unsigned soff, int lds_off) __attribute__((always_inline)) {
        // soff is wave-uniform; saying so keeps hipcc from wrapping the request in a readfirstlane (waterfall) loop
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem9 + lds_off), 16, voff,
                                                 __builtin_amdgcn_readfirstlane((int)soff), 0, 0);
    };
    // PART: 0 = A(m0), 1 = B(n0), 2 = B(n1), 3 = A(m1); the four parts of one K-tile are requested consecutively in this order
    auto request = [&](auto part_tag) __attribute__((always_inline)) {
        constexpr int PART = decltype(part_tag)::value;
        if (!req_ok) return;
        const int sbase = (rc & 1) * STAGE;
        const unsigned koff = (unsigned)rkt * (BK * 2);
        if constexpr (PART == 0) { glds(rs_a, va[0], a_off + koff, sbase + dst_a0); glds(rs_a, va[1], a_off + koff, sbase + dst_a0 + 8 * BK); }
        if constexpr (PART == 1) { glds(rs_w, vb0[0], w_off + koff, sbase + dst_b0[0]); glds(rs_w, vb0[1], w_off + koff, sbase + dst_b0[1]); }
        if constexpr (PART == 2) { glds(rs_w, vb1[0], w_off + koff, sbase + dst_b1[0]); glds(rs_w, vb1[1], w_off + koff, sbase + dst_b1[1]); }
        if constexpr (PART == 3) {
            const unsigned o = a_off + koff + 64u * (unsigned)p.lda * 2u;
            glds(rs_a, va[0], o, sbase + dst_a1); glds(rs_a, va[1], o, sbase + dst_a1 + 8 * BK);
            ++rc;                                                           // K-tile complete: advance the cursor
            if (++rkt == nk) {
                rkt = 0;
                rt += gridDim.x;
                req_ok = rt < ntiles;
                if (req_ok) set_req_tile(rt);
            }
        }
    };

    // ------------------------------------------------------------------ deferred GELU of the previous tile (GELU kernels only)
    // piece k (0..15) of this wave = rows 8k .. 8k+7 of its 128 x 64 output block; lane -> row lane>>3, 16-byte chunk lane&7 (the bytes
    // this lane stored in the epilogue); slot k&1 of the wave's scratch (bytes 1024.. : the first 256 hold the bias), lane-linear
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, -1, 0x00020000);
    unsigned d_off = 0;                                                     // wave-uniform byte offset of the pending block
    bool pending = false;
    u32x4 dv = {0u, 0u, 0u, 0u};                                            // the piece being computed: 8 fp16 values of this lane
    half_t* const dslots = scratch_all + wave * 2048 + 512;                 // two 1-KiB slots behind the bias
    auto lane_id = [&]() __attribute__((always_inline)) {                   // recomputed where needed: not kept across the K loop
        int el;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
        return el;
    };
    auto piece_voff = [&]() __attribute__((always_inline)) { const int el = lane_id(); return ((el >> 3) * (int)p.ldo + (el & 7) * 8) * 2; };
    auto piece_off = [&](int k) { return __builtin_amdgcn_readfirstlane((int)(d_off + (unsigned)(8 * k) * (unsigned)p.ldo * 2u)); };
    auto dload = [&](int k) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_o, (__attribute__((address_space(3))) void*)(dslots + (k & 1) * 512), 16, piece_voff(),
                                                 piece_off(k), 0, /*sc1*/ 16);
    };
    auto dread = [&](int k, int half) __attribute__((always_inline)) {      // this lane's 8 bytes of half `half` of piece k
        return *reinterpret_cast<const f16x4*>(dslots + (k & 1) * 512 + lane_id() * 8 + half * 4);
    };
    auto dstore = [&](int k) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(dv, rs_o, piece_voff(), piece_off(k), /*nt*/ 2);
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto dgelu = [&](f16x4 h, int half) __attribute__((always_inline)) {
#ifdef CGPT9_NO_DGELU
        const u32x2 r = __builtin_bit_cast(u32x2, h);
#else
        const u32x2 r = __builtin_bit_cast(u32x2, gelu_h4(h));
#endif
        dv[2 * half] = r[0]; dv[2 * half + 1] = r[1];
    };
    f16x4 d_in = {0, 0, 0, 0};                                              // a half read in an L segment, computed in the next M segment

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

        // One K-tile.  vmcnt bookkeeping (loads, stores and LDS-DMA retire in issue order): L(P0) of K-tile c issues l_c piece requests
        // (0 or 1) and then 2 operand requests, L(P1) issues s_c piece stores (0 or 1) and then 6 operand requests.  The wait that ends
        // L(P0) of c must retire A(m1) of c = the last 2 operations of L(P0) of c-1: it may leave s_(c-1) + 6 + l_c + 2 in flight.  The
        // wait that ends L(P1) of c must retire A(m0), B(n0), B(n1) of c+1 = the last 6 of L(P1) of c-1, and the piece requested first in
        // L(P0) of c: it may leave 2 + s_c + 6 in flight (the same count when there is no piece: l_c = 0).
        // There are only TWO copies of this body: the first K-tile of a tile (Z: zero accumulator operand, never any deferred work) and
        // the generic one, whose deferred memory operations (ld: request piece kt-1, st: store piece kt-2) and wait counts (n0, n2) are
        // wave-uniform RUN-TIME values.  (One copy per row of the table above made the register allocator give the accumulators
        // different registers in different copies and need 16 more for the permutations between them: spills, i.e. scratch
        // loads inside the counted waits.)  The generic copy's GELU arithmetic is unconditional -- on stale scratch bytes when nothing
        // is pending -- so that its M segments stay straight-line code the compiler can interleave with the MFMAs; only memory
        // operations are predicated.
        auto wait_vm = [&](int n) __attribute__((always_inline)) {
            if (!req_ok) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (!GELU || n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        };
        auto ktile = [&](auto zero_tag, int kt, bool ld, bool st_piece, int n0, int n2) __attribute__((always_inline)) {
            constexpr bool Z = decltype(zero_tag)::value != 0;
            constexpr bool DG = GELU && !Z;                                 // this copy carries the deferred-GELU arithmetic
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const half_t* st = smem9 + (c & 1) * STAGE;
            // ---------------- P0 = (m0; n0, n1)
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
            CGPT_FENCE
            if constexpr (DG) {
                d_in = dread(kt, 1);                                        // second half of piece kt-2 (slot parity of kt)
                if (ld) dload(kt - 1);
                CGPT_FENCE                                                  // the counts below assume this issue order
            }
            request(IntTag9<3>{});                                          // A(m1) of K-tile c+1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vm(n0);                                                    // A(m1) of this K-tile has landed
            CGPT_SLOT_END
            if constexpr (DG) dgelu(d_in, 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf0[j][ks], af[i][ks], (Z && ks == 0) ? zero4 : acc[i][j], 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < N1; ++j)
                        acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf1[j][ks], af[i][ks], (Z && ks == 0) ? zero4 : acc[i][2 + j], 0, 0, 0);
            CGPT_SLOT_END
            // ---------------- P1 = (m1; n1, n0)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *reinterpret_cast<const f16x8*>(st + a_rd + (4 + i) * 16 * BK + k_off0);
                af[i][1] = *reinterpret_cast<const f16x8*>(st + a_rd + (4 + i) * 16 * BK + k_off1);
            }
            CGPT_FENCE
            if constexpr (DG) { if (st_piece) dstore(kt - 2); CGPT_FENCE }
            request(IntTag9<0>{});                                          // A(m0), B(n0), B(n1) of K-tile c+2: all last read in L(P0)
            request(IntTag9<1>{});
            request(IntTag9<2>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vm(n2);                                                    // A(m0), B(n0), B(n1) of K-tile c+1 (and piece kt-1) have landed
            if constexpr (DG) d_in = dread(kt - 1, 0);                      // first half; this wave's own request: no barrier needed
            CGPT_SLOT_END
            if constexpr (DG) dgelu(d_in, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < N1; ++j)
                        acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf1[j][ks], af[i][ks], (Z && ks == 0) ? zero4 : acc[4 + i][2 + j], 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf0[j][ks], af[i][ks], (Z && ks == 0) ? zero4 : acc[4 + i][j], 0, 0, 0);
            CGPT_SLOT_END
            ++c;
        };

        ktile(IntTag9<1>{}, 0, false, false, 8, 8);
#ifdef CGPT_STAMPS
        ts_k1 = __builtin_amdgcn_s_memtime();
#endif
        {
            const bool pend = GELU && pending;                              // nk >= kDeferMinK then; rows of the table in the header
            for (int kt = 1; kt < nk; ++kt) {
                const bool ld = pend && kt <= 16, stp = pend && kt >= 2 && kt <= 17;
                const int n0 = !pend ? 8 : (kt >= 3 && kt <= 16) ? 10 : kt <= 18 ? 9 : 8;
                const int n2 = stp ? 9 : 8;
                ktile(IntTag9<0>{}, kt, ld, stp, n0, n2);
            }
            pending = false;
        }

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
            // GELU of this tile is left to the successor's K loop when there is one (wide tiles only: a piece is 16 lanes x 8 bytes)
            const bool defer_this = GELU && !NARROW && defer_on && t + (int)gridDim.x < ntiles;
            auto passes = [&](auto raw_tag) __attribute__((always_inline)) {
                constexpr bool RAW = decltype(raw_tag)::value != 0;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int i = ps * 2 + ii;
                        const int row = ii * 16 + e15;
#pragma unroll
                        for (int jj = 0; jj < TNv; ++jj) {
                            const f32x4 v = acc[i][jj] + bias4[jj];
                            f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                            if constexpr (GELU && !RAW) hv = gelu_h4(hv);
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
                            if constexpr (RAW) *reinterpret_cast<f16x8*>(dst0 + it * step) = o;   // re-read soon: default cache policy
                            else CGPT9_STORE16(o, reinterpret_cast<f16x8*>(dst0 + it * step));
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads returned before the next pass overwrites
                }
            };
            if (defer_this) {
                passes(IntTag9<1>{});
                pending = true;
                d_off = (unsigned)(((int64_t)tm * BM2 + wr * 128) * p.ldo + ncol0 + wc * 64) * 2u;
            } else {
                passes(IntTag9<0>{});
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
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - st_begin; d[1] = st_first; d[2] = st_loop; d[3] = st_epi;
    }
#endif
#undef CGPT_FENCE
#undef CGPT_SLOT_END
#endif  // __HIP_DEVICE_COMPILE__
}

template <int EPI>
hipError_t launch_v9(const GemmParams& p_in, hipStream_t stream) {
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
    GemmParams p = p_in;
    // deferred GELU: wide tiles only (N a multiple of 256), 16-byte output rows, 32-bit offsets into the output
    const int64_t out_bytes = ((int64_t)((p.M + 255) / 256) * 256) * p.ldo * 2;
    p.defer_gelu = EPI == EPI_F16_GELU && (p.N % 256) == 0 && (p.ldo % 8) == 0 && out_bytes < ((int64_t)1 << 32) && !(p.ablate & 32768);
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int grid = tiles < cus[dev] ? tiles : cus[dev];                   // one 512-thread workgroup per CU (LDS-limited), persistent
    hipLaunchKernelGGL((gemm9_f16_kernel<EPI>), dim3(grid), dim3(512), lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace

// 32-bit offsets into A and W (buffer addressing): the caller falls back to the phased kernel for operands of 4 GiB or more
bool v9_fits(const GemmParams& p) {
    const int64_t a_bytes = ((int64_t)((p.M + 255) / 256) * 256) * p.lda * 2, w_bytes = ((int64_t)((p.N + 255) / 256) * 256) * p.ldw * 2;
    return a_bytes < ((int64_t)1 << 32) && w_bytes < ((int64_t)1 << 32) && p.lda * 640 < ((int64_t)1 << 31) && p.ldw * 640 < ((int64_t)1 << 31);
}

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
