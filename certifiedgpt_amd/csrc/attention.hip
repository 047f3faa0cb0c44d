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
//
// Latency structure (round-1 profile: 58 % of wave cycles parked, hipcc had scheduled read -> wait -> 1-2 MFMAs):
//   * K / V fragments travel through explicit register rings (K: 4 key tiles ahead, V: 2 key pairs ahead) with
//     sched_barrier fences, so the compiler's counted lgkmcnt waits leave the younger reads in flight;
//   * staging is fully unrolled: all of a thread's K and V global loads are issued at once; K goes to LDS first, V is
//     written to LDS only after the wave's first QK^T + softmax (its load latency hides under them).
//   * workgroups are persistent (one per CU, LDS-limited) and walk (sample, head) items: the NEXT item's K and V are
//     requested into registers as soon as the current V has been written to LDS, and the next tile's Q one tile ahead, so
//     the HBM phase of item i+1 overlaps the MFMA / softmax phase of item i (before: all CUs staged, then all computed).
// LDS images: K rows are DPAD halfs with NO padding and a chunk swizzle that is conflict-free for gfx950's
// ds_read_b128 lane groups (192-B rows: chunk 4ds+g -> 4ds + (g ^ ((-(row>>2))&3)); 128-B rows: c ^ ((row>>1)&7));
// V rows have stride 224 B / 160 B: the 8 rows a 32-lane half reads by ds_read_b64_tr_b16 land on distinct 32-B windows.
#include "kernels.h"
#include <type_traits>
#include <utility>

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

// The same read as an instruction the compiler cannot see into, for the streaming kernel.  hipcc's wait-count pass treats the tr-read
// builtin (no memory operand) as a possible reader of every LDS-DMA request in flight and puts s_waitcnt vmcnt(0) in front of it: in
// rounds 2-5 the V reads of a chunk's units 1 and 2 each waited until the requests for the NEXT chunk -- issued one unit earlier, into
// the other buffer -- had landed (two memory latencies per chunk and wave; plain ds_read_b128 carry alias information and get no such
// wait).  The kernel's own counted waits and chunk barriers are what orders requests and reads.  The compiler does not count these
// reads in lgkmcnt either: an `s_waitcnt lgkmcnt(0)` tied to their registers stands in front of their first use; the compiler's own
// counted waits stay safe (LDS returns in order, uncounted reads only make a wait longer).
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression (instruction offsets)
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// the same with `break`: f returns false to leave the loop (later iterations are not entered: the control flow of an unrolled loop)
template <class F, int... I> __device__ __forceinline__ void static_for_while_impl(F&& f, std::integer_sequence<int, I...>) {
    (void)(f(std::integral_constant<int, I>{}) && ...);
}
template <int N, class F> __device__ __forceinline__ void static_for_while(F&& f) { static_for_while_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
template <int OFFSET_BYTES> __device__ __forceinline__ f16x4 lds_read_tr16_untracked(unsigned lds_addr) {
    static_assert(OFFSET_BYTES >= 0 && OFFSET_BYTES < 65536, "16-bit instruction offset");
    f16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFFSET_BYTES));
    return v;
}

template <int DPAD> struct AttnLayout {
    static constexpr int KROW = DPAD;                       // halfs, swizzled, unpadded
    static constexpr int VSTR = (DPAD == 96) ? 112 : 80;    // halfs
};

template <int DPAD> __device__ __forceinline__ int k_chunk_pos(int row, int ch) {
    if constexpr (DPAD == 96) return (ch & ~3) | ((ch & 3) ^ ((0 - (row >> 2)) & 3));
    else return ch ^ ((row >> 1) & 7);
}

#define CGPT_FENCE __builtin_amdgcn_sched_barrier(0);
#ifndef CGPT_ATT_PLAIN_WALK
#define CGPT_ATT_PLAIN_WALK 0       // A/B builds: 1 = the round-robin (sample, head) walk of rounds 1-2
#endif
#ifndef CGPT_ATT_ABLATE
#define CGPT_ATT_ABLATE 0           // timing studies of the streaming kernel ONLY (wrong results): 1 no exp, 2 V fragments of a chunk's first
#endif                              // unit only, 4 K fragments not re-read, 8 one of twelve P.V MFMAs, 16 QK^T of a chunk's first units only,
                                    // 32 requests for the block's first chunk only, 64 no maximum / rescale check, 128 no chunk barrier
#ifndef CGPT_ATT_LONE_CARRIED
#define CGPT_ATT_LONE_CARRIED 1     // A/B builds: 0 = rounds 2-5, a query block of its own for the lone query of Tq = 256 k + 1
#endif

// maximum over the four 16-lane groups of a wave.  The swaps' results go through scalars: __builtin_bit_cast applied to an ELEMENT of the
// returned vector reads element 0 for both (hipcc 7.2), which silently reduced the maximum over lane group 0 only.
__device__ __forceinline__ float max_over_lane_groups(float m) {
    const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    const unsigned a0 = a[0], a1 = a[1];
    m = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
    const auto c2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    const unsigned c0 = c2[0], c1 = c2[1];
    return fmaxf(__builtin_bit_cast(float, c0), __builtin_bit_cast(float, c1));
}

// HD: head_dim (88 | 64); DPAD: HD rounded up to 32; NKT: 16-key tiles held (keys padded to NKT*16); NT: threads.
template <int HD, int DPAD, int NKT, int NT>
__global__ __launch_bounds__(NT) void attention_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int KROW = AttnLayout<DPAD>::KROW, VSTR = AttnLayout<DPAD>::VSTR;
    constexpr int TKP = NKT * 16;
    constexpr int CH = DPAD / 8, NDS = DPAD / 32, NDT = DPAD / 16;
    constexpr int NV = (TKP * CH + NT - 1) / NT;            // staging items per thread
    constexpr int NWAVES = NT / 64;
    half_t* Ks = reinterpret_cast<half_t*>(smem_raw);
    half_t* Vs = Ks + TKP * KROW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r15 = lane & 15, g = lane >> 4;
    const int nitems = p.heads * p.B;
    // Walk position -> (sample, head).  The 16 heads of a ViT sample are 176-byte column blocks of the same 8 448-byte qkv rows, so
    // neighbouring heads share the 128-byte lines at their borders, and a head's rows are 2-3 lines of 176 useful bytes each: dealt
    // round-robin, neighbouring heads land on DIFFERENT XCDs (workgroups b and b + 8 share one) and every XCD's L2 fetches those lines
    // for itself -- 790 MB of reads per launch against 543 MB of Q + K + V (PMC, profiles/r03/pmc_summary.json).  With a grid that is a
    // multiple of 8, XCD x = position % 8 takes the samples x, x + 8, ... and walks each one's heads on consecutive workgroups of its
    // own, so a sample's lines are fetched once per launch.  (Speed only: any placement gives the same result.)
    const bool xcd_walk = (gridDim.x & 7) == 0 && p.B >= 64 && !CGPT_ATT_PLAIN_WALK;   // (few samples: whole samples per XCD would not balance)
    auto item_valid = [&](int it) { return xcd_walk ? ((it >> 3) / p.heads) * 8 + (it & 7) < p.B : it < nitems; };
    auto item_b = [&](int it) { return xcd_walk ? ((it >> 3) / p.heads) * 8 + (it & 7) : it / p.heads; };
    auto item_h = [&](int it) { return xcd_walk ? (it >> 3) % p.heads : it % p.heads; };

    // ---- staging registers: this thread's NV 16-byte pieces of K (or V) of one (sample, head) item.
    // Loads are UNCONDITIONAL from clamped addresses (a guarded load becomes its own basic block with a vmcnt wait at the
    // join, which serialises the whole staging); the zero fill is a select applied when the piece is written to LDS.
    // ONE register set serves both operands alternately: K of the next item is requested while the current item computes,
    // V of the current item right after its K has gone to LDS (it lands under the first QK^T + softmax).
    f16x8 sreg[NV];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // (uniform 64-bit base + 32-bit lane offset: the same NV offsets serve K and V -- launch_attention checks ldk == ldv -- and
    // take half the registers of per-lane pointers, which the compiler used to spill: a scratch reload in front of a request
    // waits for EVERY load in flight, e.g. for the V loads issued just before the first QK^T)
    auto request_kv = [&](int item, bool want_v) {
        const int hh = item_h(item), bb = item_b(item);
        const half_t* G = (want_v ? p.V : p.K) + (int64_t)bb * p.kv_batch_stride + hh * HD;
        const int ldg = (int)p.ldk;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int idx = tid + it * NT;
            const int row = min(idx / CH, p.Tk - 1), ch = min(idx % CH, HD / 8 - 1);
            sreg[it] = *reinterpret_cast<const f16x8*>(G + (row * ldg + ch * 8));
        }
    };

    const float sl2 = p.scale * 1.44269504088896340736f;   // softmax(scale*s) = 2^((s - max) * scale * log2 e) / sum
    const int nqt = (p.Tq + 15) >> 4;
    const half_t* Qb = nullptr;
    half_t* Ob = nullptr;

    f32x4 s[NKT];
    float sum = 1.f;
    f16x8 qf[NDS], qnext[NDS];
    // Q^T B-fragments of query tile qt: lane holds Q[query r15][d = 32*ds + 8*g .. +7] (zero beyond HD)
    auto request_q_from = [&](const half_t* qbase, int qt) {
        int el;                                        // lane id recomputed (volatile): no per-lane address lives across the item loop
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
        const int qoff = min(qt * 16 + (el & 15), p.Tq - 1) * (int)p.ldq;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            const int d = min(ds * 32 + (el >> 4) * 8, HD - 8);
            qnext[ds] = *reinterpret_cast<const f16x8*>(qbase + (qoff + d));
        }
    };
    auto request_q = [&](int qt) { request_q_from(Qb, qt); };
    auto q_base_of = [&](int it) { return p.Q + (int64_t)item_b(it) * p.q_batch_stride + item_h(it) * HD; };
    auto take_q = [&]() {
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) qf[ds] = (ds * 32 + g * 8 < HD) ? qnext[ds] : zero8;
    };

    // S^T = K . Q^T for the query tile in qf, then the softmax numerators in s[] and the row sums in `sum`.
    auto qk_softmax = [&]() {
        // K fragments do not depend on the query tile: an opaque zero keeps the compiler from hoisting them out of the loop
        int opq = 0;
        asm volatile("" : "+v"(opq));
        const half_t* Kq = Ks + opq;
        // this lane's fragment of key tile kt, k-step ds: row kt*16 + r15, chunk 4ds + g (swizzled)
        auto kaddr = [&](int kt, int ds) {
            const int row = kt * 16 + r15;
            return Kq + row * KROW + k_chunk_pos<DPAD>(row, ds * 4 + g) * 8;
        };
        constexpr int KD = (NKT > 2) ? 2 : NKT - 1;                 // key tiles in flight ahead of the MFMAs
        f16x8 kring[KD + 1][NDS];
#pragma unroll
        for (int kt = 0; kt < KD && kt < NKT; ++kt)
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) kring[kt % (KD + 1)][ds] = *reinterpret_cast<const f16x8*>(kaddr(kt, ds));
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt + KD < NKT) {
#pragma unroll
                for (int ds = 0; ds < NDS; ++ds)
                    kring[(kt + KD) % (KD + 1)][ds] = *reinterpret_cast<const f16x8*>(kaddr(kt + KD, ds));
            }
            CGPT_FENCE
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kring[kt % (KD + 1)][ds], qf[ds], acc, 0, 0, 0);
            s[kt] = acc;
            CGPT_FENCE
        }
        // softmax over keys (all keys of a query: this lane's registers x the 4 lanes sharing r15)
        float mx = -1e30f;
        const int klim = p.Tk - 4 * g + opq;          // key (kt*16 + 4g + r) is padding iff kt*16 + r >= klim
        // Only key tiles that reach past Tk need the padding mask.  hipcc turns a per-tile `if (kt * 16 + 16 > Tk)` into a compare + select
        // per register of EVERY tile (72 + 72 VALU instructions per query tile, ~9 % of its vector issue: round 4's instruction count), so
        // the common case -- at most the last two key tiles hold padding, Tk > 16 (NKT - 2): every ViT shape -- is a branch of its own.
        constexpr int NFULL = NKT > 2 ? NKT - 2 : 0;
        if (p.Tk >= NFULL * 16) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt >= NFULL) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt * 16 + r >= klim) s[kt][r] = -1e30f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + r >= klim) s[kt][r] = -1e30f;
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mxs = mx * sl2;
        float sm = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sl2, -mxs));
                s[kt][r] = e;
                if constexpr (DPAD == HD) sm += e;
            }
        if constexpr (DPAD == HD) {
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            sum = sm;
        }
    };

    // O = P . V for query tile qt from s[] / sum, normalised and stored.
    auto pv_store = [&](int qt) {
        int opq = 0;
        asm volatile("" : "+v"(opq));
        const half_t* vbase = Vs + opq + (4 * g + (r15 >> 2)) * VSTR + 4 * (r15 & 3);
        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NU = NKT / 2;
        constexpr int VD = (NU > 1) ? 1 : NU - 1;                   // key pairs in flight ahead of the MFMAs
        f16x4 vring[VD + 1][NDT][2];
        auto vload = [&](int u, int slot) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const half_t* a1 = vbase + (32 * u) * VSTR + dt * 16;
                vring[slot][dt][0] = lds_read_tr16(a1);
                vring[slot][dt][1] = lds_read_tr16(a1 + 16 * VSTR);
            }
        };
#pragma unroll
        for (int u = 0; u < VD && u < NU; ++u) vload(u, u % (VD + 1));
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + VD < NU) vload(u + VD, (u + VD) % (VD + 1));
            CGPT_FENCE
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf[j] = (half_t)s[2 * u][j];
                pf[4 + j] = (half_t)s[2 * u + 1][j];
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const f16x4 v1 = vring[u % (VD + 1)][dt][0], v2 = vring[u % (VD + 1)][dt][1];
                const f16x8 vf = {v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
                // operands swapped: O^T tile, row = d (4g + r), col = query (lane & 15): a lane owns 4 consecutive d of ONE query
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
            }
            CGPT_FENCE
        }
        float den = sum;                               // DPAD == HD: VALU row sum (lane-local for query r15)
        if constexpr (DPAD > HD) {                     // row HD of O^T: d-tile HD/16, g = (HD%16)/4, reg (HD%16)%4
            constexpr int DT = HD / 16, GG = (HD % 16) / 4, RR = (HD % 16) % 4;
            den = __shfl(o[DT][RR], 16 * GG + r15);
        }
        const float inv = 1.0f / den;
        const int q = qt * 16 + r15;
        // A lane owns d = 16 dt + 4g .. +3 of its query: 8 bytes per d-tile, a store instruction would write 16 rows x 32 B.
        // v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of another: applied to the
        // packed fp16 results of two d-tiles (a, b) it leaves lane g with 8 CONSECUTIVE d -- 16 (g&1 ? b : a) + 8 (g>>1) .. +7 --
        // i.e. 16-byte stores, 64-byte segments per row, half the store instructions (the stores cost 35 of this kernel's 235 us).
        static_assert(NDT % 2 == 0, "d-tiles are stored in pairs");
#pragma unroll
        for (int dp = 0; dp < NDT / 2; ++dp) {
            const int da = 2 * dp, db = 2 * dp + 1;
            const f16x4 ha = {(half_t)(o[da][0] * inv), (half_t)(o[da][1] * inv), (half_t)(o[da][2] * inv), (half_t)(o[da][3] * inv)};
            const f16x4 hb = {(half_t)(o[db][0] * inv), (half_t)(o[db][1] * inv), (half_t)(o[db][2] * inv), (half_t)(o[db][3] * inv)};
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 ua = __builtin_bit_cast(u32x2, ha), ub = __builtin_bit_cast(u32x2, hb);
            const auto s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 packed = {s0[0], s1[0], s0[1], s1[1]};           // [new a | new b] = 8 consecutive halfs
            const int d0 = ((g & 1) ? db : da) * 16 + (g >> 1) * 8;
            if (q < p.Tq && d0 < HD) *reinterpret_cast<u32x4*>(Ob + (int64_t)q * p.ldo + d0) = packed;
        }
    };

#ifdef CGPT_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memtime();
#define CGPT_ASTAMP(k) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ph[k] += tn - tlast; tlast = tn; }
#else
#define CGPT_ASTAMP(k)
#endif
    // ---- persistent walk over (sample, head) items
    int item = blockIdx.x;
    if (item_valid(item)) {
        request_kv(item, false);
        if (wave < nqt) request_q_from(q_base_of(item), wave);   // later items: requested at the end of the previous item
    }
    for (; item_valid(item); item += gridDim.x) {
#ifdef CGPT_STAMPS
        unsigned long long tlast = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        const int h = item_h(item), b = item_b(item);
        Qb = p.Q + (int64_t)b * p.q_batch_stride + h * HD;
        Ob = p.O + (int64_t)b * p.o_batch_stride + h * HD;
        int qt = wave;
        const bool have = qt < nqt;
        // K of this item: registers -> swizzled LDS image
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / CH, ch = idx - row * CH;
            const bool valid = row < p.Tk && ch * 8 < HD;
            if (idx < TKP * CH) *reinterpret_cast<f16x8*>(Ks + row * KROW + k_chunk_pos<DPAD>(row, ch) * 8) = valid ? sreg[it] : zero8;
        }
        __syncthreads();
        CGPT_ASTAMP(0)                                   // K -> LDS + barrier
        request_kv(item, true);                          // V of this item; lands under the first QK^T + softmax
        // first query tile of every wave: QK^T + softmax run before V is needed in LDS
        // after a wave has taken its LAST query tile of this item it requests its first tile of the NEXT item: a whole tile of
        // lead (issued at the top of the item the load sat exposed in front of the first QK^T: ~4k of that phase's ~10k cycles)
        const bool more_items = item_valid(item + (int)gridDim.x);
        auto request_following = [&](int cur) {
            if (cur + NWAVES < nqt) request_q(cur + NWAVES);
            else if (more_items) request_q_from(q_base_of(item + gridDim.x), wave);
        };
        if (have) {
            take_q();
            request_following(qt);
            qk_softmax();
        }
        CGPT_ASTAMP(1)                                   // first QK^T + softmax
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / CH, ch = idx - row * CH;
            const bool valid = row < p.Tk && ch * 8 < HD;
            f16x8 vv = valid ? sreg[it] : zero8;
            // the padding column d = HD carries 1.0: the P.V MFMAs then also produce sum_k P[q][k] (the softmax denominator,
            // from the same fp16-rounded P as the numerator) in row HD of O^T -- no VALU sum pass.  Padding keys have P = 0.
            if (DPAD > HD && ch * 8 == HD) vv = f16x8{(half_t)1.0f, 0, 0, 0, 0, 0, 0, 0};
            if (idx < TKP * CH) *reinterpret_cast<f16x8*>(Vs + row * VSTR + ch * 8) = vv;
        }
        __syncthreads();
        CGPT_ASTAMP(2)                                   // V -> LDS (incl. waiting for its loads) + barrier
        // the staging registers are free: request the NEXT item's K now; it lands during the rest of this item
        if (more_items) request_kv(item + gridDim.x, false);
        if (have) pv_store(qt);
        CGPT_ASTAMP(3)                                   // first P.V + store
        for (qt += NWAVES; qt < nqt; qt += NWAVES) {
            take_q();
            request_following(qt);
            qk_softmax();
            pv_store(qt);
        }
        CGPT_ASTAMP(4)                                   // remaining query tiles of this wave
        __syncthreads();                               // every wave is done with this item's LDS images
        CGPT_ASTAMP(5)                                   // waiting for the slowest wave
    }
#ifdef CGPT_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        d[0] = __builtin_amdgcn_s_memtime() - stamp_begin;
        for (int k = 0; k < 6; ++k) d[1 + k] = ph[k];
    }
#endif
}


// ------------------------------------------------------------------------------------------- streaming variant
// Tk > 288 (448^2 images: T = 1025, the reference's own image size, minigpt4.py:32; eva_vit.py:383-404): K and V no longer fit in
// LDS, so a workgroup owns 256 queries of one (sample, head) and streams the keys through LDS with an online softmax.  With the
// transposed product O^T = V^T P^T the query sits on the lane, so the exponent reference, the rescale factor and -- through the ones
// column of V -- the running denominator are all lane-local: a rescale is one multiply per accumulator register.
//   * a wave owns TWO 16-query tiles, so every K and V fragment read from LDS feeds two MFMAs (round 1's form, 8 waves x 16 queries
//     with 288-key chunks, re-read every fragment eight times: 600 of its 968 us at 64 samples x 16 heads x T = 1025 remained once
//     its chunk loads and fragment reads were ablated away, profiles/r02/attention_stream.txt);
//   * K / V chunks of 192 keys are double-buffered in LDS and filled by LDS-DMA (global_load_lds, 16 B per lane, the K image's
//     chunk swizzle applied on the SOURCE side, pad chunks -- zeros for K, the ones column for V -- written once per kernel and
//     never touched by the DMA): chunk c+1 is requested while chunk c computes; no staging registers, one barrier per chunk
//     (before: load -> barrier -> compute -> barrier with every wave waiting for the loads, 222 us of the 968);
//   * the keys of a chunk are walked in units of 32 (two key tiles = one P.V MFMA k-step) with an online softmax per unit, software-
//     pipelined inside the wave: QK^T of unit u+1 is issued before the softmax of unit u (before: QK^T of 288 keys, then 72 exps per
//     lane, then P.V);
//   * the exponent reference lags the running maximum (see RESCALE_LOG2), so the 48 accumulator registers are rescaled a few times
//     per pass instead of in nearly every unit.
//   * Tq = 256 k + 1 (the CLS token on top of a 16 n x 16 n patch grid: T = 1025 at 448^2, the reference's image size) leaves ONE query
//     for a block of its own.  As a block (rounds 2-5) it cost 0.76 of a full one -- 16 % of the kernel at T = 1025: wave 0 walked every
//     key while seven waves issued requests; with its keys split over the eight waves it still cost 0.56, because a block with nothing to
//     compute streams the pair's K and V at the latency of one 78-KB chunk in flight (profiles/r06/attention_stream_lone_query.txt).
//     Now the pair's LAST FULL block carries it: at the end of every chunk, with the chunk still in LDS, wave (unit % 8) runs the lone
//     query against ONE 32-key unit -- a 16-column tile whose columns all hold that query -- as an online softmax of its own whose
//     state (O^T column, maximum, denominator: DPAD + 4 floats per wave) lives in LDS between chunks (no registers to spare: 238 of
//     256); the eight partial softmaxes are merged after the block's last chunk.  No extra K / V traffic, four work items per pair,
//     and the block that carries the query rotates over a workgroup's items (32 workgroups per XCD would otherwise pin it).
// Work order: the query blocks of one (sample, head) re-read the same K and V (360 KB at T = 1025).  Workgroups are dealt to the 8
// XCDs round-robin by the hardware, so XCD x owns the pairs x, x+8, ... and its workgroups walk pair-major through their query blocks:
// the workgroups of one XCD sit on a few pairs at a time and K / V come out of that XCD's L2 (a plain item = blockIdx walk fetched a
// copy per XCD: HBM-bound at ~3.5 TB/s).
template <int HD, int DPAD, int TKP>
__global__ __launch_bounds__(512) void attention_stream_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int UPC = TKP / 32;                                           // TKP keys per chunk = UPC units of 32 keys
    constexpr int KROW = AttnLayout<DPAD>::KROW, VSTR = AttnLayout<DPAD>::VSTR;
    constexpr int KC = KROW / 8, VC = VSTR / 8, DC = HD / 8;                // 16-byte slots per K row / V row, data chunks per row
    constexpr int NDS = DPAD / 32, NDT = DPAD / 16;
    constexpr int KSLOTS = TKP * KC, VSLOTS = TKP * VC;
    constexpr int NREQ = (KSLOTS + VSLOTS) / 64;                            // LDS-DMA instructions per chunk (workgroup total)
    static_assert(KSLOTS % 64 == 0 && VSLOTS % 64 == 0, "chunk images are whole 1-KiB requests");
    constexpr int STAGE = TKP * (KROW + VSTR);                              // halfs per buffer: K image then V image
    half_t* const smem = reinterpret_cast<half_t*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r15 = lane & 15, g = lane >> 4;
    // query blocks per (sample, head); a lone last query rides on the last full block (see above)
    const bool lone = CGPT_ATT_LONE_CARRIED && (p.Tq & 255) == 1 && p.Tq > 256;
    const int nqb = lone ? p.Tq / 256 : (p.Tq + 255) / 256;
    constexpr int RED_STRIDE = DPAD + 4;                                    // floats per wave: O^T column (16-byte rows), maximum, denominator
    constexpr int RED_OFFSET = 2 * TKP * (KROW + VSTR) * (int)sizeof(half_t);   // bytes: behind the two chunk buffers
    constexpr int QL_OFFSET = RED_OFFSET + 8 * RED_STRIDE * (int)sizeof(float);  // the lone query's B fragment, [chunk d / 8][8 halfs]
    const int nchunks = (p.Tk + TKP - 1) / TKP;
    // balanced chunks: CK keys each (a multiple of 32, <= TKP), so that every chunk's compute covers the next chunk's load
    const int CK = ((p.Tk + 31) / 32 + nchunks - 1) / nchunks * 32;
    const float sl2 = p.scale * 1.44269504088896340736f;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // FOLD (a pad column exists, DPAD > HD): the scores leave the MFMA ready for exp2 -- Q is scaled by scale * log2 e when it is loaded
    // (fp32 multiply, one rounding to fp16; the reference scales q in fp16 too, eva_vit.py:140), and the exponent reference rides in the
    // pad column: K's image holds 1 at d = HD, the Q fragment -reference.  That takes the 16 v_fma per unit (s * scale - reference) out
    // of the softmax, a tenth of the unit's vector-issue cycles.
    constexpr bool FOLD = DPAD > HD;
    constexpr float RESCALE_LOG2 = 8.0f;

    // Both buffers zero, once: rows past the end of the keys are never requested (see request_part), so a buffer row holds either
    // what an earlier chunk left there or these zeros -- finite either way, and the keys' P is 0.  Then the
    // pad slots of both buffers, once: K chunks >= DC are zero (q is zero there too, but 0 x stale-LDS NaN would not be), V chunk DC
    // holds the ones column (DPAD > HD), the rest of a V row's tail is zero
    for (int sidx = tid; sidx < 2 * STAGE / 8; sidx += 512) *reinterpret_cast<f16x8*>(smem + sidx * 8) = zero8;
    __syncthreads();
    for (int sidx = tid; sidx < 2 * TKP * (KC - DC + VC - DC); sidx += 512) {
        const int buf = sidx / (TKP * (KC - DC + VC - DC)), rem = sidx % (TKP * (KC - DC + VC - DC));
        const int row = rem / (KC - DC + VC - DC), k = rem % (KC - DC + VC - DC);
        half_t* base = smem + buf * STAGE;
        if (k < KC - DC) {
            f16x8 v = zero8;
            if (FOLD && k == 0) v[0] = (half_t)1.0f;                       // the column that adds the Q fragment's -reference
            *reinterpret_cast<f16x8*>(base + row * KROW + k_chunk_pos<DPAD>(row, DC + k) * 8) = v;
        } else {
            const int c = DC + (k - (KC - DC));
            f16x8 v = zero8;
            if (DPAD > HD && c == DC) v[0] = (half_t)1.0f;
            *reinterpret_cast<f16x8*>(base + TKP * KROW + row * VSTR + c * 8) = v;
        }
    }

    // XCD x owns the (sample, head) pairs x, x+8, ... and walks them pair-major (see above); with 64 samples or more it owns whole
    // SAMPLES x, x+8, ... instead (pair q of its walk = sample 8 (q / heads) + x, head q % heads), so that the 128-byte lines which
    // neighbouring heads of a sample share at their borders are fetched by one L2 (see attention_kernel's walk)
    const int npairs = p.heads * p.B;
    const bool xcd_map = (gridDim.x & 7) == 0;
    const bool by_sample = xcd_map && p.B >= 64;
    const int xcd = xcd_map ? (int)(blockIdx.x & 7) : 0, nx = xcd_map ? 8 : 1;
    const int lid = xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, nl = xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int my_pairs = by_sample ? ((p.B - xcd + 7) / 8) * p.heads : (npairs - xcd + nx - 1) / nx;
    const int nwork = my_pairs * nqb;
    auto pair_of = [&](int q) { return by_sample ? ((q / p.heads) * 8 + xcd) * p.heads + q % p.heads : xcd + nx * q; };   // = sample * heads + head
    // work item w of this XCD -> query block of its pair w / nqb.  With a carried lone query the blocks of a pair are rotated by a number
    // that grows with the pair, so that a workgroup (items lid, lid + nl, ...) meets the carrying block every nqb-th item and not always
    auto qb_of = [&](int wi) { const int j = wi % nqb; return lone ? (j + (wi - j) / nl) % nqb : j; };

    // request chunk c of work item w into buffer `buf`: request r = wave + 8 i of the chunk covers slots 64 r .. 64 r + 63 of
    // the buffer image (K slots first); a lane fetches the 16-byte chunk that belongs in ITS slot, pad slots are skipped.
    // The slot -> (row, source chunk) decode does not depend on the chunk: one packed register per request, made once.
    // What a request costs is instructions, not bytes (round 6: requests for the block's first chunk only = -13 % of the kernel; a
    // request was ~25 instructions -- row clamp, 64-bit address arithmetic, lane masks reloaded from spilled SGPRs).  Now a lane's
    // byte offset inside the pair's K (or V) rows is made once per kernel, 0xffffffff where the lane has nothing to fetch (pad slot,
    // row beyond a balanced chunk, request beyond the image); per chunk there are two scalar bases and ONE scalar limit -- the bytes
    // of the chunk's rows that exist -- and a request is compare, mask, global_load_lds (scalar base + 32-bit lane offset), unmask.
    constexpr int NI = (NREQ + 7) / 8, KREQ = KSLOTS / 64;
    unsigned voff[NI];
    unsigned from_v = 0;                                                    // bit i: request i of this wave fetches V (wave-uniform)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = wave + 8 * i, slot = r * 64 + lane;
        int row, ch;
        if (r < KREQ) { row = slot / KC; ch = k_chunk_pos<DPAD>(row, slot - row * KC); }   // the swizzle is an involution
        else { const int sv = slot - KSLOTS; row = sv / VC; ch = sv - row * VC; from_v |= 1u << i; }
        voff[i] = (r < NREQ && ch < DC && row < CK) ? (unsigned)(row * (int)p.ldk + ch * 8) * 2u : 0xffffffffu;
    }
    from_v = __builtin_amdgcn_readfirstlane(from_v);
    // The requests of a chunk are issued inside the first two units of the previous chunk (request i in unit i % 2): issued in one
    // burst after the barrier, the workgroup's 78 KiB went through the CU's one address unit (64 B / clock) with all eight waves
    // waiting for their turn -- 14 % of the kernel in the phase stamps; issued any later they have less time to land.
    const char *nKb = nullptr, *nVb = nullptr;                              // the chunk being requested: first byte of its K rows / V rows
    unsigned nlimit = 0;                                                    // ... and the bytes of its rows that exist (0: nothing to request)
    auto set_next = [&](int w, int c) {
        const int pair = pair_of(w / nqb);
        const int h = pair % p.heads, b = pair / p.heads;
        const int64_t first = (int64_t)b * p.kv_batch_stride + h * HD + (int64_t)c * CK * p.ldk;    // (elements; 32-bit offsets inside a sample: launch check)
        nKb = reinterpret_cast<const char*>(p.K + first);
        nVb = reinterpret_cast<const char*>(p.V + first);
        const int rows = min(CK, p.Tk - c * CK);                           // rows past the end of the keys are not fetched
        nlimit = w < nwork ? (unsigned)(rows * (int)p.ldk) * 2u : 0u;      // (past the last work item: nothing)
    };
    auto request_part = [&](int i, int buf) {                               // request i of this wave for the chunk of set_next into buffer `buf`
        if (voff[i] < nlimit) {                                             // (a lane's row exists <=> its offset is below the limit: ch * 16 < ldk * 2)
            const char* src = (((from_v >> i) & 1) ? nVb : nKb) + voff[i];
            char* dst = smem_raw + buf * (STAGE * 2) + (wave + 8 * i) * 1024;   // 64 lanes x 16 bytes per request
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto request_unit = [&](int u, int buf) {                               // the requests that belong to unit u of the current chunk
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i % 2 == u) request_part(i, buf);                           // all of them in the chunk's first two units
    };

#ifdef CGPT_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memtime();
    unsigned long long tlast = stamp_begin;
#define CGPT_S2STAMP(k) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[k] += tn - tlast; tlast = tn; }
#else
#define CGPT_S2STAMP(k)
#endif
    // Q^T B-fragments of work item w's two tiles: lane holds Q[query r15][d = 32 ds + 8 g .. +7] (zero beyond HD).  They are requested
    // behind the previous item's epilogue, all six loads together, then scaled and padded: one memory latency per item.  (Rounds 2-5
    // asked for them inside the item's last unit as load-and-scale in one step, which hipcc emits as load, wait for EVERYTHING in
    // flight, convert -- six memory latencies in a row in the middle of that unit.  Loading straight into qf there and scaling at the
    // top of the next item makes every later use of qf in the unrolled unit loop wait for vmcnt(0), i.e. for the chunk requests; loads
    // in front of the epilogue and the scaling behind it spill 26 registers.)
    f16x8 qf[2][NDS];
    auto load_q = [&](f16x8 (&qn)[2][NDS], int wi) {
        const int pair = pair_of(wi / nqb), qb = qb_of(wi);
        const half_t* Qb = p.Q + (int64_t)(pair / p.heads) * p.q_batch_stride + (pair % p.heads) * HD;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int qrow = min(qb * 256 + wave * 32 + t * 16 + r15, p.Tq - 1);
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                const int d = min(ds * 32 + g * 8, HD - 8);
                qn[t][ds] = *reinterpret_cast<const f16x8*>(Qb + (int64_t)qrow * p.ldq + d);
            }
        }
    };
    auto finish_q = [&](const f16x8 (&qn)[2][NDS]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                f16x8 v = qn[t][ds];
                if constexpr (FOLD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * sl2);
                }
                qf[t][ds] = (ds * 32 + g * 8 < HD) ? v : zero8;       // (FOLD: element 0 of the d = HD chunk is the -reference, 0 at first)
            }
    };
    int w = lid;
    set_next(w, 0);
#pragma unroll
    for (int i = 0; i < NI; ++i) request_part(i, 0);
    int buf = 0;
    for (; w < nwork; w += nl) {
        const int pair = pair_of(w / nqb), qb = qb_of(w);
        const int h = pair % p.heads, b = pair / p.heads;
        half_t* Ob = p.O + (int64_t)b * p.o_batch_stride + h * HD;
        const int q0 = qb * 256 + wave * 32;                                // this wave's queries q0 .. q0 + 31 (tiles a, b)
        const bool have = q0 < p.Tq;
        const bool carrier = lone && qb == nqb - 1;                         // (workgroup-uniform) this block also computes query Tq - 1
#ifdef CGPT_STAMPS
        const unsigned long long item_begin = __builtin_amdgcn_s_memtime();
#endif

        if (w == lid) { f16x8 qn[2][NDS]; load_q(qn, w); finish_q(qn); }    // later items: at the end of the previous item
        f32x4 o[2][NDT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run[2] = {FOLD ? 0.f : -1e30f, FOLD ? 0.f : -1e30f}, l_run[2] = {0.f, 0.f};   // exponent reference (FOLD: in log2 units, fp16-exact)
        CGPT_S2STAMP(0)                                  // item set-up (Q loads issued)

        for (int c = 0; c < nchunks; ++c, buf ^= 1) {
            // this wave's requests for chunk c have landed; after the barrier everybody's have, and every wave is done with the
            // other buffer (chunk c-1), which the next chunk's requests may now overwrite
            if (carrier && c == 0 && wave == 0) {
                // the lone query's Q^T fragment for every wave, scaled like the others, its reference slot 0: the load rides on the wait
                // for the chunk's requests, the words are visible after the chunk barrier (the previous block's readers are past theirs)
                const half_t* qrow = p.Q + (int64_t)b * p.q_batch_stride + h * HD + (int64_t)(p.Tq - 1) * p.ldq;
                f16x8 v = zero8;
                if (lane < DC) v = *reinterpret_cast<const f16x8*>(qrow + lane * 8);
                if constexpr (FOLD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * sl2);
                }
                if (lane < DPAD / 8) *reinterpret_cast<f16x8*>(smem_raw + QL_OFFSET + lane * 16) = v;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CGPT_S2STAMP(1)                              // waiting for this wave's requests of the chunk (and the item's Q)
            if (!(CGPT_ATT_ABLATE & 128) || c == 0) __syncthreads();
            CGPT_S2STAMP(2)                              // barrier
            if (carrier && c == 0) {
                // this wave's partial softmax of the lone query: O^T column 0, maximum -inf, denominator 0.  (Its own LDS words, read and
                // written in program order; behind the barrier because wave 0 may have been merging the previous block's states till then.)
                float* st = reinterpret_cast<float*>(smem_raw + RED_OFFSET) + wave * RED_STRIDE;
                for (int i = lane; i < RED_STRIDE; i += 64) st[i] = i == DPAD ? -1e30f : 0.f;
            }
            if (c + 1 < nchunks) set_next(w, c + 1);
            else set_next(w + nl, 0);                      // (past the last work item: request_part does nothing)
            if (!have) {
#pragma unroll
                for (int i = 0; i < NI; ++i) request_part(i, buf ^ 1);
                continue;
            }
            CGPT_S2STAMP(3)
            const half_t* Ks = smem + buf * STAGE;
            const half_t* Vs = Ks + TKP * KROW;
            const int key0 = c * CK;
            const int nu = min(CK / 32, (p.Tk - key0 + 31) / 32);         // units of this chunk that hold keys

            // K fragments of unit u: lane holds row 32u + 16kt + r15, chunk 4ds + g (swizzled)
            auto read_k = [&](f16x8 (&kf)[2][NDS], int u) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const int row = u * 32 + kt * 16 + r15;
#pragma unroll
                    for (int ds = 0; ds < NDS; ++ds)
                        kf[kt][ds] = *reinterpret_cast<const f16x8*>(Ks + row * KROW + k_chunk_pos<DPAD>(row, ds * 4 + g) * 8);
                }
            };
            auto qk = [&](f32x4 (&s)[2][2], const f16x8 (&kf)[2][NDS]) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) s[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ds = 0; ds < NDS; ++ds)                           // four independent chains, one k-step each per round
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
                            s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][ds], qf[t][ds], s[t][kt], 0, 0, 0);
            };

            // the chunk's units, fully unrolled (LDS offsets become immediates, the score registers ping-pong by unit parity)
            f16x8 kf[2][NDS];
            f32x4 sc[2][2][2];                                             // [unit parity][tile][key tile]
            read_k(kf, 0);
            qk(sc[0], kf);
            read_k(kf, 1);
            const unsigned vlane = lds_address(Vs + (4 * g + (r15 >> 2)) * VSTR + 4 * (r15 & 3));   // this lane's corner of a unit's V rows
            static_for_while<UPC>([&](auto uc) __attribute__((always_inline)) -> bool {
                constexpr int u = decltype(uc)::value;
                if (u >= nu) return false;
                f32x4 (&s_cur)[2][2] = sc[u & 1];
                // V fragments of unit u (two transposed 4x16 reads per d-tile)
                f16x4 vr[NDT][2];
                {
                    // (one address register per chunk; unit, key half and d-tile are instruction offsets)
                    if (!(CGPT_ATT_ABLATE & 2) || u == 0)
                    static_for<NDT>([&](auto dtc) __attribute__((always_inline)) {
                        constexpr int dt = decltype(dtc)::value;
                        vr[dt][0] = lds_read_tr16_untracked<(u * 32 * VSTR + dt * 16) * 2>(vlane);
                        vr[dt][1] = lds_read_tr16_untracked<(u * 32 * VSTR + 16 * VSTR + dt * 16) * 2>(vlane);
                    });
                }
                if (!(CGPT_ATT_ABLATE & 32) || c == 0) request_unit(u, buf ^ 1);
                // QK^T of the NEXT unit goes to the matrix pipe first ...
                // (unconditionally inside the chunk: a conditional fragment read makes the compiler split the fp16 vectors into
                // halves and re-pack them with v_perm in front of every MFMA; units past the end read valid LDS and are never used)
                // (the products too: skipping them for a unit past the end leaves a join at which hipcc, not knowing what is in flight on
                // the other path, waits for ALL LDS reads -- the uncounted V reads included -- after the first K fragment read; a short
                // last chunk pays 12 MFMAs on valid, unused LDS contents once per block)
                if (u + 1 < UPC) {
                    if (!(CGPT_ATT_ABLATE & 16) || u == 0) qk(sc[(u + 1) & 1], kf);
                    if (u + 2 < UPC && !(CGPT_ATT_ABLATE & 4)) read_k(kf, u + 2);
                }
                // ... and this unit's softmax runs under it
                const int kb = key0 + u * 32;
                if (kb + 32 > p.Tk) {                                      // wave-uniform: the unit holds keys past the end
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (kb + kt * 16 + 4 * g + r >= p.Tk) s_cur[t][kt][r] = -1e30f;
                }
                // The exponent reference is allowed to lag behind the running maximum by up to 2^RESCALE_LOG2 (P then reaches 256,
                // far inside fp16; numerator and denominator use the same reference, so the quotient is unchanged): with 32 queries
                // in a wave SOME query's maximum moves in almost every unit, and rescaling 48 accumulator registers each time was a
                // quarter of the unit's VALU work.  Now it happens when a query's scores really outgrow the reference -- normally in
                // a block's first unit only.  The reference never exceeds the running maximum, so the largest P is >= 1.
                // The LANE's maximum over its eight keys of the unit decides whether anything has to move ("some query's maximum exceeds
                // the reference by 2^8" and "some lane's does" are the same statement); the maximum over a query's four lane groups --
                // two cross-lane swaps and their shuffling per tile, a third of the softmax's plain VALU work in rounds 2-5 -- is only
                // made where the reference really moves.
                float mx[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float m = fmaxf(fmaxf(s_cur[t][0][0], s_cur[t][0][1]), s_cur[t][0][2]);                  // v_max3 x 3 + v_max
                    const float m2 = fmaxf(fmaxf(s_cur[t][0][3], s_cur[t][1][0]), s_cur[t][1][1]);
                    m = fmaxf(fmaxf(s_cur[t][1][2], s_cur[t][1][3]), m);
                    mx[t] = fmaxf(m, m2);
                }
                f16x8 pf[2];
                if constexpr (FOLD) {
                    // scores are (q . k) scale log2 e - reference already
                    const bool first = c == 0 && u == 0;                   // wave-uniform: no reference yet (it is 0, o is 0)
                    if (first || (!(CGPT_ATT_ABLATE & 64) && __builtin_amdgcn_ballot_w64(mx[0] > RESCALE_LOG2 || mx[1] > RESCALE_LOG2) != 0)) {
                        const bool nxt = u + 1 < UPC && u + 1 < nu;        // the next unit's scores exist and carry the old reference
                        mx[0] = max_over_lane_groups(mx[0]);
                        mx[1] = max_over_lane_groups(mx[1]);
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const float ref_new = (float)(half_t)(m_run[t] + (first ? mx[t] : fmaxf(mx[t], 0.f)));   // what the fragment can hold
                            const float d = ref_new - m_run[t];
                            m_run[t] = ref_new;
                            // (not in the item's last unit: no QK^T follows)
                            if (g == 3 && !(c + 1 == nchunks && u + 1 == nu)) qf[t][NDS - 1][0] = (half_t)(-ref_new);
                            if (!first) {
                                const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                                for (int dt = 0; dt < NDT; ++dt) o[t][dt] *= alpha;
                            }
#pragma unroll
                            for (int kt = 0; kt < 2; ++kt) {
                                s_cur[t][kt] -= d;
                                if (nxt) sc[(u + 1) & 1][t][kt] -= d;
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float e[8];
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) e[kt * 4 + r] = (CGPT_ATT_ABLATE & 1) ? s_cur[t][kt][r] : __builtin_amdgcn_exp2f(s_cur[t][kt][r]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[t][j] = (half_t)e[j];
                    }
                } else {
                    bool need = false;
#pragma unroll
                    for (int t = 0; t < 2; ++t) need = need || ((mx[t] - m_run[t]) * sl2 > RESCALE_LOG2);   // (lane maxima: see above)
                    if (__builtin_amdgcn_ballot_w64(need) != 0) {          // wave-uniform
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const float m_new = fmaxf(m_run[t], max_over_lane_groups(mx[t]));
                            const float alpha = __builtin_amdgcn_exp2f((m_run[t] - m_new) * sl2);   // first unit: 0 (o is 0)
                            l_run[t] *= alpha;
#pragma unroll
                            for (int dt = 0; dt < NDT; ++dt) o[t][dt] *= alpha;
                            m_run[t] = m_new;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float mxs = m_run[t] * sl2;
                        float e[8];
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) e[kt * 4 + r] = __builtin_amdgcn_exp2f(fmaf(s_cur[t][kt][r], sl2, -mxs));
                        l_run[t] += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[t][j] = (half_t)e[j];
                    }
                }
                // O^T += V^T P^T for both tiles (the V reads are the untracked kind: one wait, tied to their registers)
                if constexpr (NDT == 6)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[0][0]), "+v"(vr[0][1]), "+v"(vr[1][0]), "+v"(vr[1][1]), "+v"(vr[2][0]), "+v"(vr[2][1]),
                                 "+v"(vr[3][0]), "+v"(vr[3][1]), "+v"(vr[4][0]), "+v"(vr[4][1]), "+v"(vr[5][0]), "+v"(vr[5][1]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[0][0]), "+v"(vr[0][1]), "+v"(vr[1][0]), "+v"(vr[1][1]), "+v"(vr[2][0]), "+v"(vr[2][1]),
                                 "+v"(vr[3][0]), "+v"(vr[3][1]));
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const f16x8 vf = __builtin_shufflevector(vr[dt][0], vr[dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        if (!(CGPT_ATT_ABLATE & 8) || (dt == 0 && t == 0)) o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[t], o[t][dt], 0, 0, 0);
                }
                return true;
            });
#pragma unroll
            for (int u = 1; u < UPC; ++u)
                if (u >= nu) request_unit(u, buf ^ 1);    // a short last chunk: the shares of the units it does not have
            if (carrier) {
                // The lone query against this wave's unit of the chunk, if it has one (unit u of the pass belongs to wave u % 8; a chunk has
                // fewer than eight).  All 16 columns of the tile hold the same query, so every lane group reads the same state words.
                const int ul = (wave - c * (CK / 32)) & 7;
                if (ul < nu) {
                    float* st = reinterpret_cast<float*>(smem_raw + RED_OFFSET) + wave * RED_STRIDE;
                    f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ds = 0; ds < NDS; ++ds) {
                        const f16x8 qfr = *reinterpret_cast<const f16x8*>(smem_raw + QL_OFFSET + (ds * 4 + g) * 16);
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) {
                            const int row = ul * 32 + kt * 16 + r15;
                            const f16x8 kfr = *reinterpret_cast<const f16x8*>(Ks + row * KROW + k_chunk_pos<DPAD>(row, ds * 4 + g) * 8);
                            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr, qfr, s[kt], 0, 0, 0);
                        }
                    }
                    const int kb = key0 + ul * 32;
                    float mx = -1e30f;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = FOLD ? s[kt][r] : s[kt][r] * sl2;                     // FOLD: Q was scaled when it was loaded
                            if (kb + kt * 16 + 4 * g + r >= p.Tk) v = -1e30f;
                            s[kt][r] = v;
                            mx = fmaxf(mx, v);
                        }
                    mx = max_over_lane_groups(mx);
                    const float m_old = st[DPAD];
                    const float m_new = fmaxf(m_old, mx);
                    const float alpha = __builtin_amdgcn_exp2f(m_old - m_new);              // first unit: 0 (the state is 0)
                    f16x8 pfr;
                    float sum = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = __builtin_amdgcn_exp2f(s[kt][r] - m_new);
                            sum += e;
                            pfr[kt * 4 + r] = (half_t)e;
                        }
                    const unsigned vaddr = lds_address(Vs + (ul * 32 + 4 * g + (r15 >> 2)) * VSTR + 4 * (r15 & 3));
                    f32x4 ol[NDT];
                    f16x4 vl[NDT][2];
                    static_for<NDT>([&](auto dtc) __attribute__((always_inline)) {
                        constexpr int dt = decltype(dtc)::value;
                        vl[dt][0] = lds_read_tr16_untracked<dt * 16 * 2>(vaddr);
                        vl[dt][1] = lds_read_tr16_untracked<(16 * VSTR + dt * 16) * 2>(vaddr);
                        ol[dt] = *reinterpret_cast<const f32x4*>(st + dt * 16 + 4 * g) * alpha;
                    });
                    if constexpr (NDT == 6)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vl[0][0]), "+v"(vl[0][1]), "+v"(vl[1][0]), "+v"(vl[1][1]), "+v"(vl[2][0]), "+v"(vl[2][1]),
                                     "+v"(vl[3][0]), "+v"(vl[3][1]), "+v"(vl[4][0]), "+v"(vl[4][1]), "+v"(vl[5][0]), "+v"(vl[5][1]));
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vl[0][0]), "+v"(vl[0][1]), "+v"(vl[1][0]), "+v"(vl[1][1]), "+v"(vl[2][0]), "+v"(vl[2][1]),
                                     "+v"(vl[3][0]), "+v"(vl[3][1]));
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        const f16x8 vf = __builtin_shufflevector(vl[dt][0], vl[dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
                        ol[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pfr, ol[dt], 0, 0, 0);
                    }
                    if (r15 == 0) {
#pragma unroll
                        for (int dt = 0; dt < NDT; ++dt) *reinterpret_cast<f32x4*>(st + dt * 16 + 4 * g) = ol[dt];
                        if (g == 0) st[DPAD] = m_new;
                    }
                    if constexpr (!(DPAD > HD)) {                                           // no ones column in V: the row sum by VALU
                        sum += __shfl_xor(sum, 16);
                        sum += __shfl_xor(sum, 32);
                        const float l_old = st[DPAD + 1];
                        if (lane == 0) st[DPAD + 1] = l_old * alpha + sum;
                    }
                }
            }
            CGPT_S2STAMP(4)                              // the chunk's units
        }
        if (carrier) {
            // the eight partial softmaxes meet: wave 0 merges them and stores row Tq - 1.  (States and Q fragment are rewritten behind /
            // in front of the next carrying block's first chunk barrier, which wave 0 reaches after this merge.)
            __syncthreads();
            if (wave == 0) {
                const float* red = reinterpret_cast<const float*>(smem_raw + RED_OFFSET);
                float mall = -1e30f;
#pragma unroll
                for (int k = 0; k < 8; ++k) mall = fmaxf(mall, red[k * RED_STRIDE + DPAD]);
                float acc[2] = {0.f, 0.f}, dsum = 0.f;
#pragma nounroll
                for (int k = 0; k < 8; ++k) {
                    const float wk = __builtin_amdgcn_exp2f(red[k * RED_STRIDE + DPAD] - mall);   // a wave without units: 2^-inf = 0
                    acc[0] += wk * red[k * RED_STRIDE + lane];
                    if (lane + 64 < DPAD) acc[1] += wk * red[k * RED_STRIDE + 64 + lane];
                    dsum += wk * red[k * RED_STRIDE + ((DPAD > HD) ? HD : DPAD + 1)];        // the ones column of V, or the VALU row sums
                }
                const float inv = 1.0f / dsum;
                half_t* orow = Ob + (int64_t)(p.Tq - 1) * p.ldo;
                if (lane < HD) orow[lane] = (half_t)(acc[0] * inv);
                if (lane + 64 < HD) orow[lane + 64] = (half_t)(acc[1] * inv);
            }
        }
        if (have) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float den;
                if constexpr (DPAD > HD) {
                    constexpr int DT = HD / 16, GG = (HD % 16) / 4, RR = (HD % 16) % 4;
                    den = __shfl(o[t][DT][RR], 16 * GG + r15);
                } else {
                    den = l_run[t];
                    den += __shfl_xor(den, 16);
                    den += __shfl_xor(den, 32);
                }
                const float inv = 1.0f / den;
                const int q = q0 + t * 16 + r15;
#pragma unroll
                for (int dp = 0; dp < NDT / 2; ++dp) {
                    const int da = 2 * dp, db = 2 * dp + 1;
                    const f16x4 ha = {(half_t)(o[t][da][0] * inv), (half_t)(o[t][da][1] * inv), (half_t)(o[t][da][2] * inv), (half_t)(o[t][da][3] * inv)};
                    const f16x4 hb = {(half_t)(o[t][db][0] * inv), (half_t)(o[t][db][1] * inv), (half_t)(o[t][db][2] * inv), (half_t)(o[t][db][3] * inv)};
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x2 ua = __builtin_bit_cast(u32x2, ha), ub = __builtin_bit_cast(u32x2, hb);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                    const u32x4 packed = {s0[0], s1[0], s0[1], s1[1]};
                    const int d0 = ((g & 1) ? db : da) * 16 + (g >> 1) * 8;
                    if (q < p.Tq && d0 < HD) *reinterpret_cast<u32x4*>(Ob + (int64_t)q * p.ldo + d0) = packed;
                }
            }
        }
        if (w + nl < nwork) { f16x8 qn[2][NDS]; load_q(qn, w + nl); CGPT_FENCE finish_q(qn); }   // (fence: all six loads, then one wait)
#ifdef CGPT_STAMPS
        if ((p.Tq & 255) == 1 && p.Tq > 256 && qb == (p.Tq - 1) / 256 - (lone ? 1 : 0))           // time inside the lone query's block (as a
            ph[5] += __builtin_amdgcn_s_memtime() - item_begin;                                    // block of its own | the block that carries it)
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CGPT_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        d[0] = __builtin_amdgcn_s_memtime() - stamp_begin;
        for (int k = 0; k < 6; ++k) d[1 + k] = ph[k];
    }
#endif
}


// per-device launch state (one handle per device, possibly several devices per process)
constexpr int kMaxDevicesA = 64;
inline hipError_t device_cus(int& dev, int& cus) {
    static int table[kMaxDevicesA] = {0};
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevicesA) return hipErrorInvalidDevice;
    if (table[dev] == 0) {
        int n = 0;
        if (hipError_t e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
        table[dev] = n > 0 ? n : 256;
    }
    cus = table[dev];
    return hipSuccess;
}

template <int HD, int DPAD, int TKP>
hipError_t launch_stream(const AttnParams& p, hipStream_t stream) {
    // two chunk buffers + a carried lone query's eight partial softmaxes (DPAD + 4 floats each) and its Q fragment (DPAD halfs)
    constexpr int lds_bytes = 2 * TKP * (AttnLayout<DPAD>::KROW + AttnLayout<DPAD>::VSTR) * (int)sizeof(half_t) + 8 * (DPAD + 4) * (int)sizeof(float)
                              + DPAD * (int)sizeof(half_t);
    static_assert(lds_bytes <= 160 * 1024, "LDS of a gfx950 CU");
    int dev = 0, num_cus = 0;
    if (hipError_t e = device_cus(dev, num_cus); e != hipSuccess) return e;
    static bool configured[kMaxDevicesA] = {false};
    if (!configured[dev]) {
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_stream_kernel<HD, DPAD, TKP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            e != hipSuccess) return e;
        configured[dev] = true;
    }
    const bool lone = CGPT_ATT_LONE_CARRIED && (p.Tq & 255) == 1 && p.Tq > 256;          // (the kernel's own rule)
    const int items = p.heads * p.B * (lone ? p.Tq / 256 : (p.Tq + 255) / 256);
    int grid = items < num_cus ? items : num_cus;
    if (grid >= 8) grid &= ~7;                      // a multiple of 8 enables the XCD-aware work order
    hipLaunchKernelGGL((attention_stream_kernel<HD, DPAD, TKP>), dim3(grid), dim3(512), lds_bytes, stream, p);
    return hipGetLastError();
}

#undef CGPT_FENCE

template <int HD, int DPAD, int NKT, int NT>
hipError_t launch_one(const AttnParams& p, hipStream_t stream) {
    constexpr int lds_bytes = NKT * 16 * (AttnLayout<DPAD>::KROW + AttnLayout<DPAD>::VSTR) * (int)sizeof(half_t);
    int dev = 0, num_cus = 0;
    if (hipError_t e = device_cus(dev, num_cus); e != hipSuccess) return e;
    static bool configured[kMaxDevicesA] = {false};
    if (!configured[dev]) {
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<HD, DPAD, NKT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            e != hipSuccess) return e;
        configured[dev] = true;
    }
    const int items = p.heads * p.B;
    const int per_cu = lds_bytes > 80 * 1024 ? 1 : (lds_bytes > 40 * 1024 ? 2 : 4);     // resident workgroups per CU (LDS)
    const int grid = items < num_cus * per_cu ? items : num_cus * per_cu;
    hipLaunchKernelGGL((attention_kernel<HD, DPAD, NKT, NT>), dim3(grid), dim3(NT), lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_attention(const AttnParams& p_in, hipStream_t stream) {
    AttnParams p = p_in;
    p.dbg = g_gemm_dbg;                             // diagnostic stamp buffer (null outside -DCGPT_STAMPS experiments)
    if (p.B <= 0 || p.heads <= 0 || p.Tq <= 0 || p.Tk <= 0) return hipErrorInvalidValue;
    if ((p.ldq % 8) || (p.ldk % 8) || (p.ldv % 8) || (p.ldo % 8)) return hipErrorInvalidValue;   // 16-byte row alignment
    if (p.ldk != p.ldv || p.Tk * p.ldk >= (1ll << 30) || p.Tq * p.ldq >= (1ll << 30)) return hipErrorInvalidValue;   // 32-bit offsets within a sample
    const bool small = p.Tk <= 32;
    if (p.Tk > 288) {                               // K/V streamed through LDS in 192-key chunks (448^2 images)
        if (p.head_dim == 88) return launch_stream<88, 96, 192>(p, stream);
        if (p.head_dim == 64) return launch_stream<64, 64, 192>(p, stream);
        return hipErrorInvalidValue;
    }
    if (p.head_dim == 88) return small ? launch_one<88, 96, 2, 128>(p, stream) : launch_one<88, 96, 18, 512>(p, stream);
    if (p.head_dim == 64) return small ? launch_one<64, 64, 2, 128>(p, stream) : launch_one<64, 64, 18, 512>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace cgpt
