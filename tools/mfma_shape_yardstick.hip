// Yardstick for VERDICT r4 item 1: what can a GEMM K loop built from v_mfma_f32_32x32x16_f16 with 128 x 128 wave tiles (one wave per SIMD)
// reach at best on this chip, next to the shipped design (v_mfma_f32_16x16x32_f16, 128 x 64 wave tiles, two waves per SIMD)?
// Every variant is a free-running loop with the FULL operand traffic of a 256 x 256 x 64 workgroup tile and NO synchronisation at all (no
// barrier, nothing waited for beyond what the data dependences of the loop itself need): fragment reads by ds_read_b128 from a 64-KiB LDS image
// of random fp16 data, and the operand stream (64 KiB per K = 64 and CU, 1-KiB pieces from an L2-resident buffer) either by LDS-DMA
// (global_load_lds_dwordx4) or staged through registers (global_load_dwordx4, ds_write_b128 four k-steps later).  Such a loop is an UPPER bound
// for a kernel of that design: the real kernel adds waits, barriers, address arithmetic, tile boundaries and epilogues.
//   SHAPE 16: 16x16x32 MFMAs, k-step 32;  SHAPE 32: 32x32x16 MFMAs, k-step 16.   TM x TN MFMA tiles per wave, W waves per CU.
//   Per k-step and wave: TM + TN fragment reads, TM x TN MFMAs, kstep / W pieces of the stream.
// Each variant runs back to back for ~1.2 s before ~0.4 s are timed; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz.
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/mfma_shape_yardstick.hip -o /tmp/yard && /tmp/yard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE> struct Acc;
template <> struct Acc<16> { typedef f32x4 type; };
template <> struct Acc<32> { typedef f32x16 type; };

// STG: 0 = no operand stream, 1 = LDS-DMA, 2 = global_load_dwordx4 + ds_write_b128.  RD: fragment reads.  IL: interleave reads / stream
// operations between the MFMAs with sched_group_barrier (one memory operation per MFMA) instead of leaving the order to the compiler.
template <int SHAPE, int TM, int TN, int THREADS, int STG, bool RD, bool IL>
__global__ __launch_bounds__(THREADS) void yard_loop(const f16x8* __restrict__ src, float* out, unsigned long long* clk, int iters) {
    constexpr int W = THREADS / 64;
    constexpr int KSTEP = SHAPE == 32 ? 16 : 32;
    constexpr int PIECES = KSTEP / W;
    static_assert(PIECES >= 1, "at least one piece per k-step");
    __shared__ __attribute__((aligned(16))) f16x8 lds[8192];                 // 64 KiB image that is read + 64 KiB that the stream fills
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 8192; i += THREADS) lds[i] = src[(blockIdx.x * 4096 + i) & 0xFFFF];
    __syncthreads();
    typedef typename Acc<SHAPE>::type acc_t;
    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < (SHAPE == 32 ? 16 : 4); ++e) acc[i][j][e] = 0.f;
    f16x8 a[2][TM], b[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[s][i] = src[(blockIdx.x * 977 + tid * 8 + i + s * 31) & 0xFFFF];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[s][j] = src[(blockIdx.x * 1409 + tid * 8 + 4 + j + s * 17) & 0xFFFF];
    }
    f16x8 ring[4][PIECES];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < PIECES; ++q) ring[s][q] = src[(tid + s * 64 + q * 256) & 0xFFFF];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int rd = tid;
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {                                        // four k-steps per trip: static fragment-set and ring indices
            constexpr int dummy = 0; (void)dummy;
            const int cur = s & 1, nxt = cur ^ 1;
            if (RD) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = lds[(rd + i * 64) & 4095];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = lds[(rd + (TM + j) * 64) & 4095];
                rd += (TM + TN) * 64;
            }
            if (STG == 1) {
#pragma unroll
                for (int q = 0; q < PIECES; ++q) {
                    const f16x8* g0 = src + ((blockIdx.x * 64 + (it + s) * 131 + q * 4096 + lane) & 0xFFFF);
                    const int slot = ((it + s) * PIECES + q) & 7;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g0,
                                                     (__attribute__((address_space(3))) void*)(lds + 4096 + (wv & 7) * 512 + slot * 64), 16, 0, 0);
                }
            }
            if (STG == 2) {
#pragma unroll
                for (int q = 0; q < PIECES; ++q) {
                    const int slot = ((it + s) * PIECES + q) & 7;
                    // loaded four k-steps ago; inline asm so that the store into the never-read half is not eliminated
                    const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(lds + 4096 + (wv & 7) * 512 + slot * 64 + lane);
                    asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(ring[s][q]) : "memory");
                    ring[s][q] = src[(blockIdx.x * 64 + (it + s) * 131 + q * 4096 + lane) & 0xFFFF];
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (SHAPE == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                }
            if (IL) {
                // one memory operation behind each of the first MFMAs of the step, the rest of the MFMAs in one group
                constexpr int NMEM_DS = (RD ? TM + TN : 0) + (STG == 2 ? PIECES : 0);
                constexpr int NMEM_VM = (STG != 0 ? PIECES : 0);
#pragma unroll
                for (int k = 0; k < NMEM_DS; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x080, 1, 0); }
#pragma unroll
                for (int k = 0; k < NMEM_VM; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < (SHAPE == 32 ? 16 : 4); ++e) sum += acc[i][j][e];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < PIECES; ++q) sum += (float)ring[s][q][0];
    out[blockIdx.x * THREADS + tid] = sum;
    if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int TM, int TN, int THREADS, int STG, bool RD, bool IL> void run(const char* name, const f16x8* src) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 2 * 8);
    hipMemset(clk, 0, 256 * 2 * 8);
    const double flop_per_mfma = SHAPE == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    const int iters = (SHAPE == 32 ? 16000 : 8000) * 16 / (TM * TN) * (SHAPE == 32 ? 1 : 4) / 4 * 4;   // ~equal FLOP per launch for every variant
    const double flop = 256.0 * (THREADS / 64) * (double)iters * TM * TN * flop_per_mfma;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL((yard_loop<SHAPE, TM, TN, THREADS, STG, RD, IL>), dim3(256), dim3(THREADS), 0, 0, src, out, clk, iters); };
    launch();
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("%-74s LAUNCH FAILED\n", name); return; }
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float one; hipEventElapsedTime(&one, e0, e1);
    const int warm = (int)(1200.0 / one) + 1, timed = (int)(400.0 / one) + 1;
    for (int i = 0; i < warm; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < timed; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    const double ghz = cyc / real * 0.1, tf = flop * timed / (ms * 1e-3) / 1e12;
    // SIMD cycles per 16 384 FLOP (= one 16x16x32 MFMA, half a 32x32x16): 16 = the matrix pipe never idles
    printf("%-74s %7.1f TFLOP/s = %.3f of 2 500   clock %.2f GHz   SIMD cycles per 16 KFLOP %.2f\n", name, tf, tf / 2500.0, ghz,
           ghz * 1e9 * 1024.0 * 16384.0 / (tf * 1e12));
    fflush(stdout);
    hipFree(out); hipFree(clk);
}

int main() {
    const size_t n = 65536 + 16;
    std::vector<_Float16> hr(n * 8);
    srand(7);
    for (auto& v : hr) v = (_Float16)(((rand() & 0xFFFF) / 32768.0f - 1.0f) * 0.5f);
    f16x8* dr;
    hipMalloc(&dr, n * 16);
    hipMemcpy(dr, hr.data(), n * 16, hipMemcpyHostToDevice);
    //   shape TM TN threads stream reads interleave
    run<16, 8, 4, 512, 0, false, false>("16x16x32 bare, 128x64 wave tile, 2 waves/SIMD", dr);
    run<32, 4, 4, 256, 0, false, false>("32x32x16 bare, 128x128 wave tile, 1 wave/SIMD", dr);
    run<32, 4, 2, 512, 0, false, false>("32x32x16 bare, 128x64 wave tile, 2 waves/SIMD", dr);
    run<16, 8, 8, 256, 0, false, false>("16x16x32 bare, 128x128 wave tile, 1 wave/SIMD", dr);
    run<16, 8, 4, 512, 0, true, false>("16x16x32 128x64 x2: + fragment reads (shipped design's reads)", dr);
    run<16, 8, 4, 512, 1, true, false>("16x16x32 128x64 x2: + reads + LDS-DMA stream  [= shipped design]", dr);
    run<32, 4, 4, 256, 0, true, false>("32x32x16 128x128 x1: + fragment reads", dr);
    run<32, 4, 4, 256, 0, true, true>("32x32x16 128x128 x1: + fragment reads, interleaved", dr);
    run<32, 4, 4, 256, 1, true, false>("32x32x16 128x128 x1: + reads + LDS-DMA stream", dr);
    run<32, 4, 4, 256, 1, true, true>("32x32x16 128x128 x1: + reads + LDS-DMA stream, interleaved", dr);
    run<32, 4, 4, 256, 2, true, false>("32x32x16 128x128 x1: + reads + register-staged stream", dr);
    run<32, 4, 4, 256, 2, true, true>("32x32x16 128x128 x1: + reads + register-staged stream, interleaved", dr);
    run<16, 8, 8, 256, 1, true, false>("16x16x32 128x128 x1: + reads + LDS-DMA stream  [= round 1's v6]", dr);
    run<16, 8, 8, 256, 1, true, true>("16x16x32 128x128 x1: + reads + LDS-DMA stream, interleaved", dr);
    run<32, 4, 2, 512, 1, true, false>("32x32x16 128x64 x2: + reads + LDS-DMA stream (shipped tiling, other MFMA)", dr);
    run<16, 8, 4, 512, 1, true, false>("16x16x32 128x64 x2: + reads + LDS-DMA stream  [shipped design, again]", dr);
    return 0;
}
