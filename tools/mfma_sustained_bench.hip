// What dense fp16 MFMA rate does an MI355X SUSTAIN?  The 2.5 PFLOP/s of the data sheet is 256 CUs x 4 SIMDs x 1 024 FLOP per clock at 2.4 GHz;
// under an MFMA-dense load the chip lowers its clock (MI355X_MICROARCH.md, "DVFS give-back"), so the number a GEMM can be compared with
// is the rate of a loop that does NOTHING but v_mfma_f32_16x16x32_f16 -- operands in registers, no LDS, no memory -- on the same kind of data.
// Variants: random / zero operands (data-dependent power), one / two waves per SIMD, and the GEMM's own LDS traffic added (one
// ds_read_b128 per 2.67 MFMAs, as the 128 x 64 wave tile reads 24 fragments per 64 MFMAs).  Every variant runs back to back for ~1.5 s
// before it is timed for ~0.5 s; the in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz.
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/mfma_sustained_bench.hip -o /tmp/mfma && /tmp/mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ORDER (MFMA-only variants): 0 = the A operand stays for four consecutive MFMAs (i outer, j inner: the GEMM's order shares one operand
// between neighbours too), 1 = no operand shared between consecutive MFMAs (diagonal walk), 2 = both operands the same registers throughout.
template <bool LDS, bool DMA = false, int ORDER = 0>
__global__ __launch_bounds__(512) void mfma_loop(const f16x8* __restrict__ src, float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[DMA ? 8192 : 4096];    // 64 KiB of operand-shaped data (+ 64 KiB that the LDS-DMA requests fill)
    const int tid = threadIdx.x;
    if (LDS) { for (int i = tid; i < 4096; i += blockDim.x) lds[i] = src[(blockIdx.x * 4096 + i) & 0xFFFF]; __syncthreads(); }
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(blockIdx.x * 977 + tid * 8 + i) & 0xFFFF]; b[i] = src[(blockIdx.x * 1409 + tid * 8 + 4 + i) & 0xFFFF]; }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int rd = tid;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {                               // static register indices (a runtime index would go to scratch)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ORDER == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                    else if (ORDER == 1) acc[j][(i + j) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[(i + j) & 3], acc[j][(i + j) & 3], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc[i][j], 0, 0, 0);
                }
            if (LDS) {                                                       // 6 fragment reads per 16 MFMAs (24 per 64)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = lds[(rd + i * 64) & 4095];
                b[2 * half] = lds[(rd + 256) & 4095];
                b[2 * half + 1] = lds[(rd + 320) & 4095];
                rd += 384;
            }
            if (DMA) {                                                       // the GEMM's operand stream: 2 LDS-DMA requests of 1 KiB per 16 MFMAs (8 per 64),
                const int wv = tid >> 6, slot = ((it + half) * 2) & 7;      // from an L2-resident buffer into the other half of LDS; never waited for
                const f16x8* g0 = src + ((blockIdx.x * 64 + (it + half) * 131 + (tid & 63)) & 0xFFFF);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g0,
                                                 (__attribute__((address_space(3))) void*)(lds + 4096 + wv * 512 + slot * 64), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g0 + 4096),
                                                 (__attribute__((address_space(3))) void*)(lds + 4096 + wv * 512 + ((slot + 1) & 7) * 64), 16, 0, 0);
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <bool LDS, bool DMA = false, int ORDER = 0> void run(const char* name, int threads, const f16x8* src) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 2 * 8);
    const int iters = 20000;                                                 // 320 000 MFMAs per wave
    const double flop = 256.0 * (threads / 64) * iters * 16.0 * (2.0 * 16 * 16 * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL((mfma_loop<LDS, DMA, ORDER>), dim3(256), dim3(threads), 0, 0, src, out, clk, iters); };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float one; hipEventElapsedTime(&one, e0, e1);
    const int warm = (int)(1500.0 / one) + 1, timed = (int)(500.0 / one) + 1;
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
    // SIMD cycles per MFMA from the two independent measurements (event time and in-kernel clock): 16 = the pipe never idles
    printf("%-44s %d waves/SIMD: %7.1f TFLOP/s = %.3f of 2 500   in-kernel clock %.2f GHz   SIMD cycles per MFMA %.2f\n", name, threads / 256, tf,
           tf / 2500.0, ghz, ghz * 1e9 * 1024.0 * 16384.0 / (tf * 1e12));
    hipFree(out); hipFree(clk);
}

int main() {
    const size_t n = 65536 + 16;
    std::vector<_Float16> hr(n * 8), hz(n * 8, (_Float16)0.f);
    srand(7);
    for (auto& v : hr) v = (_Float16)(((rand() & 0xFFFF) / 32768.0f - 1.0f) * 0.5f);   // uniform [-0.5, 0.5): bounded sums, full-range mantissas
    f16x8 *dr, *dz;
    hipMalloc(&dr, n * 16); hipMalloc(&dz, n * 16);
    hipMemcpy(dr, hr.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(dz, hz.data(), n * 16, hipMemcpyHostToDevice);
    for (int th : {256, 512}) {
        run<false>("MFMA only, random operands", th, dr);
        run<true>("MFMA + the GEMM's LDS fragment reads, random", th, dr);
        run<true, true>("... + its LDS-DMA requests (8 KiB / 64 MFMAs)", th, dr);
        run<false>("MFMA only, zero operands", th, dz);
        run<false, false, 1>("MFMA only, random, no operand shared by neighbours", th, dr);
        run<false, false, 2>("MFMA only, random, same two operands throughout", th, dr);
        run<false>("MFMA only, random operands (again)", th, dr);
    }
    return 0;
}
