// diag.hip -- the yardstick for the MFMA roofline fraction: the dense fp16 MFMA rate THIS device sustains on random operands when it
// does nothing else (v_mfma_f32_16x16x32_f16 back to back, operands in registers, no LDS, no memory).  The data sheet's 2.5 PFLOP/s
// is 1 024 SIMDs x 1 024 FLOP per clock at 2.4 GHz; under an MFMA-dense load the chip holds 1.8-1.95 GHz (box to box), so a kernel's
// fraction of the data-sheet peak mixes its own efficiency with the device's clock.  bench.py reports both (profiles/r04/mfma_sustained.txt;
// tools/mfma_sustained_bench.hip is the stand-alone form with the LDS / LDS-DMA variants).
#include "kernels.h"
#include <vector>

namespace cgpt {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(512) void mfma_sustained_kernel(float* out, unsigned long long* clk, int iters, unsigned seed) {
    // operands: a cheap per-lane hash -> uniform fp16 in [-0.5, 0.5), full-range mantissas (zero or constant operands draw far less power)
    auto rnd = [&](unsigned k) {
        unsigned x = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + k * 2246822519u + seed);
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        return (_Float16)(((float)(x & 0xFFFF) * (1.0f / 65536.0f)) - 0.5f);
    };
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = rnd(i * 8 + e); b[i][e] = rnd(64 + i * 8 + e); }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

}  // namespace

// Runs the loop back to back for `seconds` (the first 3/4 settle the clock, the last 1/4 is timed) on the CURRENT device's null stream -- the
// caller selects the device (hipSetDevice / torch.cuda.set_device) first.  Synchronous; allocates and frees its own 0.5 MB.
// tflops: dense fp16 MFMA rate; clock_ghz: in-kernel d(s_memtime) / d(s_memrealtime) x 100 MHz.  Every HIP call is checked: a launch that
// fails (e.g. a code object for another architecture) is an error, never a zero-duration "measurement".
hipError_t run_mfma_sustained(double seconds, double* tflops, double* clock_ghz) {
    if (tflops) *tflops = 0.0;
    if (clock_ghz) *clock_ghz = 0.0;
    int dev = 0, cus = 256;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    float* out = nullptr;
    unsigned long long* clk = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipSuccess;
    auto cleanup = [&](hipError_t code) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (out) (void)hipFree(out);
        if (clk) (void)hipFree(clk);
        return code;
    };
#define CGPT_DIAG_TRY(call) do { err = (call); if (err != hipSuccess) return cleanup(err); } while (0)
    CGPT_DIAG_TRY(hipMalloc(&out, (size_t)cus * 512 * sizeof(float)));
    CGPT_DIAG_TRY(hipMalloc(&clk, (size_t)cus * 2 * sizeof(unsigned long long)));
    CGPT_DIAG_TRY(hipMemset(clk, 0, (size_t)cus * 2 * sizeof(unsigned long long)));
    CGPT_DIAG_TRY(hipEventCreate(&e0));
    CGPT_DIAG_TRY(hipEventCreate(&e1));
    const int iters = 20000;                                                // 16 MFMAs per iteration and wave: ~3 ms per launch
    const double flop = (double)cus * 8.0 * iters * 16.0 * (2.0 * 16 * 16 * 32);
    auto launch = [&]() -> hipError_t {
        hipLaunchKernelGGL(mfma_sustained_kernel, dim3(cus), dim3(512), 0, 0, out, clk, iters, 12345u);
        return hipGetLastError();
    };
    CGPT_DIAG_TRY(launch());
    CGPT_DIAG_TRY(hipDeviceSynchronize());                                  // the first launch ran (faults surface here)
    CGPT_DIAG_TRY(hipEventRecord(e0));
    CGPT_DIAG_TRY(launch());
    CGPT_DIAG_TRY(hipEventRecord(e1));
    CGPT_DIAG_TRY(hipEventSynchronize(e1));
    float one = 0.f;
    CGPT_DIAG_TRY(hipEventElapsedTime(&one, e0, e1));
    if (!(one > 0.f)) return cleanup(hipErrorUnknown);
    if (seconds < 0.2) seconds = 0.2;
    const int warm = (int)(seconds * 750.0 / one) + 1, timed = (int)(seconds * 250.0 / one) + 1;
    for (int i = 0; i < warm; ++i) CGPT_DIAG_TRY(launch());
    CGPT_DIAG_TRY(hipEventRecord(e0));
    for (int i = 0; i < timed; ++i) CGPT_DIAG_TRY(launch());
    CGPT_DIAG_TRY(hipEventRecord(e1));
    CGPT_DIAG_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    CGPT_DIAG_TRY(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)cus * 2);
    CGPT_DIAG_TRY(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
#undef CGPT_DIAG_TRY
    double cyc = 0, real = 0;
    for (int i = 0; i < cus; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    if (!(ms > 0.f) || !(real > 0)) return cleanup(hipErrorUnknown);      // no time elapsed or no workgroup wrote its clock: not a measurement
    if (tflops) *tflops = flop * timed / (ms * 1e-3) / 1e12;
    if (clock_ghz) *clock_ghz = cyc / real * 0.1;
    return cleanup(hipSuccess);
}

}  // namespace cgpt
