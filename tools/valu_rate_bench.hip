// VALU issue-rate microbenchmark behind the GELU epilogue design (profiles/r01/gemm_variants.txt):
// ns per wave instruction for independent instruction streams, one (256 threads) or two (512 threads) waves per SIMD, 256 workgroups.
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/valu_rate_bench.hip -o /tmp/valu && /tmp/valu
// MI355X, two waves per SIMD: v_fma_f32 4.2 ns, v_pk_fma_f32 4.5, v_pk_mul_f32 4.2, v_rcp_f32 8.3, v_exp_f32 8.2, v_max_f32 2.1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    float x[8]; f32x2 y[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + i + threadIdx.x; y[i] = f32x2{a + i, b + threadIdx.x}; }
    const f32x2 ca = {a, a}, cb = {b, b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) x[i] = fmaf(x[i], a, b);
            if (KIND == 1) y[i] = __builtin_elementwise_fma(y[i], ca, cb);
            if (KIND == 2) x[i] = __builtin_amdgcn_rcpf(x[i]);
            if (KIND == 3) x[i] = __builtin_amdgcn_exp2f(x[i]);
            if (KIND == 4) y[i] = y[i] * ca;
            if (KIND == 5) x[i] = fmaxf(x[i], a);
            if (KIND == 6) { x[i] = fmaf(x[i], a, b); y[i] = __builtin_elementwise_fma(y[i], ca, cb); }   // mixed
            if (KIND == 7) { x[i] = __builtin_amdgcn_rcpf(x[i]); y[i] = __builtin_elementwise_fma(y[i], ca, cb); }   // trans + pk
            if (KIND == 8) { x[i] = __builtin_amdgcn_rcpf(x[i]); x[(i + 4) & 7] = fmaf(x[(i + 4) & 7], a, b); }   // trans + fma
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + y[i][0] + y[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long*)out)[100000] = t1 - t0;
}
template <int KIND> void run(const char* name, int threads, int per) {
    float* d; hipMalloc(&d, 1 << 22);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ticks; hipMemcpy(&ticks, (char*)d + 100000 * 8, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 32 * per;
    printf("%-22s threads %3d: %.2f us, %.1f ns per wave-instr per wave (memtime ticks %llu)\n", name, threads, ms * 1e3, ms * 1e6 / n, ticks);
    hipFree(d);
}
int main() {
    for (int th : {256, 512}) {
        run<0>("v_fma_f32", th, 1); run<1>("v_pk_fma_f32", th, 1); run<4>("v_pk_mul_f32", th, 1); run<2>("v_rcp_f32", th, 1); run<3>("v_exp_f32", th, 1);
        run<5>("v_max_f32", th, 1); run<6>("fma + pk_fma", th, 2); run<7>("rcp + pk_fma", th, 2); run<8>("rcp + fma", th, 2);
    }
    return 0;
}
