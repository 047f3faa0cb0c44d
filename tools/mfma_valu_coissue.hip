// How do the matrix pipe and the vector ALU of ONE SIMD share time between its two waves?  (Round 6: the streaming attention kernel's unit costs
// what its MFMAs, exps, LDS reads and requests cost one after the other -- profiles/r06/attention_stream_ablation.txt -- as if nothing overlapped.)
// 256 workgroups x 512 threads = two waves per SIMD (wave w and w + 4 share SIMD w % 4).  Every wave runs `iters` iterations of a body and stamps
// s_memtime around its loop; the host prints shader cycles per iteration for wave 0 (first wave of SIMD 0) and wave 4 (its partner).
//   body A: 12 independent v_mfma_f32_16x16x32_f16             body V: 16 v_exp_f32 + 24 v_fma_f32 (independent chains)
//   modes: 0 A|A (both waves MFMA only)   1 V|V   2 A|V (wave w MFMA only, wave w+4 VALU only)   3 A;V|A;V (blocks in sequence, both waves)
//          4 A,V interleaved instruction by instruction (1 MFMA, then 3-4 VALU), both waves   5 A;V on ONE wave per SIMD (256 threads)
//          6 interleaved on ONE wave per SIMD   7 / 8 as 3 with s_setprio 0 in front of the MFMA block and 2 / 1 in front of the VALU block
//          9 as 2 with the VALU-only wave at s_setprio 3
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters, float a, float b) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[12];
    float x[16], y[24];
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(a + i); fb[i] = (_Float16)(b + threadIdx.x * 0.001f); }
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{a, b, a, b};
    for (int i = 0; i < 16; ++i) x[i] = a * 0.01f + i * 0.001f + threadIdx.x * 1e-6f;
    for (int i = 0; i < 24; ++i) y[i] = b + i;
    const bool split = MODE == 2 || MODE == 9;
    const bool do_a = MODE == 0 || (split && wave < 4) || (MODE >= 3 && !split);
    const bool do_v = MODE == 1 || (split && wave >= 4) || (MODE >= 3 && !split);
    if (MODE == 9 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4 || MODE == 6) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
                if (i < 4) { x[4 * i] = __builtin_amdgcn_exp2f(x[4 * i]); x[4 * i + 1] = __builtin_amdgcn_exp2f(x[4 * i + 1]); }
                else if (i < 8) { x[4 * (i - 4) + 2] = __builtin_amdgcn_exp2f(x[4 * (i - 4) + 2]); x[4 * (i - 4) + 3] = __builtin_amdgcn_exp2f(x[4 * (i - 4) + 3]); }
                y[2 * i] = fmaf(y[2 * i], a, b);
                y[2 * i + 1] = fmaf(y[2 * i + 1], a, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (do_a) {
                if (MODE == 7 || MODE == 8) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (do_v) {
                if (MODE == 7 || MODE == 8) __builtin_amdgcn_s_setprio(MODE == 7 ? 2 : 1);
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
#pragma unroll
                for (int i = 0; i < 24; ++i) y[i] = fmaf(y[i], a, b);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 24; ++i) s += y[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE> void run(const char* name, int threads) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * sizeof(unsigned long long)); hipMalloc(&sink, 256 * 512 * sizeof(float));
    hipMemset(d, 0, 256 * 8 * sizeof(unsigned long long));
    const int iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink, 200, 1.0001f, 0.5f);   // warm-up
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double w0 = 0, w4 = 0;
    for (int b = 0; b < 256; ++b) { w0 += (double)h[b * 8]; w4 += (double)h[b * 8 + 4]; }
    printf("%-58s %8.1f us | cycles per iteration: wave 0 %7.1f  wave 4 %7.1f  (%.2f GHz)\n", name, ms * 1e3, w0 / 256 / iters,
           threads > 256 ? w4 / 256 / iters : 0.0, w0 / 256 / (ms * 1e6));
    hipFree(d); hipFree(sink);
}

int main() {
    printf("body A = 12 MFMA 16x16x32 f16 (192 matrix-pipe cycles); body V = 16 v_exp_f32 + 24 v_fma_f32 (16 x 8 + 24 x 4 = 224 issue cycles)\n");
    run<0>("0  A | A   both waves MFMA only", 512);
    run<1>("1  V | V   both waves VALU only", 512);
    run<2>("2  A | V   wave w MFMA only, wave w+4 VALU only", 512);
    run<3>("3  A;V | A;V   blocks in sequence, both waves", 512);
    run<4>("4  A,V interleaved per instruction, both waves", 512);
    run<5>("5  A;V   one wave per SIMD", 256);
    run<6>("6  A,V interleaved, one wave per SIMD", 256);
    run<7>("7  A;V | A;V with s_setprio 0 before A, 2 before V", 512);
    run<8>("8  A;V | A;V with s_setprio 0 before A, 1 before V", 512);
    run<9>("9  A | V with the VALU wave at s_setprio 3", 512);
    return 0;
}
