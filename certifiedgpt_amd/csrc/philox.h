// philox.h -- counter-based Gaussian stream of the build (product side; oracle/philox.py restates it).
//
// Replaces `torch.randn_like(batch, device='cuda') * sigma` (reference randomized_smoothing/smoothing.py:96),
// whose device-global stream depends on batch size and device count.  Element e of Monte-Carlo sample s under
// seed S uses Philox4x32-10(counter = (e/4, hi_word, s_lo, s_hi), key = (S_lo, S_hi)); words (r0,r1) and (r2,r3)
// feed one Box-Muller pair each, so any partition of the sample range over batches / GPUs sees the same draws.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cgpt {

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// 4 standard normals for element group g of `stream` (a sample index or a tensor id).
__device__ inline float4 normal4(uint64_t seed, uint64_t stream, uint32_t g, uint32_t hi_word) {
    const u32x4 r = philox4x32_10(g, hi_word, (uint32_t)stream, (uint32_t)(stream >> 32),
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    const float k = 5.9604644775390625e-08f;  // 2^-24
    const float u1a = (float)((r.x >> 8) + 1u) * k, u2a = (float)(r.y >> 8) * k;
    const float u1b = (float)((r.z >> 8) + 1u) * k, u2b = (float)(r.w >> 8) * k;
    const float ra = sqrtf(-2.0f * logf(u1a)), rb = sqrtf(-2.0f * logf(u1b));
    float sa, ca, sb, cb;
    sincospif(2.0f * u2a, &sa, &ca);
    sincospif(2.0f * u2b, &sb, &cb);
    return make_float4(ra * ca, ra * sa, rb * cb, rb * sb);
}

}  // namespace cgpt
