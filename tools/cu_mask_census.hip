// Which XCD does a CU-mask bit belong to?  hipExtStreamCreateWithCUMask(stream, words, mask): bit i of the mask enables "CU i" -- this tool
// launches a census kernel on streams with a few mask patterns and prints how many workgroups ran on each XCC (HW_REG_XCC_ID), so that a
// caller can build "even XCDs" / "odd XCDs" masks without guessing (round 5, profiles/r05/xcd_group_probe.txt).
// Build + run on the GPU box:  hipcc -O2 --offload-arch=gfx950 tools/cu_mask_census.hip -o /tmp/cumask && /tmp/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void census(unsigned* xcc_of_wg, unsigned* hwid_of_wg) {
    unsigned x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    // keep the workgroup resident for a moment so that the grid spreads over every enabled CU
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) {}              // 20 us
    if (threadIdx.x == 0) { xcc_of_wg[blockIdx.x] = x & 0xF; hwid_of_wg[blockIdx.x] = h; }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%-36s stream creation FAILED\n", name); return; }
    const int n = 2048;
    unsigned *d, *dh;
    hipMalloc(&d, n * 4); hipMalloc(&dh, n * 4);
    hipMemsetAsync(d, 0xFF, n * 4, s);
    hipLaunchKernelGGL(census, dim3(n), dim3(256), 65536, s, d, dh);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(n), hh(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hh.data(), dh, n * 4, hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int i = 0; i < n; ++i) cnt[h[i] & 15]++;
    // distinct (xcc, se, cu) triples seen = CUs actually used (HW_ID: cu_id bits 11:8, sh 12, se 15:13 on gfx9; printed raw-masked)
    std::vector<unsigned> seen;
    for (int i = 0; i < n; ++i) { unsigned key = ((h[i] & 15) << 16) | (hh[i] & 0xFF00); bool f = false; for (auto k : seen) if (k == key) { f = true; break; } if (!f) seen.push_back(key); }
    int bits = 0; for (auto w : mask) bits += __builtin_popcount(w);
    printf("%-36s bits set %3d  distinct CUs seen %3zu  workgroups per XCC:", name, bits, seen.size());
    for (int x = 0; x < 8; ++x) printf(" %4d", cnt[x]);
    printf("\n");
    hipFree(d); hipFree(dh); hipStreamDestroy(s);
}

int main() {
    std::vector<uint32_t> all(8, 0xFFFFFFFFu);
    run("all 256 bits", all);
    std::vector<uint32_t> lo(8, 0), hi(8, 0), even(8, 0), odd(8, 0), mod0(8, 0), first32(8, 0), blk1(8, 0);
    for (int i = 0; i < 256; ++i) {
        if (i < 128) lo[i / 32] |= 1u << (i % 32); else hi[i / 32] |= 1u << (i % 32);
        if ((i & 1) == 0) even[i / 32] |= 1u << (i % 32); else odd[i / 32] |= 1u << (i % 32);
        if ((i & 7) == 0) mod0[i / 32] |= 1u << (i % 32);
        if (i < 32) first32[i / 32] |= 1u << (i % 32);
        if (i >= 32 && i < 64) blk1[i / 32] |= 1u << (i % 32);
    }
    run("bits 0..127", lo);
    run("bits 128..255", hi);
    run("even bits", even);
    run("odd bits", odd);
    run("bits with i % 8 == 0", mod0);
    run("bits 0..31", first32);
    run("bits 32..63", blk1);
    return 0;
}
