"""Raw GEMM A/B harness behind profiles/r01/gemm_variants.txt: times cgpt_linear_f16 on the ViT-G shapes for a chosen kernel
(cgpt_set_option "gemm_kernel": 1 v1, 2/3 v2, 4/5 v3, 6/7 v4, 8 v5, 9/10 v6, 11 v8) and ablation flags ("gemm_ablate": 1 no in-loop loads,
2 no epilogue stores).  Run on the GPU box:  python tools/gemm_bench.py"""
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ctypes as C, torch
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
def P(t): return C.c_void_p(t.data_ptr()) if t is not None else None
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(name, M, N, K, epi, kern, abl=0, iters=10):
    Mp = (M + 255) // 256 * 256
    A = (torch.randn(Mp, K, device=DEV) * 0.5).half(); W = (torch.randn((N + 255) // 256 * 256, K, device=DEV) * 0.05).half()
    bias = torch.randn(N, device=DEV)
    out = torch.zeros(M, N, device=DEV, dtype=torch.float16 if epi < 2 else torch.float32)
    aux = out if epi == 3 else None
    _lib.check(L.cgpt_set_option(b"gemm_kernel", kern)); _lib.check(L.cgpt_set_option(b"gemm_ablate", abl))
    f = lambda: _lib.check(L.cgpt_linear_f16(P(A), K, P(W), K, P(bias), P(out), N, P(aux), N, M, N, K, epi, st()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    _lib.check(L.cgpt_set_option(b"gemm_kernel", 0)); _lib.check(L.cgpt_set_option(b"gemm_ablate", 0))
    print(f"{name:8s} M{M} N{N} K{K} epi{epi} kern{kern} abl{abl}: {ms*1e3:7.0f} us  {2.0*M*N*K/ms/1e9:6.0f} TF", flush=True)
if __name__ == "__main__":
    for M in (6682, 51400):
        for name, N, K, epi in (("qkv", 4224, 1408, 0), ("proj", 1408, 1408, 0), ("fc1", 6144, 1408, 1), ("fc2", 1408, 6144, 0)):
            for kern in (1, 2, 4, 6, 8):
                run(name, M, N, K, epi, kern, 0)
    for abl in (0, 1, 2, 3):
        run("fc2", 51400, 1408, 6144, 0, 4, abl)
