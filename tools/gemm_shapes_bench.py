"""Raw timing of the four ViT-G GEMM shapes at the bench's batch (255 samples = 65 535 rows), interleaved rounds in ONE process: fc1 with
its GELU epilogue, fc1 without GELU (what the epilogue costs), fc2, qkv, proj.  us per launch and TFLOP/s (median and min over rounds).
CGPT_LIB_PATH selects another build of the library for an A/B on the same box; CGPT_GEMM_KERNELS="14,16" interleaves several
gemm_kernel overrides in the same rounds (same box, same process: the only comparison that means anything).
python tools/gemm_shapes_bench.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, statistics, torch
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
def P(t): return C.c_void_p(t.data_ptr()) if t is not None else None
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 65535; Mp = 65536
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
KERNELS = [int(k) for k in os.environ.get("CGPT_GEMM_KERNELS", "0").split(",")]
def mk(N, K):
    g = torch.Generator(device=DEV).manual_seed(N + K)
    return ((torch.randn(Mp, K, device=DEV, generator=g) * 0.7).half(), (torch.randn((N + 255) // 256 * 256, K, device=DEV, generator=g) * 0.02).half(),
            torch.randn(N, device=DEV, generator=g) * 0.1, torch.zeros(M, N, device=DEV, dtype=torch.float16))
ops = {"fc1": mk(6144, 1408), "fc2": mk(1408, 6144), "qkv": mk(4224, 1408), "proj": mk(1408, 1408)}
def launch(name, epi, abl, kern=0):
    A, W, b, out = ops[name]; N, K = W.shape[0] if name != "qkv" and name != "proj" else {"qkv": 4224, "proj": 1408}[name], A.shape[1]
    N = {"fc1": 6144, "fc2": 1408, "qkv": 4224, "proj": 1408}[name]
    _lib.check(L.cgpt_set_option(b"gemm_ablate", abl)); _lib.check(L.cgpt_set_option(b"gemm_kernel", kern))
    _lib.check(L.cgpt_linear_f16(P(A), K, P(W), K, P(b), P(out), N, None, N, M, N, K, epi, st()))
base = [("fc1 + GELU", "fc1", 1, 0), ("fc1 no GELU", "fc1", 0, 0), ("fc2", "fc2", 0, 0),
        ("qkv", "qkv", 0, 0), ("proj", "proj", 0, 0)]
variants = [(f"{v[0]} k{k}", v[1], v[2], v[3], k) for v in base for k in KERNELS]
res = {v[0]: [] for v in variants}
for v in variants:
    for _ in range(3): launch(*v[1:])
torch.cuda.synchronize()
for r in range(ROUNDS):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): launch(*v[1:])
        e1.record(); torch.cuda.synchronize()
        res[v[0]].append(e0.elapsed_time(e1) * 100.0)
_lib.check(L.cgpt_set_option(b"gemm_ablate", 0)); _lib.check(L.cgpt_set_option(b"gemm_kernel", 0))
for v in variants:
    N, K = {"fc1": (6144, 1408), "fc2": (1408, 6144), "qkv": (4224, 1408), "proj": (1408, 1408)}[v[1]]
    us = res[v[0]]
    print(f"{v[0]:24s} us per launch: median {statistics.median(us):7.1f}  min {min(us):7.1f}   {2.0 * M * N * K / statistics.median(us) / 1e6:6.0f} TF", flush=True)
