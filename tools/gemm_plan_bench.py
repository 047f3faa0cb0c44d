"""Column plans of the two-phase GEMM (gemm9.hip, round 4) at a rank's shard sizes: the four ViT-G GEMM shapes at M = samples x 257 rows for
25 / 50 / 100 samples (the 8- / 4- / 2-GPU shard of one image's n0 + n = 200 draws), classic tiling (gemm_plan 0) against the automatic
plan (1), the best plan regardless of the model (2) and the forced 192- / 128- / 64-column tilings (13 / 12 / 11), interleaved rounds in
one process; us per launch (median) and the ratio to the classic tiling.  The forced tilings calibrate the cost of a tile by its width
(kTileCost in gemm9.hip).   python tools/gemm_plan_bench.py [rounds] [samples,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, statistics, torch
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
def P(t): return C.c_void_p(t.data_ptr()) if t is not None else None
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SAMPLES = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [25, 50, 100]
PLANS = [int(v) for v in os.environ.get("CGPT_PLANS", "0,1,2,13,12,11").split(",")]
SHAPES = {"qkv": (4224, 1408, 0), "proj": (1408, 1408, 0), "fc1+gelu": (6144, 1408, 1), "fc2": (1408, 6144, 0)}
for ns in SAMPLES:
    M = ns * 257; Mp = (M + 255) // 256 * 256
    total = {pl: 0.0 for pl in PLANS}
    for name, (N, K, epi) in SHAPES.items():
        g = torch.Generator(device=DEV).manual_seed(N + K)
        A = (torch.randn(Mp, K, device=DEV, generator=g) * 0.7).half()
        W = (torch.randn((N + 255) // 256 * 256, K, device=DEV, generator=g) * 0.02).half()
        b = torch.randn(N, device=DEV, generator=g) * 0.1
        out = torch.zeros(M, N, device=DEV, dtype=torch.float16)
        def launch(plan):
            _lib.check(L.cgpt_set_option(b"gemm_plan", plan))
            _lib.check(L.cgpt_linear_f16(P(A), K, P(W), K, P(b), P(out), N, None, N, M, N, K, epi, st()))
        res = {pl: [] for pl in PLANS}
        for pl in PLANS:
            for _ in range(3): launch(pl)
        torch.cuda.synchronize()
        for r in range(ROUNDS):
            for pl in PLANS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): launch(pl)
                e1.record(); torch.cuda.synchronize()
                res[pl].append(e0.elapsed_time(e1) * 50.0)
        med = {pl: statistics.median(v) for pl, v in res.items()}
        for pl in PLANS: total[pl] += med[pl]
        print(f"{ns:3d} samples (M = {M:6d}) {name:9s} " + "  ".join(f"plan {pl:2d}: {med[pl]:7.1f} us ({med[pl] / med[PLANS[0]]:.3f})" for pl in PLANS), flush=True)
    print(f"{ns:3d} samples: one block's four GEMMs " + "  ".join(f"plan {pl:2d}: {total[pl]:7.1f} us ({total[pl] / total[PLANS[0]]:.3f})" for pl in PLANS), flush=True)
_lib.check(L.cgpt_set_option(b"gemm_plan", 1))
