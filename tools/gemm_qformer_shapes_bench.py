"""Raw timing of the Q-Former's linears at the bench's batch (255 samples x 32 queries = 8 160 rows) under the automatic kernel choice and under
the forced tiles (cgpt_set_option "gemm_kernel": 14 = 256x256 two-phase, 3 = 256x128 direct-to-LDS, 1 = 128x128 register-staged), interleaved in one
process; us per launch (median over rounds), tiles of 256x256 per shape.  VERDICT r5 item 7.   python tools/gemm_qformer_shapes_bench.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, statistics, torch
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
def P(t): return C.c_void_p(t.data_ptr()) if t is not None else None
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 8160; Mp = 8192
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
# name: (N, K, epilogue)  0 = fp16 out, 1 = GELU, 3 = + residual (fp32 aux)
SHAPES = {"q/k/v fused 768->2304": (2304, 768, 0), "self-output 768->768 + resid": (768, 768, 3), "ffn up 768->3072 GELU": (3072, 768, 1),
          "ffn down 3072->768 + resid": (768, 3072, 3), "llama_proj 768->4096": (4096, 768, 0)}
ops = {}
for name, (N, K, epi) in SHAPES.items():
    g = torch.Generator(device=DEV).manual_seed(N + K)
    ops[name] = ((torch.randn(Mp, K, device=DEV, generator=g) * 0.7).half(), (torch.randn((N + 255) // 256 * 256, K, device=DEV, generator=g) * 0.02).half(),
                 torch.randn(N, device=DEV, generator=g) * 0.1,
                 torch.zeros(Mp, N, device=DEV, dtype=torch.float32 if epi == 3 else torch.float16),      # EPI_RESID writes fp32 (kernels.h)
                 torch.randn(Mp, N, device=DEV, generator=g) if epi == 3 else None)
    A, W, b, out, aux = ops[name]
    # operand shapes as the 256-row kernels read / write them: A for round_up(M, 256) rows, W for round_up(N, 256) rows, bias N, out M x N
    assert A.shape == (Mp, K) and W.shape[0] % 256 == 0 and W.shape[0] >= N and b.numel() == N and out.shape == (Mp, N) and Mp >= M
    assert out.element_size() == (4 if epi == 3 else 2) and (aux is None or (aux.shape == (Mp, N) and aux.dtype == torch.float32))
def launch(name, kern):
    N, K, epi = SHAPES[name]; A, W, b, out, aux = ops[name]
    _lib.check(L.cgpt_set_option(b"gemm_kernel", kern))
    _lib.check(L.cgpt_linear_f16(P(A), K, P(W), K, P(b), P(out), N, P(aux), N, M, N, K, epi, st()))
KERNELS = [0, 14, 3, 1]
res = {(n, k): [] for n in SHAPES for k in KERNELS}
for key in res:
    for _ in range(3): launch(*key)
torch.cuda.synchronize()
for r in range(ROUNDS):
    for key in res:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): launch(*key)
        e1.record(); torch.cuda.synchronize()
        res[key].append(e0.elapsed_time(e1) * 50.0)
_lib.check(L.cgpt_set_option(b"gemm_kernel", 0))
for n, (N, K, epi) in SHAPES.items():
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"{n:30s} {t256:4d} tiles of 256x256 | " + "  ".join(f"k{k}: {statistics.median(res[(n, k)]):6.1f} us" for k in KERNELS) +
          f"   ({2.0 * M * N * K / statistics.median(res[(n, 0)]) / 1e6:5.0f} TF auto)", flush=True)
