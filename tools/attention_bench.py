"""Standalone timing of the ViT-G attention kernel (200 samples x 16 heads x 257 tokens, head_dim 88); also the command
used under rocprofv3 --pmc for profiles/r01.  Run on the GPU box:  python tools/attention_bench.py"""
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ctypes as C, torch
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
def P(t): return C.c_void_p(t.data_ptr())
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
import os
B, H, hd, T = int(os.environ.get("CGPT_ATT_B", "200")), 16, 88, int(os.environ.get("CGPT_ATT_T", "257"))   # CGPT_ATT_B=255: the bench batch
ld = 3 * H * hd
qkv = (torch.randn(B, T, ld, device=DEV) * 0.7).half()
out = torch.zeros(B, T, H * hd, device=DEV, dtype=torch.float16)
f = lambda: _lib.check(L.cgpt_attention_f16(C.c_void_p(qkv.data_ptr()), ld, C.c_void_p(qkv.data_ptr() + 2 * H * hd), C.c_void_p(qkv.data_ptr() + 4 * H * hd), ld, P(out), H * hd, B, H, hd, T, T, hd ** -0.5, st()))
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
res = []
for r in range(5):
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 10 * 1e3)
print(f"attention B={B} T={T}: us per launch median {sorted(res)[2]:.1f} min {min(res):.1f}")
