"""Stand-alone timing of the attention kernels: ATT_B samples x 16 heads x ATT_T tokens, head_dim 88 (defaults 64 x 1025: the streaming
kernel of 448^2 images; ATT_B=255 ATT_T=257: the resident kernel of the headline).  CGPT_LIB_PATH selects an experiment build.
Run on the GPU box:  python tools/attention_stream_bench.py"""
import sys, os; sys.path.insert(0, os.getcwd())
import ctypes as C, torch
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
def P(t): return C.c_void_p(t.data_ptr())
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, hd, T = int(os.environ.get("ATT_B", "64")), 16, 88, int(os.environ.get("ATT_T", "1025"))
TQ, TK = int(os.environ.get("ATT_TQ", T)), int(os.environ.get("ATT_TK", T))       # (ATT_TQ / ATT_TK: queries and keys apart, e.g. 1024 x 1025: no lone query)
ld = 3 * H * hd
qkv = (torch.randn(B, T, ld, device=DEV) * 0.7).half()
out = torch.zeros(B, T, H * hd, device=DEV, dtype=torch.float16)
f = lambda: _lib.check(L.cgpt_attention_f16(C.c_void_p(qkv.data_ptr()), ld, C.c_void_p(qkv.data_ptr() + 2 * H * hd), C.c_void_p(qkv.data_ptr() + 4 * H * hd), ld, P(out), H * hd, B, H, hd, TQ, TK, hd ** -0.5, st()))
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
fl = 4.0 * B * H * TQ * TK * hd
print(f"{os.environ.get('CGPT_LIB_PATH','product').split('/')[-1]} attention B{B} {TQ}x{TK}: {us:.1f} us  {fl / us / 1e6:.1f} TF/s (algorithmic, hd 88)", flush=True)
