"""Phase stamps (s_memtime) of the streaming attention kernel; needs the stamps build: make -C certifiedgpt_amd/csrc STAMPS=1
(writes scratch/libcgpt_stamp.so; STAMP_LIB selects another one, e.g. one built with EXTRA=-DCGPT_ATT_LONE_SPLIT=0).
Run on the GPU box:  python tools/attention_stream_stamps.py
The last column is the time a wave spends, as whole work items, inside the blocks that deal with the lone query of Tq = 256 k + 1:
the block that carries it (product) or, with STAMP_LONE_MODE=block for a -DCGPT_ATT_LONE_CARRIED=0 build, the block that holds only it."""
import sys, os; sys.path.insert(0, os.getcwd())
import certifiedgpt_amd._lib as LL
LL.LIB_PATH = os.environ.get("STAMP_LIB", os.getcwd() + "/scratch/libcgpt_stamp.so")
import ctypes as C, torch, numpy as np
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
L.cgpt_debug_set_gemm_stamps.argtypes = [C.c_void_p]
def P(t): return C.c_void_p(t.data_ptr())
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, hd, T = int(os.environ.get("ATT_B", "64")), 16, 88, int(os.environ.get("ATT_T", "1025"))
ld = 3 * H * hd
qkv = (torch.randn(B, T, ld, device=DEV) * 0.7).half()
out = torch.zeros(B, T, H * hd, device=DEV, dtype=torch.float16)
f = lambda: _lib.check(L.cgpt_attention_f16(C.c_void_p(qkv.data_ptr()), ld, C.c_void_p(qkv.data_ptr() + 2 * H * hd), C.c_void_p(qkv.data_ptr() + 4 * H * hd), ld, P(out), H * hd, B, H, hd, T, T, hd ** -0.5, st()))
for _ in range(3): f()
torch.cuda.synchronize()
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=DEV)
L.cgpt_debug_set_gemm_stamps(P(dbg))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize(); L.cgpt_debug_set_gemm_stamps(None)
us = e0.elapsed_time(e1) * 1e3
d = dbg.cpu().numpy().reshape(256, 8, 8).astype(np.float64)
print(f"attention T={T} {us:.0f} us; ticks per wave median {np.median(d[..., 0]):.0f} = {np.median(d[..., 0]) / us:.0f} ticks/us")
names = ["setup+epilogue", "vmcnt wait", "barrier", "request issue", "units", "inside lone blocks"]
for w in range(8):
    tot = d[:, w, 0].mean()
    print(f" wave {w}: total {tot:.0f} ticks | " + "  ".join(f"{names[k]} {d[:, w, 1 + k].mean():.0f} ({100 * d[:, w, 1 + k].mean() / tot:.0f}%)" for k in range(6)))
if T % 256 == 1 and T > 256:
    tot, lone = d[:, :, 0].mean(), d[:, :, 6].mean()
    nfull = T // 256
    if os.environ.get("STAMP_LONE_MODE", "carried") == "carried":     # the product: the pair's last full block carries the lone query
        print(f" {os.path.basename(LL.LIB_PATH)}: blocks that carry the lone query take {100 * lone / tot:.1f} % of a wave's time; one such block = "
              f"{lone * (nfull - 1) / (tot - lone):.3f} of a plain full block ({nfull - 1} plain + 1 carrying block per (sample, head))")
    else:                                                             # -DCGPT_ATT_LONE_CARRIED=0 (rounds 2-5): a block of its own
        print(f" {os.path.basename(LL.LIB_PATH)}: lone-query blocks take {100 * lone / tot:.1f} % of a wave's time; one lone block = "
              f"{lone * nfull / (tot - lone):.3f} of a full block ({nfull} full blocks + 1 lone block per (sample, head))")
