"""Round 5: how evenly do the 256 persistent workgroups of the 256 x 256 GEMM finish, and at which clock does each XCD run?
Needs the stamps build (make -C certifiedgpt_amd/csrc STAMPS=1 -> scratch/libcgpt_stamp.so): every workgroup writes its first and last
s_memrealtime (100-MHz ticks) and its shader-cycle count.   CGPT_STAMP_LIB=scratch/libcgpt_stamp.so python3 tools/gemm_wg_balance.py"""
import sys; sys.path.insert(0, ".")
import certifiedgpt_amd._lib as LL
import os; LL.LIB_PATH = os.environ["CGPT_STAMP_LIB"]
import ctypes as C, torch, numpy as np
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
L.cgpt_debug_set_gemm_stamps.argtypes = [C.c_void_p]
def P(t): return C.c_void_p(t.data_ptr()) if t is not None else None
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 65535
for name, N, K, epi in (("qkv", 4224, 1408, 0), ("proj", 1408, 1408, 0), ("fc1", 6144, 1408, 1), ("fc2", 1408, 6144, 0)):
    Mp = (M + 255) // 256 * 256
    A = (torch.randn(Mp, K, device=DEV) * 0.5).half(); W = (torch.randn((N + 255) // 256 * 256, K, device=DEV) * 0.05).half()
    bias = torch.randn(N, device=DEV); out = torch.zeros(M, N, device=DEV, dtype=torch.float16)
    dbg = torch.zeros(256 * 8 * 12 + 512, dtype=torch.int64, device=DEV)
    f = lambda: _lib.check(L.cgpt_linear_f16(P(A), K, P(W), K, P(bias), P(out), N, None, N, M, N, K, epi, st()))
    for _ in range(30): f()
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        dbg.zero_(); torch.cuda.synchronize(); L.cgpt_debug_set_gemm_stamps(P(dbg))
        f(); torch.cuda.synchronize(); L.cgpt_debug_set_gemm_stamps(None)
        r = dbg.cpu().numpy()[256 * 8 * 12:256 * 8 * 12 + 512].reshape(256, 2).astype(np.float64) * 0.01   # us
        b, e = r[:, 0], r[:, 1]
        span = e.max() - b.min()
        tail = (e.max() - e).mean(); head = (b - b.min()).mean()
        xcd = np.arange(256) % 8
        per_xcd_end = [e[xcd == x].mean() - b.min() for x in range(8)]
        cyc = dbg.cpu().numpy()[:256 * 8 * 4].reshape(256, 8, 4)[:, 0, 0].astype(np.float64)   # wave 0: shader cycles from first to last instruction
        ghz = cyc / ((e - b) * 1e3)
        per_xcd_ghz = [ghz[xcd == x].mean() for x in range(8)]
        res.append((span, tail, head, e.max() - np.percentile(e, 50), e.max() - e.min(), per_xcd_end, per_xcd_ghz))
    sp = np.mean([x[0] for x in res])
    print(f"{name}: span {sp:.1f} us; mean idle tail per workgroup {np.mean([x[1] for x in res]):.2f} us ({100 * np.mean([x[1] for x in res]) / sp:.2f} %), "
          f"mean late start {np.mean([x[2] for x in res]):.2f} us, last - median end {np.mean([x[3] for x in res]):.2f} us, last - first end {np.mean([x[4] for x in res]):.2f} us", flush=True)
    print("     mean end by XCD position (us after first begin):", " ".join(f"{v:.1f}" for v in np.mean([x[5] for x in res], axis=0)), flush=True)
    print("     in-kernel clock by XCD position (GHz):           ", " ".join(f"{v:.3f}" for v in np.mean([x[6] for x in res], axis=0)), flush=True)
