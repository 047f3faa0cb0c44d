"""Randomised shape sweep of the attention entry point against an fp32 reference: 48 (head_dim, heads, samples, Tq, Tk) draws with Tk > 288 (the
streaming kernel; Tq = 256 k + 1 draws go through the carried lone query), a dominant key for the LAST query, two launches compared bit for bit.
Round 6: 48 / 48 within 1.3e-3 of fp32, identical bits.   python tools/attention_shape_sweep.py"""
import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ctypes as C, random, torch
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
def P(t): return C.c_void_p(t.data_ptr())
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
def ref(q, k, v, heads, hd, scale):
    B, Tq, _ = q.shape; Tk = k.shape[1]
    qh = q.float().view(B, Tq, heads, hd).permute(0, 2, 1, 3); kh = k.float().view(B, Tk, heads, hd).permute(0, 2, 1, 3); vh = v.float().view(B, Tk, heads, hd).permute(0, 2, 1, 3)
    return (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).permute(0, 2, 1, 3).reshape(B, Tq, heads * hd)
rng = random.Random(7)
worst = 0.0
for it in range(48):
    hd = rng.choice([64, 88]); heads = rng.choice([1, 2, 3, 5]); B = rng.choice([1, 2, 3, 9, 33, 70])
    Tq = rng.choice([1, 17, 257, 300, 513, 769, 1025, 1281]); Tk = rng.randint(289, 1400)
    if B * heads * Tq * Tk > 3e8: B = max(1, int(3e8 // (heads * Tq * Tk)))
    D = heads * hd
    g = torch.Generator().manual_seed(it)
    q = torch.randn(B, Tq, D, generator=g).half(); k = torch.randn(B, Tk, D, generator=g).half(); v = torch.randn(B, Tk, D, generator=g).half()
    j = rng.randrange(Tk); q[0, Tq - 1, :hd] *= 4; k[0, j, :hd] = q[0, Tq - 1, :hd] * 0.7
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    od = torch.full((B, Tq, D), float("nan"), device=DEV, dtype=torch.float16)
    _lib.check(L.cgpt_attention_f16(P(qd), D, P(kd), P(vd), D, P(od), D, B, heads, hd, Tq, Tk, hd ** -0.5, st()))
    od2 = torch.full((B, Tq, D), float("nan"), device=DEV, dtype=torch.float16)
    _lib.check(L.cgpt_attention_f16(P(qd), D, P(kd), P(vd), D, P(od2), D, B, heads, hd, Tq, Tk, hd ** -0.5, st()))
    torch.cuda.synchronize()
    err = float((od.cpu().float() - ref(q, k, v, heads, hd, hd ** -0.5)).abs().max())
    same = torch.equal(od, od2)
    worst = max(worst, err)
    print(f"{it:2d} hd{hd} h{heads} B{B} {Tq}x{Tk} spike@{j}: max err {err:.2e} {'same bits' if same else 'DIFFERENT BITS'}", flush=True)
    assert err < 6e-3 and same and not torch.isnan(od).any()
print("sweep ok, worst", worst)
