"""GPU unit tests of the raw HIP kernels through the C-ABI (include/cgpt.h), each against a plain PyTorch fp32
reference of the same op.  Failures print where the first mismatches are, so one gpurun round trip is enough to
locate an indexing bug."""
import ctypes as C

import numpy as np
import pytest
import torch

import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
from oracle import philox

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gemm_kernels():
    """Kernel overrides this build of libcgpt.so accepts (results are identical for every one): the product library has the
    automatic choice (0), the 128x128 kernel (1), the 256x128 (3), the 256x256 phased (4) and the 256x256 two-phase quadrant kernel (14);
    `make LAB=1` adds 2, 5..12."""
    L = cg.lib()
    lab = L.cgpt_set_option(b"gemm_kernel", 12) == 0
    L.cgpt_set_option(b"gemm_kernel", 0)
    return [k for k in range(15) if k != 13] if lab else [0, 1, 3, 4, 14]


GEMM_KERNELS = _gemm_kernels()


def P(t):
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ru(v, m):
    return (v + m - 1) // m * m


def report(name, got, ref, tol):
    err = (got - ref).abs()
    bad = err > tol
    if bad.any():
        idx = bad.nonzero()[:8].tolist()
        rows = sorted(set(i[0] for i in bad.nonzero().tolist()))[:16]
        msg = (f"{name}: {int(bad.sum())}/{bad.numel()} mismatches > {tol}; max err {float(err.max()):.4g}; "
               f"first idx {idx}; got {[float(got[tuple(i)]) for i in idx[:4]]} ref {[float(ref[tuple(i)]) for i in idx[:4]]}; "
               f"bad rows {rows}")
        pytest.fail(msg)


def run_gemm(M, N, K, bias=True, seed=0, asym=False):
    L = cg.lib()
    g = torch.Generator(device="cpu").manual_seed(seed)
    Mp, Np = ru(M, 256), ru(N, 256)
    A = torch.zeros(Mp, K, dtype=torch.float16)
    W = torch.zeros(Np, K, dtype=torch.float16)
    if asym:   # A = I (first K rows), asymmetric W: catches swapped row/col maps (cdna guide section 3)
        A[:min(M, K), :] = torch.eye(K, dtype=torch.float16)[:min(M, K)]
        W[:N] = (torch.arange(N)[:, None] * 0.5 + torch.arange(K)[None, :] * 0.01).half()
    else:
        A[:M] = (torch.randn(M, K, generator=g) * 0.5).half()
        W[:N] = (torch.randn(N, K, generator=g) * 0.5).half()
    b = torch.randn(N, generator=g) if bias else None
    Ad, Wd = A.to(DEV), W.to(DEV)
    bd = b.to(DEV) if bias else None
    Cd = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(L.cgpt_gemm_f16(P(Ad), K, P(Wd), K, P(bd) if bias else None, P(Cd), N, M, N, K, stream()))
    torch.cuda.synchronize()
    ref = A[:M].float() @ W[:N].float().t()
    if bias:
        ref = ref + b
    return Cd.cpu(), ref


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 384, 192), (1, 200, 64), (257, 1408, 1408), (513, 640, 6144),
                                   (1024, 512, 128), (2000, 384, 192), (1300, 768, 1408), (1029, 256, 64)])
@pytest.mark.parametrize("kernel", GEMM_KERNELS)
def test_gemm_matches_fp32_reference(M, N, K, kernel):
    L = cg.lib()
    _lib.check(L.cgpt_set_option(b"gemm_kernel", kernel))
    try:
        got, ref = run_gemm(M, N, K, seed=M + N + K)
    finally:
        _lib.check(L.cgpt_set_option(b"gemm_kernel", 0))
    report(f"gemm {M}x{N}x{K}", got, ref, 1e-3 + 1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("kernel", [k for k in GEMM_KERNELS if k])
def test_gemm_identity_asymmetric(kernel):
    L = cg.lib()
    _lib.check(L.cgpt_set_option(b"gemm_kernel", kernel))
    try:
        got, ref = run_gemm(256, 256, 128, bias=False, asym=True)
    finally:
        _lib.check(L.cgpt_set_option(b"gemm_kernel", 0))
    report("gemm A=I asym W", got, ref, 1e-2)


def test_gemm_large_shape_property():
    # BASELINE shape fc2: M = 100*257 rows, K = 6144, N = 1408 -- checked on a strided sample of rows
    L = cg.lib()
    M, N, K = 25700, 1408, 6144
    g = torch.Generator(device=DEV).manual_seed(1)
    A = (torch.randn(ru(M, 256), K, device=DEV, generator=g) * 0.3).half()
    W = torch.zeros(ru(N, 256), K, device=DEV, dtype=torch.float16)
    W[:N] = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
    Cd = torch.empty(M, N, device=DEV)
    _lib.check(L.cgpt_gemm_f16(P(A), K, P(W), K, None, P(Cd), N, M, N, K, stream()))
    rows = torch.arange(0, M, 97, device=DEV)
    ref = A[rows].float() @ W[:N].float().t()
    report("gemm fc2-shape", Cd[rows].cpu(), ref.cpu(), 5e-3)
    # linearity: the same GEMM on 2*A gives exactly 2*C (power-of-two scaling is exact in fp16/fp32)
    C2 = torch.empty(M, N, device=DEV)
    A2 = (A * 2).contiguous()
    _lib.check(L.cgpt_gemm_f16(P(A2), K, P(W), K, None, P(C2), N, M, N, K, stream()))
    assert torch.equal(C2, Cd * 2)


def attn_ref(q, k, v, heads, hd, scale):
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    qh = q.float().view(B, Tq, heads, hd).permute(0, 2, 1, 3)
    kh = k.float().view(B, Tk, heads, hd).permute(0, 2, 1, 3)
    vh = v.float().view(B, Tk, heads, hd).permute(0, 2, 1, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, Tq, heads * hd)


@pytest.mark.parametrize("B,heads,hd,Tq,Tk", [(3, 2, 88, 257, 257), (2, 16, 88, 257, 257), (2, 2, 88, 17, 17),
                                               (3, 12, 64, 32, 257), (2, 12, 64, 32, 32), (2, 2, 64, 8, 17),
                                               (1, 2, 64, 8, 8), (1, 1, 88, 1, 1),
                                               # B >= 64: the XCD-aware walk (whole samples per XCD, a sample's heads on consecutive
                                               # workgroups of one XCD); 67 and 130 samples leave the XCDs unequal shares, 130 x 16 and
                                               # 70 x 12 items are several per workgroup
                                               (67, 16, 88, 257, 257), (130, 16, 88, 40, 40), (70, 12, 64, 32, 257), (64, 3, 64, 8, 17),
                                               (66, 4, 88, 300, 300), (65, 2, 64, 20, 500),   # ... and the streaming kernel's walk by samples
                                               # Tk > 288: K/V streamed through LDS in 192-key chunks (448^2 images, T = 1025)
                                               (2, 2, 88, 401, 401), (1, 16, 88, 1025, 1025), (2, 12, 64, 32, 1025),
                                               (1, 2, 88, 130, 300), (1, 2, 64, 289, 289), (1, 1, 88, 1, 577),
                                               # ... more work items than workgroups (requests and Q of the NEXT item issued inside the
                                               # current one), chunk / unit boundaries (Tk = 2 x 192 + 1, 2 x 192), three query blocks
                                               (24, 16, 88, 577, 577), (40, 12, 64, 300, 400), (3, 5, 88, 257, 385), (2, 3, 88, 700, 384)])
def test_attention_matches_fp32_reference(B, heads, hd, Tq, Tk):
    L = cg.lib()
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Tq + Tk)
    D = heads * hd
    ldq = ldkv = ru(D, 8)
    q = torch.randn(B, Tq, ldq, generator=g).half()
    k = torch.randn(B, Tk, ldkv, generator=g).half()
    v = torch.randn(B, Tk, ldkv, generator=g).half()
    # one spiky query/key pair: softmax must stay finite with a dominant logit
    q[0, 0, :hd] *= 6
    k[0, Tk - 1, :hd] = q[0, 0, :hd] * 0.5
    scale = hd ** -0.5
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    od = torch.full((B, Tq, ldq), float("nan"), device=DEV, dtype=torch.float16)
    _lib.check(L.cgpt_attention_f16(P(qd), ldq, P(kd), P(vd), ldkv, P(od), ldq, B, heads, hd, Tq, Tk, scale, stream()))
    torch.cuda.synchronize()
    ref = attn_ref(q[..., :D], k[..., :D], v[..., :D], heads, hd, scale)
    got = od.cpu().float()[..., :D]
    report(f"attention B{B} h{heads} d{hd} {Tq}x{Tk}", got.reshape(B * Tq, D), ref.reshape(B * Tq, D), 6e-3)


@pytest.mark.parametrize("hd", [88, 64])
def test_streaming_attention_rising_scores_rescale_path(hd):
    """Streaming kernel (Tk > 288): the exponent reference lags the running maximum and is moved when a query's scores outgrow it by
    2^8.  Here the scores of every query RISE along the keys (k_j = j / Tk * c * q-direction + noise): the reference moves many times,
    in the middle of chunks as well as at their first unit, with the next unit's scores already computed against the old reference
    (head_dim 88) -- and, for head_dim 64, through the unfolded form with its VALU row sums."""
    L = cg.lib()
    B, heads, Tq, Tk = 2, 3, 300, 700
    g = torch.Generator(device="cpu").manual_seed(hd)
    D = heads * hd
    q = torch.randn(B, Tq, D, generator=g)
    direction = torch.nn.functional.normalize(torch.randn(B, 1, heads, hd, generator=g), dim=-1)
    q = (q.view(B, Tq, heads, hd) * 0.2 + direction * 6.0).reshape(B, Tq, D).half()
    ramp = torch.linspace(0.0, 1.0, Tk).view(1, Tk, 1, 1)
    k = (direction * ramp * 160.0 + torch.randn(B, Tk, heads, hd, generator=g) * 0.5).reshape(B, Tk, D).half()     # logits 0 .. ~100
    v = torch.randn(B, Tk, D, generator=g).half()
    scale = hd ** -0.5
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    od = torch.full((B, Tq, D), float("nan"), device=DEV, dtype=torch.float16)
    _lib.check(L.cgpt_attention_f16(P(qd), D, P(kd), P(vd), D, P(od), D, B, heads, hd, Tq, Tk, scale, stream()))
    torch.cuda.synchronize()
    ref = attn_ref(q, k, v, heads, hd, scale)
    logits = (q.float().view(B, Tq, heads, hd)[:, 0] * k.float().view(B, Tk, heads, hd)[:, -1]).sum(-1) * scale
    assert float(logits.min()) > 40.0                                     # the ramp really spans many rescale thresholds
    report(f"streaming attention, rising scores, d{hd}", od.cpu().float().reshape(B * Tq, D), ref.reshape(B * Tq, D), 6e-3)


@pytest.mark.parametrize("hd,Tq,Tk,B,heads", [(88, 1025, 1025, 2, 3), (88, 257, 577, 3, 2), (64, 513, 1025, 2, 2), (88, 1, 1025, 9, 2),
                                              (64, 257, 289, 1, 1), (88, 1025, 353, 1, 2),
                                              # more work items than workgroups: a workgroup meets carrying and plain blocks in turn (the
                                              # rotation), rewrites the lone query's LDS state and Q fragment item after item -- 384 items of
                                              # four blocks per pair; 320 items that ALL carry (one block per pair); 66 samples: the walk by samples
                                              (88, 1025, 1025, 24, 4), (88, 257, 577, 40, 8), (64, 513, 400, 66, 3)])
def test_streaming_attention_lone_query_is_split_by_keys(hd, Tq, Tk, B, heads):
    """Tq = 256 k + 1 (the CLS token on a 16 n x 16 n patch grid; T = 1025 at the reference's image size, minigpt4.py:32): the last query
    block of the streaming kernel holds one query, whose keys are split over the eight waves by 32-key unit and merged through LDS.
    The LAST query is given a dominant key at positions that land in every wave's units, in the half-filled last unit and on both
    sides of chunk borders (192 keys), and -- second input -- scores that rise along the keys, so that the partials of the waves carry
    maxima dozens of powers of two apart; rows 0 .. Tq - 2 go the ordinary way and are checked with it."""
    L = cg.lib()
    D = heads * hd
    scale = hd ** -0.5
    g = torch.Generator(device="cpu").manual_seed(hd + Tq + Tk)
    spikes = sorted({0, 31, 32, 95, 191, 192, 223, 255, 256, 287, Tk // 2, Tk - 34, Tk - 33, Tk - 2, Tk - 1})
    for case in ("spikes", "ramp"):
        q = torch.randn(B, Tq, D, generator=g).half()
        k = torch.randn(B, Tk, D, generator=g).half()
        v = torch.randn(B, Tk, D, generator=g).half()
        if case == "spikes":
            # sample b, head h: the lone query points at key spikes[(b * heads + h) % len]
            for b in range(B):
                for h in range(heads):
                    j = spikes[(b * heads + h) % len(spikes)]
                    q[b, Tq - 1, h * hd:(h + 1) * hd] *= 5
                    k[b, j, h * hd:(h + 1) * hd] = q[b, Tq - 1, h * hd:(h + 1) * hd] * 0.6
        else:
            direction = torch.nn.functional.normalize(torch.randn(B, 1, heads, hd, generator=g), dim=-1)
            q = (q.float().view(B, Tq, heads, hd) * 0.2 + direction * 6.0).reshape(B, Tq, D).half()
            ramp = torch.linspace(0.0, 1.0, Tk).view(1, Tk, 1, 1)
            k = (direction * ramp * 160.0 + torch.randn(B, Tk, heads, hd, generator=g) * 0.5).reshape(B, Tk, D).half()
        qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
        od = torch.full((B, Tq, D), float("nan"), device=DEV, dtype=torch.float16)
        _lib.check(L.cgpt_attention_f16(P(qd), D, P(kd), P(vd), D, P(od), D, B, heads, hd, Tq, Tk, scale, stream()))
        # a second launch must give the same BITS: the partial softmaxes meet in LDS behind barriers, a race would show as a difference
        od2 = torch.full((B, Tq, D), float("nan"), device=DEV, dtype=torch.float16)
        _lib.check(L.cgpt_attention_f16(P(qd), D, P(kd), P(vd), D, P(od2), D, B, heads, hd, Tq, Tk, scale, stream()))
        torch.cuda.synchronize()
        assert torch.equal(od, od2), f"two launches differ ({case}, d{hd} {Tq}x{Tk}, B{B} h{heads})"
        ref = attn_ref(q, k, v, heads, hd, scale)
        got = od.cpu().float()
        report(f"streaming attention, lone query ({case}) d{hd} {Tq}x{Tk}: the lone row", got[:, Tq - 1], ref[:, Tq - 1], 6e-3)
        report(f"streaming attention, lone query ({case}) d{hd} {Tq}x{Tk}: all rows", got.reshape(B * Tq, D), ref.reshape(B * Tq, D), 6e-3)


@pytest.mark.parametrize("rows,D", [(5, 176), (7, 128), (33, 768), (257, 1408), (3, 4096)])
def test_layernorm_matches_torch(rows, D):
    L = cg.lib()
    g = torch.Generator(device="cpu").manual_seed(D)
    x = torch.randn(rows, D, generator=g) * 3 + 1
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    ld16 = ru(D, 64)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y16 = torch.zeros(rows, ld16, dtype=torch.float16, device=DEV)
    y32 = torch.zeros(rows, D, device=DEV)
    for eps in (1e-6, 1e-12):
        _lib.check(L.cgpt_layernorm(P(xd), D, P(wd), P(bd), eps, P(y16), ld16, P(y32), D, rows, D, stream()))
        ref = torch.nn.functional.layer_norm(x, (D,), w, b, eps)
        report(f"layernorm f32 {rows}x{D}", y32.cpu(), ref, 2e-5 * max(1.0, float(ref.abs().max())))
        report(f"layernorm f16 {rows}x{D}", y16.cpu().float()[:, :D], ref, 2e-3 * max(1.0, float(ref.abs().max())))
    assert float(y16[:, D:].abs().max()) == 0.0 if ld16 > D else True


def test_noise_stream_matches_oracle_and_is_shard_independent():
    x = torch.randn(3, 56, 56)
    xd = x.to(DEV)
    full = cg.noise_batch(xd, 5, 6, 0.5, 42)
    torch.cuda.synchronize()
    ref = x.numpy()[None] + np.float32(0.5) * philox.noise_batch(42, 5, 6, (3, 56, 56))
    assert np.abs(full.cpu().numpy() - ref).max() < 5e-6          # fp32 vs float64 transcendental rounding only
    part = cg.noise_batch(xd, 7, 3, 0.5, 42)                      # samples 7,8,9 == rows 2,3,4 of `full`
    assert torch.equal(part, full[2:5])                           # bit-identical for any batching / sharding
    z = (cg.noise_batch(torch.zeros(3, 224, 224, device=DEV), 0, 8, 1.0, 7)).flatten()
    assert abs(float(z.mean())) < 3e-3 and abs(float(z.std()) - 1) < 3e-3
    assert abs(float((z ** 3).mean())) < 1e-2 and abs(float((z ** 4).mean()) - 3) < 3e-2
    z2 = cg.noise_batch(torch.zeros(3, 224, 224, device=DEV), 0, 8, 1.0, 8).flatten()
    assert abs(float((z * z2).mean())) < 3e-3                     # different seeds are uncorrelated


def test_vote_first_max_and_accumulates():
    g = torch.Generator(device="cpu").manual_seed(3)
    for K in (2, 10, 63, 64, 65, 1000):
        logits = torch.randn(333, K, generator=g)
        logits[5] = 1.0                                           # full tie -> class 0
        logits[6, K - 1] = logits[6, 0] = 9.0                     # two maxima -> the first one
        ld = logits.to(DEV)
        counts = torch.zeros(K, dtype=torch.int64, device=DEV)
        cg.vote(ld, counts)
        cg.vote(ld, counts)                                       # accumulates
        ref = 2 * torch.bincount(torch.argmax(logits, dim=1), minlength=K)
        assert torch.equal(counts.cpu(), ref), (K, counts.cpu().tolist()[:10], ref.tolist()[:10])


def test_vote_nan_ranks_as_maximum_like_argmax():
    """`base_classifier(...).argmax(1)` (smoothing.py:97) on a row holding NaNs returns the index of the FIRST NaN
    (torch / numpy argmax treat NaN as the maximum); the wavefront vote must vote the same class."""
    K = 130
    logits = torch.randn(8, K)
    logits[0, 70] = float("nan")                                  # a NaN in the second 64-class stripe
    logits[1, 5] = float("inf"); logits[1, 99] = float("nan")     # NaN beats +inf
    logits[2, 129] = float("nan"); logits[2, 3] = float("nan")    # several NaNs: the first one
    logits[3, :] = float("nan")                                   # all NaN -> class 0
    logits[4, :] = float("-inf")                                  # all -inf -> class 0
    logits[5, 64] = float("inf"); logits[5, 65] = float("inf")    # tie of infinities: the first
    ref = torch.bincount(torch.argmax(logits, dim=1), minlength=K)
    assert torch.argmax(logits, dim=1)[:4].tolist() == [70, 99, 3, 0]
    counts = torch.zeros(K, dtype=torch.int64, device=DEV)
    cg.vote(logits.to(DEV), counts)
    assert torch.equal(counts.cpu(), ref), (counts.cpu().nonzero().flatten().tolist(), ref.nonzero().flatten().tolist())


@pytest.mark.parametrize("M,N,K", [(300, 384, 192), (1500, 640, 128), (2313, 1408, 1408), (1028, 4224, 256)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_linear_split_last_columns_is_bit_identical(M, N, K, epi):
    """N = 256 k + 128 (ViT-G: 1408, 4224): the GEMM covers the last 384 columns with two 192-column tiles instead of one full
    and one half-empty 256-column tile.  The accumulation order per output element is unchanged, so every epilogue must give
    the same bits as the unsplit tiling (gemm_ablate bit 16384) -- and untouched memory outside [M, N] stays untouched."""
    L = cg.lib()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N + epi)
    A = torch.zeros(ru(M, 256), K, dtype=torch.float16); A[:M] = (torch.randn(M, K, generator=g) * 0.5).half()
    W = torch.zeros(ru(N, 256), K, dtype=torch.float16); W[:N] = (torch.randn(N, K, generator=g) * 0.1).half()
    b = torch.randn(N, generator=g)
    aux = torch.randn(M, N + 8, generator=g).to(DEV) if epi == 3 else None
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    outs = []
    try:
        for abl in (16384, 0):
            out = torch.full((M + 1, N + 8), 7.0, device=DEV, dtype=torch.float16 if epi < 2 else torch.float32)
            _lib.check(L.cgpt_set_option(b"gemm_ablate", abl))
            _lib.check(L.cgpt_linear_f16(P(Ad), K, P(Wd), K, P(bd), P(out), N + 8, P(aux) if aux is not None else None, N + 8, M, N, K, epi, stream()))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        _lib.check(L.cgpt_set_option(b"gemm_ablate", 0))
    assert torch.equal(outs[0], outs[1])
    assert bool((outs[1][:, N:] == 7.0).all()) and bool((outs[1][M] == 7.0).all())
    ref = A[:M].float() @ W[:N].float().t() + b
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 3:
        ref = ref + aux[:, :N].cpu()
    report(f"linear split {M}x{N}x{K} epi{epi}", outs[1][:M, :N].float().cpu(), ref, 2e-3 + 1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("epi", [0, 1, 3])
def test_gemm_grid_option_is_bit_identical(epi):
    """cgpt_set_option("gemm_grid", n): the persistent 256 x 256 kernel on fewer workgroups than CUs (CUs left to another stream's
    kernels, tools/two_stream_probe.py).  A tile's arithmetic does not depend on the workgroup that computes it: same bits."""
    L = cg.lib()
    M, N, K = 4100, 4224, 256                                  # 17 x 17 tiles: more than 128, the two-phase kernel is picked
    g = torch.Generator(device="cpu").manual_seed(91 + epi)
    A = torch.zeros(ru(M, 256), K, dtype=torch.float16); A[:M] = (torch.randn(M, K, generator=g) * 0.5).half()
    W = torch.zeros(ru(N, 256), K, dtype=torch.float16); W[:N] = (torch.randn(N, K, generator=g) * 0.1).half()
    b = torch.randn(N, generator=g)
    aux = torch.randn(M, N, generator=g).to(DEV) if epi == 3 else None
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    outs = []
    try:
        for grid in (0, 224, 64, 8):
            out = torch.full((M, N), 7.0, device=DEV, dtype=torch.float16 if epi < 2 else torch.float32)
            _lib.check(L.cgpt_set_option(b"gemm_grid", grid))
            _lib.check(L.cgpt_linear_f16(P(Ad), K, P(Wd), K, P(bd), P(out), N, P(aux) if aux is not None else None, N, M, N, K, epi, stream()))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        _lib.check(L.cgpt_set_option(b"gemm_grid", 0))
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    ref = A[:M].float() @ W[:N].float().t() + b
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 3:
        ref = ref + aux.cpu()
    report(f"linear gemm_grid {M}x{N}x{K} epi{epi}", outs[0].float().cpu(), ref, 2e-3 + 1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("epi", [0, 1])
def test_fp16_epilogue_value_does_not_depend_on_kernel_or_tile_path(epi):
    """The same rows through (a) a small launch (M = 300: another kernel, guarded per-element epilogue with the scalar GELU) and
    (b) a large one (M = 1300: 256x256 tiles, full tiles with the packed-fp32 GELU, the last tile row partial) must agree bit for
    bit (a fused fma+convert in one path and fma, then convert in the other differ by one fp16 ulp in 3 of 100 000 values): the counts of Smooth.certify may not depend on how the samples were cut into batches (smoothing.py:91-98)."""
    L = cg.lib()
    N, K = 640, 256
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.zeros(ru(1300, 256), K, dtype=torch.float16); A[:1300] = (torch.randn(1300, K, generator=g) * 0.7).half()
    W = torch.zeros(ru(N, 256), K, dtype=torch.float16); W[:N] = (torch.randn(N, K, generator=g) * 0.15).half()
    b = torch.randn(N, generator=g)
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    outs = []
    for M in (300, 1300):
        out = torch.zeros(M, N, device=DEV, dtype=torch.float16)
        _lib.check(L.cgpt_linear_f16(P(Ad), K, P(Wd), K, P(bd), P(out), N, None, N, M, N, K, epi, stream()))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1][:300])
    try:                                                     # ... and every forced kernel on the small launch
        for kernel in (1, 3, 4, 14):
            _lib.check(L.cgpt_set_option(b"gemm_kernel", kernel))
            out = torch.zeros(300, N, device=DEV, dtype=torch.float16)
            _lib.check(L.cgpt_linear_f16(P(Ad), K, P(Wd), K, P(bd), P(out), N, None, N, 300, N, K, epi, stream()))
            torch.cuda.synchronize()
            assert torch.equal(out, outs[0]), kernel
    finally:
        _lib.check(L.cgpt_set_option(b"gemm_kernel", 0))
    # the partial last tile row of the large launch (guarded path) against the same rows inside full tiles of a third launch
    out3 = torch.zeros(1536, N, device=DEV, dtype=torch.float16)
    _lib.check(L.cgpt_linear_f16(P(Ad), K, P(Wd), K, P(bd), P(out3), N, None, N, 1536, N, K, epi, stream()))
    torch.cuda.synchronize()
    assert torch.equal(outs[1], out3[:1300])


@pytest.mark.parametrize("M,N,K,epi", [(25700, 1408, 6144, 0), (5140, 6144, 1408, 1), (2313, 1408, 3072, 3), (1300, 640, 64, 0), (1029, 768, 128, 2)])
def test_two_phase_quadrant_kernel_is_bit_identical_to_the_phased_kernel(M, N, K, epi):
    """gemm_kernel 14 (gemm9.hip: quadrant parts requested 1.5 K-tiles ahead behind counted vmcnt waits, the automatic choice for
    K >= 3072) against gemm_kernel 4 on the same operands, several launches each with competing traffic on a second stream: a
    fragment read that overtakes its LDS-DMA request shows up as rare wrong tiles, not as a failure of a single clean run."""
    L = cg.lib()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    A = torch.zeros(ru(M, 256), K, device=DEV, dtype=torch.float16); A[:M] = (torch.randn(M, K, device=DEV, generator=g) * 0.5).half()
    W = torch.zeros(ru(N, 256), K, device=DEV, dtype=torch.float16); W[:N] = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
    b = torch.randn(N, device=DEV, generator=g)
    aux = torch.randn(M, N, device=DEV, generator=g) if epi == 3 else None
    dt = torch.float16 if epi < 2 else torch.float32
    side = torch.cuda.Stream()
    ja = torch.randn(4096, 4096, device=DEV, dtype=torch.float16)

    def run(kernel):
        out = torch.full((M, N), 3.0, device=DEV, dtype=dt)
        _lib.check(L.cgpt_set_option(b"gemm_kernel", kernel))
        _lib.check(L.cgpt_linear_f16(P(A), K, P(W), K, P(b), P(out), N, P(aux) if aux is not None else None, N, M, N, K, epi, stream()))
        return out
    try:
        ref = run(4)
        for it in range(8):
            if it & 1:
                with torch.cuda.stream(side):
                    ja @ ja
            assert torch.equal(run(14), ref), it
    finally:
        _lib.check(L.cgpt_set_option(b"gemm_kernel", 0))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(513, 512, 128), (1300, 6144, 1408), (300, 384, 192)])
def test_linear_gelu_epilogue_matches_exact_erf_gelu(M, N, K):
    """cgpt_linear_f16 with the fused GELU epilogue (Mlp.fc1 + nn.GELU, eva_vit.py:59-61) against fp32 torch: exact-erf GELU of
    the fp32 linear output, compared at fp16 output resolution."""
    L = cg.lib()
    g = torch.Generator(device="cpu").manual_seed(M + N)
    A = torch.zeros(ru(M, 256), K, dtype=torch.float16); A[:M] = (torch.randn(M, K, generator=g) * 0.6).half()
    W = torch.zeros(ru(N, 256), K, dtype=torch.float16); W[:N] = (torch.randn(N, K, generator=g) * 0.1).half()
    b = torch.randn(N, generator=g)
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float16)
    _lib.check(L.cgpt_linear_f16(P(Ad), K, P(Wd), K, P(bd), P(out), N, None, N, M, N, K, 1, stream()))
    torch.cuda.synchronize()
    lin = A[:M].float() @ W[:N].float().t() + b
    ref = torch.nn.functional.gelu(lin)                       # erf form
    got = out.cpu().float()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    tol = 1e-3 * ref.abs() + 2e-4                             # fp16 rounding (2^-11 relative) + accumulation noise
    assert bool((err <= tol).all()), (float(err.max()), float((err / tol).max()))
    # the negative tail, where gelu(x) is tiny (but still a normal fp16 number): resolved to fp16 precision
    tail = (lin < -3.0) & (lin > -4.2)
    if tail.any():
        assert float((err[tail] / ref[tail].abs().clamp_min(1e-7)).max()) <= 3e-2


@pytest.mark.parametrize("M,N,K", [(10317, 6144, 1408), (5000, 768, 1216), (3000, 256, 6144)])
def test_gelu_epilogue_gives_the_same_bits_on_every_kernel_under_load(M, N, K):
    """EPI_F16_GELU (Mlp.fc1 + nn.GELU, eva_vit.py:59-61): every kernel (128x128, phased 256x256, two-phase quadrant with the zero
    accumulator operand in a tile's first K-tile) gives the same bits, over several launches with competing traffic on a second stream,
    incl. 19 / 22 / 96 K-tiles and a ragged last tile row; and they equal exact-erf GELU of the fp32 linear output to fp16 resolution."""
    L = cg.lib()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    A = torch.zeros(ru(M, 256), K, device=DEV, dtype=torch.float16); A[:M] = (torch.randn(M, K, device=DEV, generator=g) * 0.6).half()
    W = torch.zeros(ru(N, 256), K, device=DEV, dtype=torch.float16); W[:N] = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
    b = torch.randn(N, device=DEV, generator=g)
    side = torch.cuda.Stream()
    ja = torch.randn(4096, 4096, device=DEV, dtype=torch.float16)

    def run(kernel, epi=1):
        out = torch.full((M + 3, N), 5.0, device=DEV, dtype=torch.float16)          # 3 guard rows behind the matrix
        _lib.check(L.cgpt_set_option(b"gemm_kernel", kernel))
        _lib.check(L.cgpt_linear_f16(P(A), K, P(W), K, P(b), P(out), N, None, N, M, N, K, epi, stream()))
        return out
    try:
        ref = run(4)
        assert torch.equal(run(1), ref)
        for it in range(4):
            if it & 1:
                with torch.cuda.stream(side):
                    ja @ ja
            got = run(14)
            assert torch.equal(got, ref), (it, int((got != ref).sum()))
    finally:
        _lib.check(L.cgpt_set_option(b"gemm_kernel", 0))
    torch.cuda.synchronize()
    assert bool((ref[M:] == 5.0).all())
    rows = torch.arange(0, M, max(1, M // 1024), device=DEV)
    want = torch.nn.functional.gelu(A[rows].float() @ W[:N].float().t() + b)
    err = (ref[rows].float() - want).abs()
    tol = 1e-3 * want.abs() + 2e-4            # fp16 rounding (2^-11 relative) + accumulation-order noise of the fp32 linear
    assert bool((err <= tol).all()), float((err / tol).max())


def test_mfma_sustained_diagnostic_reports_a_plausible_rate():
    """cgpt_mfma_sustained (the yardstick bench.py prints beside the data-sheet peak): an MFMA-only loop on random operands.  The rate
    must lie between what a badly throttled device would give and the data sheet's 2.5 PFLOP/s, and equal 1 024 SIMDs x 16 384 FLOP /
    ~16-17 cycles at the clock it reports (the matrix pipe never idles in that loop)."""
    import ctypes as C
    L = cg.lib()
    tf, ghz = C.c_double(), C.c_double()
    _lib.check(L.cgpt_mfma_sustained(0.5, C.byref(tf), C.byref(ghz)))
    assert 900.0 < tf.value < 2600.0, tf.value
    assert 1.0 < ghz.value < 2.6, ghz.value
    cycles_per_mfma = ghz.value * 1e9 * 1024 * 16384 / (tf.value * 1e12)
    assert 15.5 < cycles_per_mfma < 19.0, (cycles_per_mfma, tf.value, ghz.value)
    assert L.cgpt_mfma_sustained(0.0, C.byref(tf), C.byref(ghz)) != 0
