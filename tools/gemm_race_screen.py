"""Race screen of the two-phase quadrant GEMM (gemm_kernel 14, the library's default large-shape kernel) against the phased kernel
(gemm_kernel 4): bitwise equality of the outputs over many launches -- fixed ViT / Q-Former shapes plus random shapes (ragged M, N
not a multiple of 256 incl. the 192-column split, K from one K-tile up), all four epilogues, with competing traffic on a second
stream every third launch (uneven timing).  A fragment read that overtakes its LDS-DMA request, or a request that overtakes a read,
shows up as rare wrong tiles that a single clean run does not reveal.  CGPT_SCREEN_KERNEL=16 screens another kernel (default 0 = the
automatic choice).  Run on the GPU box:  python tools/gemm_race_screen.py [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, random, torch
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
L = cg.lib(); DEV = "cuda:0"
def P(t): return C.c_void_p(t.data_ptr()) if t is not None else None
def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
LAUNCHES = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SCREEN = int(os.environ.get("CGPT_SCREEN_KERNEL", "0"))
side = torch.cuda.Stream()
ja = torch.randn(6144, 6144, device=DEV, dtype=torch.float16)
shapes = [(65535, 1408, 6144, 0), (65535, 6144, 1408, 1), (40000, 3072, 1152, 1), (65535, 4224, 1408, 0), (65535, 1408, 1408, 0), (65535, 9216, 1408, 0),
          (8160, 4096, 768, 2), (6425, 1408, 6144, 0), (3341, 4224, 1408, 0), (65535, 1000, 128, 2), (65535, 1408, 64, 0), (2570, 768, 3072, 3)]
rnd = random.Random(20260104)
for _ in range(30):
    M = rnd.choice([1024, 1285, 2049, 3341, 5140, 8224, 12850, 25700, 33000])
    N = rnd.choice([384, 640, 768, 1000, 1408, 1536, 2304, 3072, 4224, 6144])
    K = 64 * rnd.choice([1, 2, 3, 4, 6, 11, 12, 22, 24, 48])
    shapes.append((M, N, K, rnd.randrange(4)))
bad = 0
for (M, N, K, epi) in shapes:
    Mp = (M + 255) // 256 * 256
    g = torch.Generator(device=DEV).manual_seed(M * 31 + N * 7 + K)
    A = (torch.randn(Mp, K, device=DEV, generator=g) * 0.5).half()
    W = torch.zeros((N + 255) // 256 * 256, K, device=DEV, dtype=torch.float16); W[:N] = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
    bias = torch.randn(N, device=DEV, generator=g)
    ld = (N + 7) // 8 * 8
    dt = torch.float16 if epi < 2 else torch.float32
    aux = torch.randn(M, ld, device=DEV, generator=g) if epi == 3 else None
    def run(kern):
        out = torch.full((M, ld), 3.0, device=DEV, dtype=dt)
        _lib.check(L.cgpt_set_option(b"gemm_kernel", kern))
        _lib.check(L.cgpt_linear_f16(P(A), K, P(W), K, P(bias), P(out), ld, P(aux), ld, M, N, K, epi, st()))
        return out
    ref = run(4); torch.cuda.synchronize()
    nbad = 0
    for it in range(LAUNCHES):
        if it % 3 == 1:
            with torch.cuda.stream(side):
                ja @ ja
        if not torch.equal(run(SCREEN), ref):
            nbad += 1
    torch.cuda.synchronize()
    print(f"{M}x{N}x{K} epi{epi}: {LAUNCHES - nbad}/{LAUNCHES} launches bit-identical to the phased kernel", flush=True)
    bad += nbad
_lib.check(L.cgpt_set_option(b"gemm_kernel", 0))
print(f"RACE SCREEN (gemm_kernel {SCREEN}) {'CLEAN' if bad == 0 else 'FAILED'}: {len(shapes)} shapes x {LAUNCHES} launches, {bad} mismatching launches")
