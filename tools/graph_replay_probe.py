"""Does replaying the forward from a hipGraph shrink the ~6-us kernel boundaries (profiles/r04/kernel_gaps.txt)?  Captures one
HipClassifier.sample_counts_images call (51 images x 200 draws = 40 batches of 255, ~11 000 kernel nodes) with torch.cuda.graph and
times replay against the eager call, interleaved.  Measurement only (the sample indices are baked into the captured kernels' arguments).
Run on the GPU box:  python tools/graph_replay_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import certifiedgpt_amd as cg

G = int(os.environ.get("CGPT_PROBE_IMAGES", "51"))
clf = cg.HipClassifier(mode="vit_head", num_classes=1000, max_batch=255, device=0)
clf.init_synthetic(seed=0)
xs = torch.randn(G, 3, 224, 224, device="cuda:0")
eager = lambda: clf.sample_counts_images(xs, 0, 100, 100, 100, 200, 0.5, 42)
ref = eager(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eager()
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, stream=side):
        out = eager()
except Exception as e:
    print("capture failed:", repr(e)); sys.exit(0)
g.replay(); torch.cuda.synchronize()
print("replay counts equal eager:", bool(torch.equal(out, ref)))
def t(fn, reps=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps / G
for r in range(3):
    print(f"round {r}: eager {t(eager):.3f} ms per image | graph replay {t(g.replay):.3f} ms per image")
