"""Headline-configuration parity, run once per round on the GPU box (too slow for the test suite: 200 fp32 ViT-G forwards on the host):
ONE whole `Smooth.certify(x, n0=100, n=100, alpha=0.001, batch_size=100)` at sigma = 0.5 on the GPU (HIP path) and on the CPU oracle
(oracle/smooth_oracle.py around oracle/model_oracle.py) with the same weights, image and the GPU's own noise draws exported.
Prints the two results, the per-sample argmax agreement and the radius difference.   python tools/parity_headline.py [num_images]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import certifiedgpt_amd as cg
from oracle import model_oracle as mo, smooth_oracle as so

K, SIGMA, N0, N, ALPHA, SEED = 1000, 0.5, 100, 100, 0.001, 42
torch.set_num_threads(int(os.environ.get("CGPT_CPU_THREADS", "16")))
cfg = mo.Config(mode=mo.MODE_VIT_HEAD, num_classes=K)
clf = cg.HipClassifier(mode="vit_head", num_classes=K, max_batch=100).init_synthetic(0)
params = {n: torch.from_numpy(clf.get_weight(n)).reshape(s) for n, s in mo.param_shapes(cfg).items()}
for img in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    x = torch.from_numpy(mo.synthetic_image(cfg, seed=1234 + img)).to("cuda:0")
    s = cg.Smooth(clf, K, SIGMA, seed=SEED)
    gpu = s.certify(x, N0, N, ALPHA, 100)
    g_logits = torch.cat([clf.forward_logits(x, 0, 100, SIGMA, SEED), clf.forward_logits(x, 100, 100, SIGMA, SEED)]).cpu()
    draws = cg.noise_batch(torch.zeros_like(x), 0, N0 + N, 1.0, SEED).cpu().numpy()
    ref = []
    t0 = time.perf_counter()
    def classifier(batch):
        out = []
        for i in range(0, len(batch), 10):
            out.append(mo.forward_all(params, torch.from_numpy(np.ascontiguousarray(batch[i:i + 10])), cfg)["logits"])
            print(f"  cpu oracle: {sum(len(o) for o in ref) + sum(len(o) for o in out)}/{N0 + N} forwards, {time.perf_counter() - t0:.0f} s", flush=True)
        out = torch.cat(out); ref.append(out)
        return out.numpy()
    oracle = so.SmoothOracle(classifier, K, SIGMA, lambda first, num, shape: draws[first:first + num])
    cpu = oracle.certify(x.cpu().numpy(), N0, N, ALPHA, 100)
    r = torch.cat(ref)
    agree = int((r.argmax(1) == g_logits.argmax(1)).sum())
    top2 = r.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]) / r.abs().max()
    flips = (r.argmax(1) != g_logits.argmax(1))
    print(f"image {img}: GPU {gpu}  CPU oracle {cpu}  label equal {gpu[0] == cpu[0]}  |dR| {abs(gpu[1] - cpu[1]):.3e}  "
          f"argmax agreement {agree}/{N0 + N}  logits rel err {float((g_logits - r).abs().max() / r.abs().max()):.2e}  "
          f"fp32 top-2 margins of the disagreeing samples (rel. to max|logit|): {[round(float(m), 5) for m in margin[flips]]}  "
          f"CPU time {time.perf_counter() - t0:.0f} s", flush=True)
