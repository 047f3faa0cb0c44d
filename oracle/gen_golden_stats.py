#!/opt/conda/bin/python3.9
"""Generate tests/golden/stats_golden.json by executing the REFERENCE's own
randomized_smoothing/smoothing.py (Smooth.certify / Smooth.predict /
_count_arr / _lower_confidence_bound, smoothing.py:29-117) with its own-era
scipy 1.7.1 / statsmodels 0.12.2.

Runs ONLY in the build container (needs /root/reference and
/opt/conda/bin/python3.9); neither this harness's imports nor any reference
file travels to the GPU box -- only the JSON does.

    /opt/conda/bin/python3.9 -W ignore oracle/gen_golden_stats.py

`torch` is not installed for py3.9; smoothing.py only touches it in type
annotations (smoothing.py:19,29,58,81), so a 2-attribute stand-in module is
enough.  `_sample_noise` (the classifier loop, smoothing.py:81-99) is replaced
by a function that returns prescribed count vectors, so every other line of
the class executes unmodified.
"""
import json
import os
import sys
import types
import itertools

import numpy as np

stub = types.ModuleType("torch")
stub.nn = types.SimpleNamespace(Module=object)
stub.tensor = object
sys.modules["torch"] = stub
sys.path.insert(0, "/root/reference")
from randomized_smoothing.smoothing import Smooth  # noqa: E402
import scipy  # noqa: E402
import statsmodels  # noqa: E402
from scipy.stats import norm, binom_test  # noqa: E402
from statsmodels.stats.proportion import proportion_confint  # noqa: E402


class _Eval:
    def eval(self):
        return self


def run_certify(counts_sel, counts_est, n, alpha, sigma):
    K = len(counts_sel)
    s = Smooth(_Eval(), K, sigma)
    seq = [np.asarray(counts_sel, dtype=int), np.asarray(counts_est, dtype=int)]
    s._sample_noise = lambda x, num, bs: seq.pop(0)
    label, radius = s.certify(None, int(sum(counts_sel)), n, alpha, 1)
    return int(label), float(radius)


def run_predict(counts, alpha):
    K = len(counts)
    s = Smooth(_Eval(), K, 1.0)
    s._sample_noise = lambda x, num, bs: np.asarray(counts, dtype=int)
    out = s.predict(None, int(sum(counts)), alpha, 1)
    return int(out), type(out).__name__


def two_class(nA, n, K=3, top=1):
    c = [0] * K
    c[top] = nA
    c[(top + 1) % K] = n - nA
    return c


def main():
    rng = np.random.RandomState(20251121)
    out = {
        "generator": "oracle/gen_golden_stats.py",
        "reference": "randomized_smoothing/smoothing.py (snapshot 2025-11-21)",
        "scipy": scipy.__version__,
        "statsmodels": statsmodels.__version__,
        "certify": [], "predict": [], "lcb": [], "binom_test": [],
        "norm_ppf": [], "count_arr": [],
    }
    # ---- certify: grid over N, nA, alpha, sigma (selection picks class 1) ----
    for n in (10, 100, 125, 1000):
        nAs = sorted(set([0, 1, n // 2 - 1, n // 2, n // 2 + 1, n - 1, n]
                         + list(range(max(0, int(0.5 * n)), n + 1, max(1, n // 25)))))
        for nA, alpha, sigma in itertools.product(nAs, (0.001, 0.05), (0.25, 0.5, 1.0)):
            est = two_class(nA, n)
            sel = two_class(n, n)  # unanimous selection of class 1
            lab, rad = run_certify(sel, est, n, alpha, sigma)
            out["certify"].append(dict(counts_sel=sel, counts_est=est, n=n, alpha=alpha,
                                       sigma=sigma, label=lab, radius=rad))
    # selection ties / argmax-first semantics, cAHat differing from estimation top
    for sel, est, n in ([[5, 5], [3, 7], 10], [[5, 5], [9, 1], 10], [[0, 4, 4, 2], [1, 90, 5, 4], 100],
                        [[1, 0, 0], [0, 100, 0], 100], [[0, 0, 10], [0, 0, 10], 10]):
        for alpha, sigma in ((0.001, 0.25), (0.05, 0.5)):
            lab, rad = run_certify(sel, est, n, alpha, sigma)
            out["certify"].append(dict(counts_sel=sel, counts_est=est, n=n, alpha=alpha,
                                       sigma=sigma, label=lab, radius=rad))
    # random multi-class histograms
    for _ in range(60):
        K = int(rng.choice([2, 5, 10, 1000]))
        n = int(rng.choice([10, 100, 125, 1000]))
        p = rng.dirichlet(np.ones(min(K, 6)) * rng.choice([0.2, 1.0]))
        pf = np.zeros(K); pf[rng.choice(K, size=len(p), replace=False)] = p
        sel = rng.multinomial(n, pf).tolist()
        est = rng.multinomial(n, pf).tolist()
        alpha = float(rng.choice([0.001, 0.01, 0.05])); sigma = float(rng.choice([0.25, 0.5, 1.0]))
        lab, rad = run_certify(sel, est, n, alpha, sigma)
        out["certify"].append(dict(counts_sel=sel, counts_est=est, n=n, alpha=alpha,
                                   sigma=sigma, label=lab, radius=rad))
    # ---- predict ----
    for n in (10, 100, 125, 1000):
        for c1 in range((n + 1) // 2, n + 1, max(1, n // 50)):
            for alpha in (0.001, 0.05):
                counts = [c1, n - c1, 0]
                lab, ty = run_predict(counts, alpha)
                out["predict"].append(dict(counts=counts, alpha=alpha, label=lab, pytype=ty))
    for counts in ([50, 50, 0], [0, 50, 50], [40, 40, 20], [33, 33, 34], [0, 0, 100], [100, 0, 0],
                   [1, 0, 0], [0, 1, 1], [3, 90, 7, 0], [10, 20, 70], [70, 20, 10], [20, 70, 10]):
        for alpha in (0.001, 0.05, 0.5):
            lab, ty = run_predict(counts, alpha)
            out["predict"].append(dict(counts=counts, alpha=alpha, label=lab, pytype=ty))
    for _ in range(60):
        K = int(rng.choice([2, 5, 10, 1000])); n = int(rng.choice([10, 100, 1000]))
        p = rng.dirichlet(np.ones(min(K, 4)) * 0.5)
        pf = np.zeros(K); pf[rng.choice(K, size=len(p), replace=False)] = p
        counts = rng.multinomial(n, pf).tolist(); alpha = float(rng.choice([0.001, 0.05]))
        lab, ty = run_predict(counts, alpha)
        out["predict"].append(dict(counts=counts, alpha=alpha, label=lab, pytype=ty))
    # ---- the third-party scalar functions at the reference call sites ----
    s = Smooth(_Eval(), 2, 1.0)
    for n in (1, 2, 10, 100, 125, 1000, 10000, 100000):
        for nA in sorted(set([0, 1, n // 3, n // 2, (2 * n) // 3, n - 1, n])):
            for alpha in (0.001, 0.01, 0.05):
                out["lcb"].append(dict(nA=nA, n=n, alpha=alpha,
                                       value=float(s._lower_confidence_bound(nA, n, alpha))))
    for n in (1, 2, 3, 10, 11, 100, 101, 125, 1000, 2000):
        for k in sorted(set([0, 1, n // 4, n // 2, (n + 1) // 2, (3 * n) // 4, n - 1, n])):
            out["binom_test"].append(dict(k=k, n=n, p=0.5, value=float(binom_test(k, n, p=0.5))))
    for k in range(50, 101):
        out["binom_test"].append(dict(k=k, n=100, p=0.5, value=float(binom_test(k, 100, p=0.5))))
    for p in (0.5, 0.5000001, 0.51, 0.6, 0.75, 0.7753298801677749, 0.9, 0.933254300796991, 0.99, 0.999,
              0.999999, 1 - 1e-12, 0.4, 0.1, 1e-6):
        out["norm_ppf"].append(dict(p=p, value=float(norm.ppf(p))))
    for arr, length in (([0, 2, 2, 1, 2], 4), ([], 3), ([9] * 7, 10)):
        out["count_arr"].append(dict(arr=arr, length=length,
                                     counts=s._count_arr(np.asarray(arr, dtype=int), length).tolist()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "stats_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print({k: len(v) for k, v in out.items() if isinstance(v, list)}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
