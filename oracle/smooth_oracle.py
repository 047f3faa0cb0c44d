"""ORACLE (test infrastructure) -- CPU restatement of `Smooth`
(reference randomized_smoothing/smoothing.py:13-117).

The reference delegates three scalar functions to third-party libraries that
are NOT pinned anywhere in the reference (docker/tpu-docker:21-43 lists neither):

  * statsmodels.stats.proportion.proportion_confint(method="beta")  (smoothing.py:117)
        published algorithm (statsmodels 0.12.2 proportion.py, method 'beta'):
            ci_low = scipy.stats.beta.ppf(alpha/2, count, nobs - count + 1); 0 when count == 0
  * scipy.stats.binom_test(x, n, p)  (smoothing.py:76) -- removed in scipy >= 1.12;
        published algorithm of scipy 1.7.1 (stats/morestats.py, two-sided branch, "from R's binom.test")
        restated in `binom_test` below.
  * scipy.stats.norm.ppf  (smoothing.py:55)

This oracle uses the scipy that is installed (1.15.x) for beta.ppf / binom pmf,cdf,sf /
norm.ppf and is pinned against goldens emitted by the reference itself under scipy 1.7.1 /
statsmodels 0.12.2 (tests/golden/stats_golden.json; tests/test_oracle_stats.py).
"""
from math import ceil

import numpy as np
from scipy.stats import beta as _beta, binom as _binom, norm as _norm

ABSTAIN = -1  # smoothing.py:17


def lower_confidence_bound(NA: int, N: int, alpha: float) -> float:
    """smoothing.py:107-117 -> proportion_confint(NA, N, alpha=2*alpha, method="beta")[0]."""
    if NA == 0:
        return 0.0
    return float(_beta.ppf(alpha, NA, N - NA + 1))


def binom_test(x: int, n: int, p: float = 0.5) -> float:
    """scipy 1.7.1 stats.binom_test, alternative='two-sided' (call site smoothing.py:76)."""
    d = _binom.pmf(x, n, p)
    rerr = 1 + 1e-7
    if x == p * n:
        pval = 1.0
    elif x < p * n:
        i = np.arange(np.ceil(p * n), n + 1)
        y = np.sum(_binom.pmf(i, n, p) <= d * rerr, axis=0)
        pval = _binom.cdf(x, n, p) + _binom.sf(n - y, n, p)
    else:
        i = np.arange(np.floor(p * n) + 1)
        y = np.sum(_binom.pmf(i, n, p) <= d * rerr, axis=0)
        pval = _binom.cdf(y - 1, n, p) + _binom.sf(x - 1, n, p)
    return float(min(1.0, pval))


def norm_ppf(p: float) -> float:
    return float(_norm.ppf(p))


def count_arr(arr, length: int) -> np.ndarray:
    """smoothing.py:101-105."""
    counts = np.zeros(length, dtype=int)
    for idx in arr:
        counts[idx] += 1
    return counts


def certify_from_counts(counts_selection, counts_estimation, n: int, alpha: float, sigma: float):
    """smoothing.py:44-56 with the two `_sample_noise` results given."""
    counts_selection = np.asarray(counts_selection)
    counts_estimation = np.asarray(counts_estimation)
    cAHat = int(counts_selection.argmax())                  # :46 first max index on ties
    nA = int(counts_estimation[cAHat])                      # :50
    pABar = lower_confidence_bound(nA, n, alpha)            # :51
    if pABar < 0.5:                                         # :52
        return ABSTAIN, 0.0
    return cAHat, sigma * norm_ppf(pABar)                   # :55-56


def predict_from_counts(counts, alpha: float) -> int:
    """smoothing.py:72-79 with the `_sample_noise` result given."""
    counts = np.asarray(counts)
    top2 = counts.argsort()[::-1][:2]                       # :73
    count1 = int(counts[top2[0]])
    count2 = int(counts[top2[1]])
    if binom_test(count1, count1 + count2, p=0.5) > alpha:  # :76
        return ABSTAIN
    return int(top2[0])


class SmoothOracle:
    """Line-for-line shaped like the reference class; `base_classifier` maps a float32 numpy/torch
    batch [B,C,H,W] to logits [B,num_classes] on the CPU.  `noise_fn(first_sample, num, shape)` supplies the
    Gaussian draws (the product's counter-based stream restated in oracle/philox.py), replacing the
    reference's `torch.randn_like(batch, device='cuda')` (smoothing.py:96) which is not reproducible."""

    ABSTAIN = ABSTAIN

    def __init__(self, base_classifier, num_classes: int, sigma: float, noise_fn):
        self.base_classifier = base_classifier
        self.num_classes = num_classes
        self.sigma = sigma
        self.noise_fn = noise_fn
        self._cursor = 0  # global sample index: selection samples first, estimation samples after

    def certify(self, x, n0: int, n: int, alpha: float, batch_size: int):
        self._cursor = 0
        counts_selection = self._sample_noise(x, n0, batch_size)
        counts_estimation = self._sample_noise(x, n, batch_size)
        return certify_from_counts(counts_selection, counts_estimation, n, alpha, self.sigma)

    def predict(self, x, n: int, alpha: float, batch_size: int):
        self._cursor = 0
        counts = self._sample_noise(x, n, batch_size)
        return predict_from_counts(counts, alpha)

    def _sample_noise(self, x, num: int, batch_size: int) -> np.ndarray:
        """smoothing.py:81-99."""
        x = np.asarray(x, dtype=np.float32)
        counts = np.zeros(self.num_classes, dtype=int)
        for _ in range(ceil(num / batch_size)):
            this_batch_size = min(batch_size, num)
            num -= this_batch_size
            batch = np.repeat(x[None], this_batch_size, axis=0)                         # :95
            noise = self.noise_fn(self._cursor, this_batch_size, x.shape) * np.float32(self.sigma)  # :96
            self._cursor += this_batch_size
            logits = np.asarray(self.base_classifier(batch + noise))
            predictions = logits.argmax(1)                                              # :97
            counts += count_arr(predictions, self.num_classes)                          # :98
        return counts
