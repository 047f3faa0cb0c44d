"""Random-gradient-free (RGF) black-box attack against a smoothed classifier -- BASELINE configs[4] / SURVEY.md 8(f) rank 4.

The reference has NO code for its attack stage (prose only: README.md:62-64,108-120; an AttackVLM-style query attack), so
the schedule below is this build's own definition ("parity unpinned" against the reference; pinned against oracle/rgf_oracle.py,
the plain numpy restatement of the same rule).  The attacker sees only what `Smooth` exposes: vote histograms.

    for step in range(steps):                               # 8 in the BASELINE scenario
        base   = share_t(x_adv)                             # n noisy forwards (Smooth._sample_noise)
        c_i    = (share_t(x_adv + delta * u_i) - base) / delta      for q random directions u_i      # q * n forwards
        x_adv  = clip(x_adv + lr * sign(sum_i c_i u_i),  x - eps, x + eps)                            # cgpt_rgf_step
    label = smooth.predict(x_adv, n, alpha, batch_size)

share_t = vote share of the target class (targeted: ascend) or of the true class (untargeted: descend).  All evaluations of
one step reuse the same smoothing-noise sample indices (common random numbers), so the finite difference sees the change of
the image, not a fresh Monte-Carlo draw.  Directions come from the library's counter-based normal stream under `dir_seed`:
u_i is exactly what cgpt_noise_batch adds for sample index i, and cgpt_rgf_step regenerates it on the device -- no direction
is ever stored.  With torch.distributed initialised the n forwards of every evaluation are sharded by Smooth as usual.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .classifier import noise_batch, _stream_ptr

MAX_DIRS = 32


def rgf_step(x_adv, x_clean, first_dir, coeffs, lr, eps, dir_seed, out=None):
    """out = clamp(x_adv + lr * sign(sum_i coeffs[i] * u_{first_dir+i}), x_clean +- eps) on the device (cgpt_rgf_step)."""
    L = _lib.lib()
    if not (x_adv.is_cuda and x_clean.is_cuda):
        raise RuntimeError("rgf_step needs device tensors: certifiedgpt_amd has no CPU path")
    x_adv = x_adv.float().contiguous()
    x_clean = x_clean.float().contiguous()
    c = np.ascontiguousarray(coeffs, dtype=np.float32)
    if not 1 <= c.size <= MAX_DIRS:
        raise ValueError(f"1..{MAX_DIRS} directions per step")
    out = torch.empty_like(x_adv) if out is None else out
    _lib.check(L.cgpt_rgf_step(C.c_void_p(x_adv.data_ptr()), C.c_void_p(x_clean.data_ptr()), x_adv.numel(), int(first_dir),
                               int(c.size), c.ctypes.data_as(C.c_void_p), float(lr), float(eps), int(dir_seed),
                               C.c_void_p(out.data_ptr()), _stream_ptr()))
    return out


class RGFAttack(object):
    """Query attack on `smooth` (a certifiedgpt_amd.Smooth).  See the module docstring for the rule."""

    def __init__(self, smooth, steps: int = 8, num_dirs: int = 1, delta: float = 0.5, lr: float = 0.05, eps: float = 0.25,
                 dir_seed: int = 1234):
        if not 1 <= num_dirs <= MAX_DIRS:
            raise ValueError(f"num_dirs must be 1..{MAX_DIRS}")
        self.smooth, self.steps, self.num_dirs = smooth, int(steps), int(num_dirs)
        self.delta, self.lr, self.eps, self.dir_seed = float(delta), float(lr), float(eps), int(dir_seed)
        self._next_dir = 0

    def _share(self, x, label, n, batch_size, first_sample):
        self.smooth.reset(first_sample)                       # common random numbers within a step
        counts = self.smooth._sample_noise(x, n, batch_size)
        return float(counts[label]) / float(n), counts

    def attack(self, x, label: int, n: int, alpha: float, batch_size: int, targeted: bool = True):
        """Returns (x_adv, final_label, history); history[k] = vote share of `label` at x_adv before step k (+ the final one).
        targeted=True drives the smoothed prediction TOWARDS `label`; False drives it AWAY from `label` (the true class)."""
        x = x.float().contiguous()
        x_adv = x.clone()
        cursor = self.smooth._next_sample
        history = []
        sign = 1.0 if targeted else -1.0
        for _ in range(self.steps):
            # x_adv and its q probes go through the classifier together, all under the same n noise draws
            probes = [noise_batch(x_adv, self._next_dir + i, 1, self.delta, self.dir_seed)[0] for i in range(self.num_dirs)]
            self.smooth.reset(cursor)
            counts = self.smooth.sample_noise_many(torch.stack([x_adv] + probes), n, batch_size, common_noise=True)
            shares = counts[:, label].astype(np.float64) / float(n)
            base = float(shares[0])
            history.append(base)
            coeffs = [(float(shares[1 + i]) - base) / self.delta for i in range(self.num_dirs)]
            x_adv = rgf_step(x_adv, x, self._next_dir, coeffs, sign * self.lr, self.eps, self.dir_seed)
            self._next_dir += self.num_dirs
            cursor += n
        final_share, counts = self._share(x_adv, label, n, batch_size, cursor)
        history.append(final_share)
        final_label = self.smooth.predict_from_counts(counts, alpha)
        self.smooth.reset(cursor + n)
        return x_adv, final_label, history

    def forwards_per_image(self, n: int) -> int:
        return (self.steps * (1 + self.num_dirs) + 1) * n
