"""CPU restatement of the RGF attack rule of certifiedgpt_amd/rgf.py (TEST INFRASTRUCTURE: only tests/, smoke() and the
bench's cpu_baseline leg may import anything under oracle/).

The reference ships no attack code (README.md:62-64,108-120 is prose), so there is nothing of the reference's to pin this
against: "parity unpinned".  What this file pins is the product's HIP kernel and host loop against a plain numpy statement of
the same build-side rule.
"""
import numpy as np

from . import philox


def direction(dir_seed: int, index: int, shape) -> np.ndarray:
    """u_index: the N(0,1) image the noise stream holds for sample `index` under `dir_seed` (float32)."""
    return philox.normal_stream(dir_seed, index, int(np.prod(shape))).reshape(shape)


def rgf_step(x_adv, x_clean, dirs, coeffs, lr: float, eps: float) -> np.ndarray:
    """clamp(x_adv + lr * sign(sum_i coeffs[i] * dirs[i]), x_clean +- eps); float32, separate multiply and add, index order."""
    x_adv = np.asarray(x_adv, dtype=np.float32)
    x_clean = np.asarray(x_clean, dtype=np.float32)
    g = np.zeros_like(x_adv, dtype=np.float32)
    for c, u in zip(coeffs, dirs):
        g = (g + np.float32(c) * np.asarray(u, dtype=np.float32)).astype(np.float32)
    v = (x_adv + np.float32(lr) * np.sign(g).astype(np.float32)).astype(np.float32)
    return np.minimum(np.maximum(v, x_clean - np.float32(eps)), x_clean + np.float32(eps)).astype(np.float32)


def attack(share_fn, direction_fn, x, steps: int, num_dirs: int, delta: float, lr: float, eps: float, targeted: bool = True):
    """The loop of RGFAttack.attack with the vote share supplied by `share_fn(image, step) -> float` and the directions by
    `direction_fn(index) -> ndarray`.  Returns (x_adv, history)."""
    x = np.asarray(x, dtype=np.float32)
    x_adv = x.copy()
    history, next_dir = [], 0
    sign = 1.0 if targeted else -1.0
    for step in range(steps):
        base = share_fn(x_adv, step)
        history.append(base)
        dirs = [direction_fn(next_dir + i) for i in range(num_dirs)]
        coeffs = []
        for u in dirs:
            # fma(delta, u, x) as cgpt_noise_batch computes it: the float32 product is exact in float64
            xq = (np.float64(np.float32(delta)) * u.astype(np.float64) + x_adv.astype(np.float64)).astype(np.float32)
            coeffs.append((share_fn(xq, step) - base) / delta)
        x_adv = rgf_step(x_adv, x, dirs, np.asarray(coeffs, dtype=np.float32), sign * lr, eps)
        next_dir += num_dirs
    history.append(share_fn(x_adv, steps))
    return x_adv, history
