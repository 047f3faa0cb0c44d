"""certifiedgpt_amd -- MI355X-native randomized-smoothing certify/predict hot path.

Drop-in for the path `Smooth.certify` / `Smooth.predict` of leodesouza/certifiedGPT
(randomized_smoothing/smoothing.py) over MiniGPT-4's image encoder; see DESIGN.md / INTEGRATION.md.
"""
from ._lib import CgptError, lib, LIB_PATH  # noqa: F401
from .smoothing import Smooth, shard_range, batch_plan, LOCAL_ONLY  # noqa: F401
from .classifier import HipClassifier, noise_batch, vote, interpolate_pos_embed  # noqa: F401
from .rgf import RGFAttack, rgf_step  # noqa: F401

__all__ = ["Smooth", "HipClassifier", "noise_batch", "vote", "shard_range", "batch_plan", "LOCAL_ONLY", "RGFAttack", "rgf_step", "CgptError", "lib", "LIB_PATH"]
