"""Pin the model ORACLE (oracle/model_oracle.py) against outputs of the reference's own eva_vit.py / Qformer.py
classes (tests/golden/model_golden.npz, written by oracle/gen_golden_model.py in the build container)."""
import os

import numpy as np
import torch

from oracle import model_oracle as mo, philox
from conftest import GOLDEN


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert [int(v) for v in philox.philox4x32_10(0, 0, 0, 0, 0, 0)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(v) for v in philox.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)] \
        == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert [int(v) for v in philox.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)] \
        == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_noise_stream_is_batch_and_shard_independent():
    shape = (3, 8, 8)
    full = philox.noise_batch(7, 0, 6, shape)
    assert np.array_equal(full[2:5], philox.noise_batch(7, 2, 3, shape))
    z = philox.normal_stream(42, 5, 400_000)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3


def test_model_oracle_matches_reference_outputs():
    g = np.load(os.path.join(GOLDEN, "model_golden.npz"))
    cfg = mo.tiny_config(mode=mo.MODE_ENCODE_IMG, num_classes=10)
    params = mo.init_params(cfg, int(g["seed"]))
    got = mo.forward_all(params, torch.from_numpy(g["x"]), cfg)
    for k in ("vit_out", "ln_vision", "qformer", "llama"):
        err = float(np.abs(got[k].numpy() - g[k]).max())
        assert err <= 2e-5, (k, err)     # identical fp32 op sequence; slack only for BLAS/thread differences


def test_param_table_matches_reference_sizes():
    # reference VisionTransformer(ViT-G) has 985.89 M parameters, Q-Former blocks 105.14 M
    cfg = mo.Config(mode=mo.MODE_ENCODE_IMG)
    shapes = mo.param_shapes(cfg)
    vit = sum(int(np.prod(s)) for k, s in shapes.items() if k.startswith("visual_encoder."))
    assert vit == 985_894_528
    qf = sum(int(np.prod(s)) for k, s in shapes.items() if k.startswith("Qformer."))
    assert abs(qf - 105.14e6) < 0.01e6


def test_interpolate_pos_embed_matches_reference_function():
    g = np.load(os.path.join(GOLDEN, "model_golden.npz"))
    got = mo.interpolate_pos_embed(torch.from_numpy(g["pos_ck"]), 16).numpy()
    assert np.abs(got - g["pos_interp"]).max() <= 1e-6
    import certifiedgpt_amd as cg
    got2 = cg.interpolate_pos_embed(g["pos_ck"], 16)
    assert np.abs(got2 - g["pos_interp"]).max() <= 1e-6
    assert np.array_equal(cg.interpolate_pos_embed(g["pos_interp"], 16), g["pos_interp"])     # same grid: untouched
