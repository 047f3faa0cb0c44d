"""Shared helpers for the GPU parity tests: build a HipClassifier and the oracle parameters for one config."""
import dataclasses

import numpy as np
import torch

import certifiedgpt_amd as cg
from oracle import model_oracle as mo

DEV = "cuda:0"
DIM_FIELDS = ("img_size", "patch_size", "vit_dim", "vit_depth", "vit_heads", "vit_mlp", "qf_layers", "qf_dim", "qf_heads",
              "qf_ffn", "qf_queries", "qf_xattn_freq", "proj_dim")


def make_classifier(cfg: mo.Config, max_batch: int):
    d = dataclasses.asdict(cfg)
    return cg.HipClassifier(mode="encode_img" if cfg.mode == mo.MODE_ENCODE_IMG else "vit_head",
                            num_classes=cfg.num_classes, max_batch=max_batch, vit_ln_eps=cfg.vit_ln_eps,
                            ln_vision_eps=cfg.ln_vision_eps, qf_ln_eps=cfg.qf_ln_eps, **{k: d[k] for k in DIM_FIELDS})


def tiny_pair(mode, num_classes=10, seed=20251121, max_batch=8):
    """(HipClassifier with the oracle's seeded weights loaded, oracle params rounded as the device stores them, cfg)."""
    cfg = mo.tiny_config(mode=mode, num_classes=num_classes)
    params = mo.init_params(cfg, seed)
    clf = make_classifier(cfg, max_batch)
    missing, unexpected = clf.load_state_dict(params)
    assert not missing and not unexpected
    return clf, mo.round_fp16_weights(params), params, cfg


def rel_err(got, ref):
    got = torch.as_tensor(got).float().cpu()
    ref = torch.as_tensor(ref).float().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))

