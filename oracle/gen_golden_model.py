#!/usr/bin/env python3
"""Generate tests/golden/model_golden.npz by executing the REFERENCE's own model classes on a tiny shape.

Runs ONLY in the build container (needs /root/reference); neither this harness's stubs nor any reference
file travels -- only the .npz (inputs + per-stage outputs) does.  Weights are NOT stored: they are
regenerated from oracle.model_oracle.init_params(cfg, seed) (counter-based, stable).

    python oracle/gen_golden_model.py

How the reference classes are reached:
  * graphs/models/minigpt4/models/eva_vit.py is loaded BY FILE PATH after registering stand-in modules for
    names that are not on the forward arithmetic (timm helpers, the registry, download_cached_file --
    used only inside create_eva_vit_g, eva_vit.py:425-460, which needs a checkpoint that is not here).
  * graphs/models/minigpt4/models/Qformer.py is loaded BY FILE PATH after re-exporting three helpers that
    transformers moved (apply_chunking_to_forward, prune_linear_layer -> transformers.pytorch_utils;
    find_pruneable_heads_and_indices is only used by prune_heads, Qformer.py:299-320).  BertModel cannot be
    constructed under transformers 5.x (reference pins 4.30.0), so the plain nn.Module blocks BertEmbeddings +
    BertEncoder are driven exactly as BertModel.forward drives them (Qformer.py:878-950) with the
    all-zero extended masks of Qformer.py:798-801 / :713-802.
  * MiniGPT4 itself cannot be constructed (needs Vicuna + torch_xla); its glue (minigpt4.py:121-149:
    ln_vision -> Qformer -> llama_proj) is three lines, restated below with torch.nn modules.
"""
import importlib.util
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model_oracle as mo  # noqa: E402

REF = "/root/reference/graphs/models/minigpt4/models"


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_eva_vit():
    _module("timm"); _module("timm.models")
    _module("timm.models.layers", drop_path=lambda x, p, t: x, to_2tuple=lambda v: (v, v),
            trunc_normal_=torch.nn.init.trunc_normal_)
    _module("timm.models.registry", register_model=lambda f: f)
    _module("common"); _module("common.registry", registry=types.SimpleNamespace())
    for n in ("graphs", "graphs.models", "graphs.models.minigpt4", "graphs.models.minigpt4.common"):
        _module(n)
    _module("graphs.models.minigpt4.common.dist_utils", download_cached_file=None)
    spec = importlib.util.spec_from_file_location("ref_eva_vit", os.path.join(REF, "eva_vit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for n in [k for k in sys.modules if k == "timm" or k.startswith("timm.")]:
        del sys.modules[n]        # the stand-ins must not be visible to transformers' package probing
    return mod


def load_qformer():
    import transformers.modeling_utils as tmu
    import transformers.pytorch_utils as tpu
    for n in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(tmu, n):
            setattr(tmu, n, getattr(tpu, n))
    if not hasattr(tmu, "find_pruneable_heads_and_indices"):
        tmu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    spec = importlib.util.spec_from_file_location("ref_qformer", os.path.join(REF, "Qformer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.manual_seed(0)
    cfg = mo.tiny_config(mode=mo.MODE_ENCODE_IMG, num_classes=10)
    seed = 20251121
    params = mo.init_params(cfg, seed)

    # ---------------- reference ViT ----------------
    eva = load_eva_vit()
    vit = eva.VisionTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, use_mean_pooling=False,
                                embed_dim=cfg.vit_dim, depth=cfg.vit_depth, num_heads=cfg.vit_heads,
                                mlp_ratio=4.3637, qkv_bias=True, drop_path_rate=0,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6)).eval()
    assert vit.blocks[0].mlp.fc1.out_features == cfg.vit_mlp
    sd = {k[len("visual_encoder."):]: v for k, v in params.items() if k.startswith("visual_encoder.")}
    missing, unexpected = vit.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)

    # inputs: the synthetic normalised image + sigma * the counter-based noise (3 samples)
    from oracle import philox
    x0 = mo.synthetic_image(cfg, seed=1234)
    noise = philox.noise_batch(42, 0, 3, x0.shape)
    x = torch.from_numpy(x0[None] + np.float32(0.5) * noise)
    with torch.no_grad():
        ref_vit = vit(x)

    # ---------------- reference Q-Former blocks ----------------
    qf = load_qformer()
    bcfg = qf.BertConfig(hidden_size=cfg.qf_dim, num_hidden_layers=cfg.qf_layers, num_attention_heads=cfg.qf_heads,
                         intermediate_size=cfg.qf_ffn)              # other fields: BERT-base defaults
    bcfg.encoder_width = cfg.vit_dim                                # minigpt4.py:93-97
    bcfg.add_cross_attention = True
    bcfg.cross_attention_freq = cfg.qf_xattn_freq
    bcfg.query_length = cfg.qf_queries
    assert bcfg.layer_norm_eps == 1e-12 and bcfg.hidden_act == "gelu"
    emb = qf.BertEmbeddings(bcfg).eval()
    enc = qf.BertEncoder(bcfg).eval()
    emb.word_embeddings = None; emb.position_embeddings = None      # minigpt4.py:105-106
    for layer in enc.layer:
        layer.output = None; layer.intermediate = None              # minigpt4.py:107-109
    sd_emb = {k[len("Qformer.bert.embeddings."):]: v for k, v in params.items()
              if k.startswith("Qformer.bert.embeddings.")}
    m, u = emb.load_state_dict(sd_emb, strict=False)
    assert not u and set(m) <= {"position_ids"}, (m, u)
    sd_enc = {k[len("Qformer.bert.encoder."):]: v for k, v in params.items()
              if k.startswith("Qformer.bert.encoder.")}
    m, u = enc.load_state_dict(sd_enc, strict=False)
    assert not u and not m, (m, u)

    ln_vision = nn.LayerNorm(cfg.vit_dim)                            # base_model.py:281-287 (fp32 LN)
    ln_vision.load_state_dict({"weight": params["ln_vision.weight"], "bias": params["ln_vision.bias"]})
    llama_proj = nn.Linear(cfg.qf_dim, cfg.proj_dim)                 # minigpt4.py:76-78
    llama_proj.load_state_dict({"weight": params["llama_proj.weight"], "bias": params["llama_proj.bias"]})
    with torch.no_grad():
        B = x.shape[0]
        image_embeds = ln_vision(ref_vit)                            # minigpt4.py:129
        query_tokens = params["query_tokens"].expand(B, -1, -1)      # :132
        hidden = emb(query_embeds=query_tokens)                      # Qformer.py:878-883
        # BertModel.forward: ones masks -> get_extended_attention_mask / invert_attention_mask -> zeros
        att = torch.zeros(B, 1, 1, cfg.qf_queries)
        enc_att = torch.zeros(B, 1, 1, cfg.tokens)
        out = enc(hidden, attention_mask=att, head_mask=[None] * cfg.qf_layers,
                  encoder_hidden_states=image_embeds, encoder_attention_mask=enc_att,
                  query_length=cfg.qf_queries)                       # Qformer.py:937-949
        ref_q = out.last_hidden_state if hasattr(out, "last_hidden_state") else out[0]
        ref_llama = llama_proj(ref_q)                                # minigpt4.py:141

    # ---------------- reference interpolate_pos_embed (eva_vit.py:383-404): a 2x2-grid checkpoint resized to this 4x4 model
    ck_pos = philox.normal_stream(77, 0, (1 + 4) * cfg.vit_dim).reshape(1, 5, cfg.vit_dim)
    ck = {"pos_embed": torch.from_numpy(ck_pos.copy())}
    eva.interpolate_pos_embed(vit, ck)
    ref_interp = ck["pos_embed"].numpy()
    assert ref_interp.shape == (1, cfg.tokens, cfg.vit_dim)

    path = os.path.join(ROOT, "tests", "golden", "model_golden.npz")
    np.savez_compressed(path, seed=np.int64(seed), x=x.numpy(), vit_out=ref_vit.numpy(), pos_ck=ck_pos, pos_interp=ref_interp,
                        ln_vision=image_embeds.numpy(), qformer=ref_q.numpy(), llama=ref_llama.numpy())
    print("wrote", path, os.path.getsize(path), "bytes; vit_out", tuple(ref_vit.shape), "qformer", tuple(ref_q.shape))

    # self-check: the oracle restatement reproduces the reference's outputs
    got = mo.forward_all(params, x, cfg)
    for k, ref in (("vit_out", ref_vit), ("ln_vision", image_embeds), ("qformer", ref_q), ("llama", ref_llama)):
        print(k, "max|oracle-ref| =", float((got[k] - ref).abs().max()), "max|ref| =", float(ref.abs().max()))


if __name__ == "__main__":
    main()
