"""ORACLE (test infrastructure) -- fp32 PyTorch-CPU restatement of the base classifier the reference's
`Smooth` would wrap: EVA-ViT-G forward_features, ln_vision, Q-Former query path, llama_proj
(MiniGPT4.encode_img, graphs/models/minigpt4/models/minigpt4.py:121-149) plus the build-side label head.

Parameters are a flat dict {reference state_dict name: torch.float32 tensor}.  Pinned against the reference's
own classes by tests/golden/model_golden.npz (oracle/gen_golden_model.py; tests/test_oracle_model.py).

CPU semantics: on a CPU device the reference runs without autocast (base_model.py:135-136), i.e. fp32
arithmetic; the HIP path uses fp16 operands with fp32 accumulation (the reference's `cuda` autocast
contract, base_model.py:141-142, eva_vit.py:407-414), so GPU-vs-oracle comparisons carry a tolerance.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from . import philox

MODE_VIT_HEAD = 0
MODE_ENCODE_IMG = 1


@dataclass
class Config:
    """Mirror of cgpt_config (include/cgpt.h).  Defaults = create_eva_vit_g (eva_vit.py:425-438) and
    init_Qformer (minigpt4.py:90-119, BERT-base)."""
    mode: int = MODE_VIT_HEAD
    num_classes: int = 1000
    img_size: int = 224
    patch_size: int = 14
    vit_dim: int = 1408
    vit_depth: int = 39
    vit_heads: int = 16
    vit_mlp: int = 6144            # int(1408 * 4.3637)
    vit_ln_eps: float = 1e-6
    ln_vision_eps: float = 1e-5
    qf_layers: int = 12
    qf_dim: int = 768
    qf_heads: int = 12
    qf_ffn: int = 3072
    qf_queries: int = 32
    qf_xattn_freq: int = 2
    qf_ln_eps: float = 1e-12
    proj_dim: int = 4096

    @property
    def tokens(self):
        return (self.img_size // self.patch_size) ** 2 + 1


def tiny_config(mode=MODE_VIT_HEAD, num_classes=10):
    """Small shape used by fixtures/tests: 2 ViT heads of 88, 2 Q-Former heads of 64."""
    return Config(mode=mode, num_classes=num_classes, img_size=56, patch_size=14, vit_dim=176, vit_depth=2,
                  vit_heads=2, vit_mlp=int(176 * 4.3637), qf_layers=2, qf_dim=128, qf_heads=2, qf_ffn=256,
                  qf_queries=8, qf_xattn_freq=2, proj_dim=192)


# --------------------------------------------------------------------------------------- parameters
def param_shapes(cfg: Config):
    """Ordered {name: shape} of every tensor the classifier uses (reference state_dict names)."""
    D, P = cfg.vit_dim, cfg.patch_size
    s = {}
    s["visual_encoder.cls_token"] = (1, 1, D)                                   # eva_vit.py:270
    s["visual_encoder.pos_embed"] = (1, cfg.tokens, D)                          # :272
    s["visual_encoder.patch_embed.proj.weight"] = (D, 3, P, P)                  # :202
    s["visual_encoder.patch_embed.proj.bias"] = (D,)
    for i in range(cfg.vit_depth):
        b = f"visual_encoder.blocks.{i}."
        s[b + "norm1.weight"] = (D,); s[b + "norm1.bias"] = (D,)                # :162
        s[b + "attn.q_bias"] = (D,); s[b + "attn.v_bias"] = (D,)                # :83-84
        s[b + "attn.qkv.weight"] = (3 * D, D)                                   # :81
        s[b + "attn.proj.weight"] = (D, D); s[b + "attn.proj.bias"] = (D,)      # :120
        s[b + "norm2.weight"] = (D,); s[b + "norm2.bias"] = (D,)                # :168
        s[b + "mlp.fc1.weight"] = (cfg.vit_mlp, D); s[b + "mlp.fc1.bias"] = (cfg.vit_mlp,)   # :54
        s[b + "mlp.fc2.weight"] = (D, cfg.vit_mlp); s[b + "mlp.fc2.bias"] = (D,)             # :56
    s["ln_vision.weight"] = (D,); s["ln_vision.bias"] = (D,)                    # base_model.py:281
    if cfg.mode == MODE_ENCODE_IMG:
        H = cfg.qf_dim
        s["query_tokens"] = (1, cfg.qf_queries, H)                              # minigpt4.py:99
        s["Qformer.bert.embeddings.LayerNorm.weight"] = (H,)                    # Qformer.py:65
        s["Qformer.bert.embeddings.LayerNorm.bias"] = (H,)
        for i in range(cfg.qf_layers):
            L = f"Qformer.bert.encoder.layer.{i}."
            atts = ["attention"] + (["crossattention"] if i % cfg.qf_xattn_freq == 0 else [])
            for att in atts:
                kv_in = cfg.vit_dim if att == "crossattention" else H           # Qformer.py:127-129
                s[L + att + ".self.query.weight"] = (H, H); s[L + att + ".self.query.bias"] = (H,)
                s[L + att + ".self.key.weight"] = (H, kv_in); s[L + att + ".self.key.bias"] = (H,)
                s[L + att + ".self.value.weight"] = (H, kv_in); s[L + att + ".self.value.bias"] = (H,)
                s[L + att + ".output.dense.weight"] = (H, H); s[L + att + ".output.dense.bias"] = (H,)
                s[L + att + ".output.LayerNorm.weight"] = (H,); s[L + att + ".output.LayerNorm.bias"] = (H,)
            s[L + "intermediate_query.dense.weight"] = (cfg.qf_ffn, H)          # Qformer.py:399
            s[L + "intermediate_query.dense.bias"] = (cfg.qf_ffn,)
            s[L + "output_query.dense.weight"] = (H, cfg.qf_ffn)                # :400
            s[L + "output_query.dense.bias"] = (H,)
            s[L + "output_query.LayerNorm.weight"] = (H,); s[L + "output_query.LayerNorm.bias"] = (H,)
        s["llama_proj.weight"] = (cfg.proj_dim, H); s["llama_proj.bias"] = (cfg.proj_dim,)   # minigpt4.py:76
        s["head.weight"] = (cfg.num_classes, cfg.proj_dim)
    else:
        s["head.weight"] = (cfg.num_classes, D)
    s["head.bias"] = (cfg.num_classes,)
    return s


def init_params(cfg: Config, seed: int, randomize_affine: bool = True):
    """Seeded parameters following the reference init law: Linear/Conv weights ~ N(0, .02) (trunc at +-2
    never binds: eva_vit.py:316-318, Qformer.py:664-674), proj/fc2 weights / sqrt(2*layer_id)
    (eva_vit.py:308-314), query_tokens ~ N(0,.02) (minigpt4.py:99-102).  With randomize_affine the
    biases / LayerNorm affine parameters get small random values instead of the (0 / 1) init so that every
    term of the forward is exercised by parity tests.  Draws come from the counter-based stream
    (oracle/philox.py) with stream id = index of the tensor, so the table is stable forever."""
    out = {}
    for tid, (name, shape) in enumerate(param_shapes(cfg).items()):
        n = int(np.prod(shape))
        z = philox.normal_stream(seed, tid, n, hi_word=1).reshape(shape)
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("LayerNorm.weight") \
                or name == "ln_vision.weight":
            w = 1.0 + (0.1 * z if randomize_affine else 0.0 * z)
        elif name.endswith("bias") or name.endswith("q_bias") or name.endswith("v_bias"):
            w = (0.02 * z) if randomize_affine else 0.0 * z
        else:
            w = 0.02 * z
            if name.startswith("head."):
                w = 0.05 * z   # build-side head: spread the logits so that argmax is not a near-tie
            for key in ("attn.proj.weight", "mlp.fc2.weight"):
                if name.endswith(key):
                    layer_id = int(name.split(".")[2]) + 1
                    w = w / math.sqrt(2.0 * layer_id)
        out[name] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return out


def round_fp16_weights(params):
    """What the device computes with: matrices (GEMM operands) rounded through fp16
    (convert_weights_to_fp16, eva_vit.py:407-414); 1-D parameters (biases, LayerNorm affine) and
    cls_token / pos_embed / query_tokens stay fp32.  Mirrors cgpt_load_weight (include/cgpt.h)."""
    out = {}
    for k, v in params.items():
        is_mat = v.dim() >= 2 and not (k.endswith("cls_token") or k.endswith("pos_embed") or k == "query_tokens")
        out[k] = v.half().float() if is_mat else v.clone()
    return out


# ------------------------------------------------------------------------------------------ forward
def vit_forward(p, x, cfg: Config):
    """VisionTransformer.forward_features (eva_vit.py:332-349): [B,3,H,W] -> [B,T,D]; no final norm."""
    pre = "visual_encoder."
    B = x.shape[0]
    D, Hh = cfg.vit_dim, cfg.vit_heads
    h = F.conv2d(x, p[pre + "patch_embed.proj.weight"], p[pre + "patch_embed.proj.bias"], stride=cfg.patch_size)
    h = h.flatten(2).transpose(1, 2)                                            # :209
    h = torch.cat((p[pre + "cls_token"].expand(B, -1, -1), h), dim=1)           # :337-338
    h = h + p[pre + "pos_embed"]                                                # :340
    scale = (D // Hh) ** -0.5                                                   # :79
    for i in range(cfg.vit_depth):
        b = f"{pre}blocks.{i}."
        y = F.layer_norm(h, (D,), p[b + "norm1.weight"], p[b + "norm1.bias"], cfg.vit_ln_eps)
        qkv_bias = torch.cat((p[b + "attn.q_bias"], torch.zeros_like(p[b + "attn.v_bias"]), p[b + "attn.v_bias"]))  # :127
        qkv = F.linear(y, p[b + "attn.qkv.weight"], qkv_bias)                   # :129
        qkv = qkv.reshape(B, -1, 3, Hh, D // Hh).permute(2, 0, 3, 1, 4)         # :130
        q, k, v = qkv[0] * scale, qkv[1], qkv[2]                                # :131-133
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)                        # :134,147
        y = (attn @ v).transpose(1, 2).reshape(B, -1, D)                        # :150
        h = h + F.linear(y, p[b + "attn.proj.weight"], p[b + "attn.proj.bias"]) # :151,180
        y = F.layer_norm(h, (D,), p[b + "norm2.weight"], p[b + "norm2.bias"], cfg.vit_ln_eps)
        y = F.gelu(F.linear(y, p[b + "mlp.fc1.weight"], p[b + "mlp.fc1.bias"]))  # :60-61 exact-erf GELU
        h = h + F.linear(y, p[b + "mlp.fc2.weight"], p[b + "mlp.fc2.bias"])     # :64,181
    return h


def _bert_attention(p, prefix, hidden, kv_src, cfg: Config):
    """BertAttention = BertSelfAttention (Qformer.py:169-275) + BertSelfOutput (:278-289); all masks are 0."""
    B, Tq, H = hidden.shape
    nh, hd = cfg.qf_heads, cfg.qf_dim // cfg.qf_heads
    q = F.linear(hidden, p[prefix + ".self.query.weight"], p[prefix + ".self.query.bias"])
    k = F.linear(kv_src, p[prefix + ".self.key.weight"], p[prefix + ".self.key.bias"])
    v = F.linear(kv_src, p[prefix + ".self.value.weight"], p[prefix + ".self.value.bias"])
    q = q.view(B, Tq, nh, hd).permute(0, 2, 1, 3)
    k = k.view(B, -1, nh, hd).permute(0, 2, 1, 3)
    v = v.view(B, -1, nh, hd).permute(0, 2, 1, 3)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)                               # :244
    ctx = (s.softmax(dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, Tq, H)         # :250-266
    o = F.linear(ctx, p[prefix + ".output.dense.weight"], p[prefix + ".output.dense.bias"])
    return F.layer_norm(o + hidden, (H,), p[prefix + ".output.LayerNorm.weight"],
                        p[prefix + ".output.LayerNorm.bias"], cfg.qf_ln_eps)    # :288


def qformer_forward(p, image_embeds, cfg: Config):
    """Qformer.bert(query_embeds=query_tokens, encoder_hidden_states=image_embeds) query path
    (minigpt4.py:132-139; Qformer.py:78-108 embeddings, :402-484 layers) -> last_hidden_state [B,Q,H]."""
    B = image_embeds.shape[0]
    H = cfg.qf_dim
    h = p["query_tokens"].expand(B, -1, -1)
    h = F.layer_norm(h, (H,), p["Qformer.bert.embeddings.LayerNorm.weight"],
                     p["Qformer.bert.embeddings.LayerNorm.bias"], cfg.qf_ln_eps)  # :106
    for i in range(cfg.qf_layers):
        L = f"Qformer.bert.encoder.layer.{i}."
        h = _bert_attention(p, L + "attention", h, h, cfg)                      # :415-421
        if i % cfg.qf_xattn_freq == 0:                                          # :386-395,432-441
            h = _bert_attention(p, L + "crossattention", h, image_embeds, cfg)
        y = F.gelu(F.linear(h, p[L + "intermediate_query.dense.weight"], p[L + "intermediate_query.dense.bias"]))  # :481
        y = F.linear(y, p[L + "output_query.dense.weight"], p[L + "output_query.dense.bias"])
        h = F.layer_norm(y + h, (H,), p[L + "output_query.LayerNorm.weight"],
                         p[L + "output_query.LayerNorm.bias"], cfg.qf_ln_eps)   # :483, :372
    return h


def forward_all(p, x, cfg: Config):
    """Returns dict of stages: vit_out, ln_vision (CLS row only in MODE_VIT_HEAD), qformer, llama, logits."""
    with torch.no_grad():
        x = torch.as_tensor(x, dtype=torch.float32)
        out = {}
        vit = vit_forward(p, x, cfg)
        out["vit_out"] = vit
        D = cfg.vit_dim
        if cfg.mode == MODE_VIT_HEAD:
            cls = F.layer_norm(vit[:, 0], (D,), p["ln_vision.weight"], p["ln_vision.bias"], cfg.ln_vision_eps)
            out["logits"] = F.linear(cls, p["head.weight"], p["head.bias"])
            return out
        emb = F.layer_norm(vit, (D,), p["ln_vision.weight"], p["ln_vision.bias"], cfg.ln_vision_eps)  # minigpt4.py:129
        out["ln_vision"] = emb
        qf = qformer_forward(p, emb, cfg)
        out["qformer"] = qf
        llama = F.linear(qf, p["llama_proj.weight"], p["llama_proj.bias"])      # minigpt4.py:141
        out["llama"] = llama
        out["logits"] = F.linear(llama.mean(dim=1), p["head.weight"], p["head.bias"])
        return out


def interpolate_pos_embed(pos_embed_checkpoint, num_patches):
    """eva_vit.py:383-404 restated: keep the extra (class) token, bicubic-resize the square grid of patch tokens."""
    pe = torch.as_tensor(pos_embed_checkpoint, dtype=torch.float32)
    embedding_size = pe.shape[-1]
    num_extra_tokens = 1                                               # model.pos_embed.shape[-2] - num_patches (:388)
    orig_size = int((pe.shape[-2] - num_extra_tokens) ** 0.5)          # :390
    new_size = int(num_patches ** 0.5)                                 # :392
    if orig_size == new_size:
        return pe
    extra_tokens = pe[:, :num_extra_tokens]
    pos_tokens = pe[:, num_extra_tokens:]
    pos_tokens = pos_tokens.reshape(-1, orig_size, orig_size, embedding_size).permute(0, 3, 1, 2)
    pos_tokens = F.interpolate(pos_tokens, size=(new_size, new_size), mode="bicubic", align_corners=False)   # :400-401
    pos_tokens = pos_tokens.permute(0, 2, 3, 1).flatten(1, 2)
    return torch.cat((extra_tokens, pos_tokens), dim=1)                # :403


def make_classifier(p, cfg: Config):
    """base_classifier([B,C,H,W] float32) -> logits [B,num_classes] numpy, for SmoothOracle."""
    def f(batch):
        return forward_all(p, batch, cfg)["logits"].numpy()
    return f


def synthetic_image(cfg: Config, seed: int = 1234):
    """x = (u - mean)/std, u ~ U[0,1)^{3xHxW}, CLIP statistics
    (processors/base_processor.py:18-20).  u comes from the Philox words (24-bit uniforms)."""
    n = 3 * cfg.img_size * cfg.img_size
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    r = philox.philox4x32_10(g, np.uint64(2), np.uint64(0), np.uint64(0), seed & 0xFFFFFFFF, seed >> 32)
    u = (np.stack(r, axis=1).reshape(-1)[:n] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    u = u.reshape(3, cfg.img_size, cfg.img_size)
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)[:, None, None]
    std = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)[:, None, None]
    return ((u - mean) / std).astype(np.float32)
