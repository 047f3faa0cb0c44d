"""HipClassifier -- the base classifier `Smooth` wraps, running entirely in libcgpt.so on one MI355X.

Plays the role of the `base_classifier: torch.nn.Module` argument of the reference's `Smooth.__init__`
(randomized_smoothing/smoothing.py:19-27: "maps from [batch x channel x height x width] to [batch x num_classes]",
must expose `.eval()` and `__call__`), with MiniGPT-4's image encoder as the body
(MiniGPT4.encode_img, graphs/models/minigpt4/models/minigpt4.py:121-149).  PyTorch is used only for device
memory, the current stream and (in Smooth) torch.distributed.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

VIT_G = dict(img_size=224, patch_size=14, vit_dim=1408, vit_depth=39, vit_heads=16, vit_mlp=6144)   # eva_vit.py:425-438
QFORMER = dict(qf_layers=12, qf_dim=768, qf_heads=12, qf_ffn=3072, qf_queries=32, qf_xattn_freq=2, proj_dim=4096)  # minigpt4.py:90-119


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class HipClassifier:
    """mode: "vit_head" (EVA-ViT-G + ln_vision(CLS) + Linear head; BASELINE config 2) or
    "encode_img" (ViT + ln_vision + Q-Former + llama_proj + head on the mean query token)."""

    def __init__(self, mode="vit_head", num_classes=1000, max_batch=100, device=0, vit_ln_eps=1e-6,
                 ln_vision_eps=1e-5, qf_ln_eps=1e-12, **dims):
        L = _lib.lib()
        cfg = dict(VIT_G)
        cfg.update(QFORMER)
        unknown = set(dims) - set(cfg)
        if unknown:
            raise TypeError(f"unknown config fields: {sorted(unknown)}")
        cfg.update(dims)
        self.mode = {"vit_head": _lib.MODE_VIT_HEAD, "encode_img": _lib.MODE_ENCODE_IMG}[mode]
        self.num_classes = int(num_classes)
        self.max_batch = int(max_batch)
        self.device = torch.device("cuda", device)
        self.cfg = cfg
        c = _lib.Config(struct_size=C.sizeof(_lib.Config), mode=self.mode, device=device, num_classes=num_classes,
                        max_batch=max_batch, vit_ln_eps=vit_ln_eps, ln_vision_eps=ln_vision_eps, qf_ln_eps=qf_ln_eps,
                        **cfg)
        self._h = C.c_void_p()
        _lib.check(L.cgpt_create(C.byref(c), C.byref(self._h)))
        self._L = L
        self.tokens = (cfg["img_size"] // cfg["patch_size"]) ** 2 + 1
        self.chw = (3, cfg["img_size"], cfg["img_size"])

    # ---- nn.Module-shaped surface used by Smooth (smoothing.py:42,71,97)
    def eval(self):
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.cgpt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights
    def weight_names(self):
        out, i = [], 0
        while True:
            n = self._L.cgpt_weight_name(self._h, i)
            if n is None:
                return out
            out.append(n.decode())
            i += 1

    def load_state_dict(self, state_dict, strict=True):
        """state_dict: {reference parameter name: array-like float32 in the PyTorch shape}."""
        names = set(self.weight_names())
        missing = sorted(names - set(state_dict))
        unexpected = sorted(set(state_dict) - names)
        if strict and (missing or unexpected):
            raise KeyError(f"missing={missing[:5]}... unexpected={unexpected[:5]}...")
        for k, v in state_dict.items():
            if k not in names:
                continue
            a = np.ascontiguousarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32)
            if k == "visual_encoder.pos_embed" and a.size != self._L.cgpt_weight_numel(self._h, k.encode()):
                a = np.ascontiguousarray(interpolate_pos_embed(a, self.tokens - 1), dtype=np.float32)
            _lib.check(self._L.cgpt_load_weight(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        return missing, unexpected

    def get_weight(self, name):
        n = self._L.cgpt_weight_numel(self._h, name.encode())
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, dtype=np.float32)
        _lib.check(self._L.cgpt_get_weight(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), n))
        return out

    def init_synthetic(self, seed=0):
        _lib.check(self._L.cgpt_init_synthetic_weights(self._h, seed, _stream_ptr()))
        return self

    # ---- forward
    def _check_x(self, x, batched):
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
            raise TypeError("expected a float32 CUDA(HIP) tensor")
        shape = tuple(x.shape[1:]) if batched else tuple(x.shape)
        if shape != self.chw:
            raise ValueError(f"expected image shape {self.chw}, got {shape}")
        return x.contiguous()

    def __call__(self, images):
        """[B,3,H,W] float32 -> logits [B,num_classes] float32 (no noise)."""
        images = self._check_x(images, True)
        B = images.shape[0]
        out = torch.empty((B, self.num_classes), dtype=torch.float32, device=images.device)
        for lo in range(0, B, self.max_batch):
            hi = min(B, lo + self.max_batch)
            _lib.check(self._L.cgpt_classify(self._h, C.c_void_p(images[lo:hi].data_ptr()), hi - lo,
                                             C.c_void_p(out[lo:hi].data_ptr()), _stream_ptr()))
        return out

    def sample_counts(self, x, first_sample, num, batch_size, sigma, seed, counts=None):
        """The `_sample_noise` engine (smoothing.py:81-99) for samples first_sample..first_sample+num-1.
        Returns (and accumulates into) an int64 device tensor [num_classes]; no host sync."""
        x = self._check_x(x, False)
        if counts is None:
            counts = torch.zeros(self.num_classes, dtype=torch.int64, device=x.device)
        if num > 0:
            _lib.check(self._L.cgpt_sample_counts(self._h, C.c_void_p(x.data_ptr()), first_sample, num,
                                                  min(batch_size, self.max_batch), sigma, seed,
                                                  C.c_void_p(counts.data_ptr()), _stream_ptr()))
        return counts

    def sample_counts_pair(self, x, first_a, num_a, first_b, num_b, batch_size, sigma, seed):
        """Selection + estimation draws of one `certify` in a single device pass (cgpt_sample_counts2).
        Returns an int64 device tensor [2, num_classes]: row 0 = votes of samples [first_a, first_a+num_a),
        row 1 = votes of [first_b, first_b+num_b).  Identical to two sample_counts calls."""
        x = self._check_x(x, False)
        counts = torch.zeros((2, self.num_classes), dtype=torch.int64, device=x.device)
        if num_a + num_b > 0:
            _lib.check(self._L.cgpt_sample_counts2(self._h, C.c_void_p(x.data_ptr()), first_a, num_a,
                                                   C.c_void_p(counts[0].data_ptr()), first_b, num_b,
                                                   C.c_void_p(counts[1].data_ptr()), min(batch_size, self.max_batch), sigma,
                                                   seed, _stream_ptr()))
        return counts

    def sample_counts_images(self, xs, first_a, num_a, first_b, num_b, image_stride, sigma, seed):
        """The pair pass for several images at once (cgpt_sample_counts_images): image i draws samples
        first_a + i*image_stride + [0,num_a) and first_b + i*image_stride + [0,num_b).  Returns int64 [G, 2, num_classes];
        identical to G sample_counts_pair calls, but the rows of all images are cut into classifier batches of max_batch rows
        that are not aligned to image boundaries."""
        xs = self._check_x(xs, True)
        G = xs.shape[0]
        counts = torch.zeros((G, 2, self.num_classes), dtype=torch.int64, device=xs.device)
        if num_a + num_b <= 0 or G == 0:
            return counts
        _lib.check(self._L.cgpt_sample_counts_images(self._h, C.c_void_p(xs.data_ptr()), G, first_a, num_a, first_b, num_b,
                                                     image_stride, C.c_void_p(counts.data_ptr()), sigma, seed, _stream_ptr()))
        return counts

    def encode_img(self, images):
        """MiniGPT4.encode_img (minigpt4.py:121-149): [B,3,H,W] float32 -> (inputs_llama [B, queries, proj_dim] float32,
        atts_llama ones [B, queries] long).  mode="encode_img" classifiers only."""
        images = self._check_x(images, True)
        B = images.shape[0]
        out = torch.empty((B, self.cfg["qf_queries"], self.cfg["proj_dim"]), dtype=torch.float32, device=images.device)
        _lib.check(self._L.cgpt_encode_img(self._h, C.c_void_p(images.data_ptr()), B, C.c_void_p(out.data_ptr()), _stream_ptr()))
        return out, torch.ones(out.shape[:-1], dtype=torch.long, device=images.device)

    def encode_img_noisy(self, x, first_sample, num, sigma, seed):
        """encode_img(x + sigma * eps_s) for the samples s = first_sample .. first_sample+num-1 of the noise stream, the noise
        fused into the patch-embed operand (cgpt_encode_img_noisy) -> inputs_llama [num, queries, proj_dim] float32."""
        x = self._check_x(x, False)
        out = torch.empty((num, self.cfg["qf_queries"], self.cfg["proj_dim"]), dtype=torch.float32, device=x.device)
        if num > 0:
            _lib.check(self._L.cgpt_encode_img_noisy(self._h, C.c_void_p(x.data_ptr()), first_sample, num, sigma, seed,
                                                     C.c_void_p(out.data_ptr()), _stream_ptr()))
        return out

    def forward_logits(self, x, first_sample, num, sigma, seed):
        x = self._check_x(x, False)
        out = torch.empty((num, self.num_classes), dtype=torch.float32, device=x.device)
        _lib.check(self._L.cgpt_forward_logits(self._h, C.c_void_p(x.data_ptr()), first_sample, num, sigma, seed,
                                               C.c_void_p(out.data_ptr()), _stream_ptr()))
        return out

    def activation(self, what, num):
        shape = {"vit_out": (num, self.tokens, self.cfg["vit_dim"]), "ln_vision": (num, self.tokens, self.cfg["vit_dim"]),
                 "qformer": (num, self.cfg["qf_queries"], self.cfg["qf_dim"]),
                 "llama": (num, self.cfg["qf_queries"], self.cfg["proj_dim"])}[what]
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        _lib.check(self._L.cgpt_get_activation(self._h, what.encode(), C.c_void_p(out.data_ptr()), out.numel(), _stream_ptr()))
        return out

    # ---- measurement hooks
    def profile(self, on=True):
        _lib.check(self._L.cgpt_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, kind=0):
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        _lib.check(self._L.cgpt_profile_read(self._h, kind, C.byref(ms), C.byref(fl), C.byref(n)))
        return ms.value, fl.value, n.value

    def profile_batches(self):
        """Samples of every classifier forward since profile(True), in order (cgpt_profile_batches): how the sample ranges were cut."""
        n = C.c_int64()
        _lib.check(self._L.cgpt_profile_batches(self._h, None, 0, C.byref(n)))
        buf = (C.c_int32 * max(1, n.value))()
        _lib.check(self._L.cgpt_profile_batches(self._h, buf, n.value, C.byref(n)))
        return [int(v) for v in buf[:n.value]]

    def profile_clock(self, kind=0):
        """GHz the profiled GEMMs of `kind` ran at (in-kernel s_memtime / s_memrealtime of every workgroup; cgpt_profile_clock).
        Read before profile_read(0), which resets the sums.  0.0 when nothing was profiled."""
        ghz = C.c_double()
        _lib.check(self._L.cgpt_profile_clock(self._h, kind, C.byref(ghz)))
        return ghz.value


def interpolate_pos_embed(pos_embed, num_patches):
    """Resize a checkpoint's position embedding [1, 1+P0, D] to this model's grid (reference interpolate_pos_embed,
    eva_vit.py:383-404: the class token is kept, the patch tokens are resized with bicubic interpolation,
    align_corners=False).  Host-side, at load time only."""
    pe = torch.as_tensor(np.asarray(pos_embed), dtype=torch.float32)
    pe = pe.reshape(1, -1, pe.shape[-1])
    D = pe.shape[-1]
    extra = 1
    orig = int((pe.shape[-2] - extra) ** 0.5)
    new = int(num_patches ** 0.5)
    if orig == new:
        return pe.numpy()
    tok = pe[:, extra:].reshape(-1, orig, orig, D).permute(0, 3, 1, 2)
    tok = torch.nn.functional.interpolate(tok, size=(new, new), mode="bicubic", align_corners=False)
    tok = tok.permute(0, 2, 3, 1).flatten(1, 2)
    return torch.cat((pe[:, :extra], tok), dim=1).numpy()


def noise_batch(x, first_sample, num, sigma, seed):
    """x[3,H,W] -> [num,3,H,W] noisy copies (smoothing.py:95-96) from the counter-based stream."""
    L = _lib.lib()
    x = x.contiguous()
    out = torch.empty((num,) + tuple(x.shape), dtype=torch.float32, device=x.device)
    _lib.check(L.cgpt_noise_batch(C.c_void_p(x.data_ptr()), x.numel(), first_sample, num, sigma, seed,
                                  C.c_void_p(out.data_ptr()), _stream_ptr()))
    return out


def vote(logits, counts):
    """counts[argmax(logits[b])] += 1 on the device (smoothing.py:97-98,101-105)."""
    L = _lib.lib()
    logits = logits.float().contiguous()
    _lib.check(L.cgpt_vote(C.c_void_p(logits.data_ptr()), logits.shape[0], logits.shape[1],
                           C.c_void_p(counts.data_ptr()), _stream_ptr()))
    return counts
