"""GPU parity tests proper: the HIP path through the C-ABI against (a) golden outputs of the reference's own
eva_vit.py / Qformer.py classes (tests/golden/model_golden.npz), (b) the fp32 CPU oracle on identical inputs and
weights, (c) size-independent properties at BASELINE.json's full ViT-G size.

Tolerances.  The reference's precision contract on a HIP device is fp16 weights + fp16 autocast
(eva_vit.py:407-414, base_model.py:141-142); this path uses fp16 MFMA operands with fp32 accumulation and an fp32
residual stream, the oracle is pure fp32.  Activations therefore agree to ~1e-2 relative (written per test);
the vote / abstain / radius logic given counts is bit-exact / 1e-9 (tests/test_capi_cpu.py)."""
import os

import numpy as np
import pytest
import torch

import certifiedgpt_amd as cg
from oracle import model_oracle as mo, smooth_oracle as so, philox
from conftest import GOLDEN
from gpu_util import DEV, tiny_pair, make_classifier, rel_err

pytestmark = pytest.mark.gpu


def test_weights_round_trip_and_names():
    clf, p16, params, cfg = tiny_pair(mo.MODE_ENCODE_IMG)
    assert clf.weight_names() == list(mo.param_shapes(cfg).keys())       # same registry, same order as the oracle
    for name in ("visual_encoder.blocks.1.attn.qkv.weight", "visual_encoder.blocks.0.attn.v_bias", "visual_encoder.pos_embed",
                 "Qformer.bert.encoder.layer.0.crossattention.self.value.weight",
                 "Qformer.bert.encoder.layer.1.attention.self.key.bias", "query_tokens", "head.weight"):
        got = clf.get_weight(name)
        assert np.array_equal(got, p16[name].numpy().reshape(-1)), name    # exactly fp16-rounded matrices / fp32 vectors
    import ctypes as C
    a = np.zeros(3, dtype=np.float32)
    assert clf._L.cgpt_load_weight(clf._h, b"head.bias", a.ctypes.data_as(C.c_void_p), 3) == 1      # numel mismatch
    assert clf._L.cgpt_load_weight(clf._h, b"no.such.weight", a.ctypes.data_as(C.c_void_p), 3) == 4  # not found
    assert b"no.such.weight" in clf._L.cgpt_last_error()


def test_tiny_model_matches_reference_goldens():
    """Golden vectors = outputs of the reference's own VisionTransformer / BertEncoder classes."""
    g = np.load(os.path.join(GOLDEN, "model_golden.npz"))
    clf, p16, params, cfg = tiny_pair(mo.MODE_ENCODE_IMG, seed=int(g["seed"]))
    x = torch.from_numpy(g["x"]).to(DEV)
    clf(x)
    n = x.shape[0]
    for what, tol in (("vit_out", 1e-2), ("ln_vision", 1e-2), ("qformer", 2e-2), ("llama", 2e-2)):
        e = rel_err(clf.activation(what, n), g[what])
        assert e <= tol, (what, e)


@pytest.mark.parametrize("mode", [mo.MODE_VIT_HEAD, mo.MODE_ENCODE_IMG])
def test_tiny_model_matches_oracle_all_stages(mode):
    clf, p16, params, cfg = tiny_pair(mode, max_batch=8)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    noisy = cg.noise_batch(x0, 0, 8, 0.5, 42)
    logits = clf(noisy)
    ref = mo.forward_all(p16, noisy.cpu(), cfg)
    stages = ["vit_out"] + (["ln_vision", "qformer", "llama"] if mode == mo.MODE_ENCODE_IMG else [])
    for what in stages:
        e = rel_err(clf.activation(what, 8), ref[what])
        assert e <= 1e-2, (what, e)
    e = rel_err(logits, ref["logits"])
    assert e <= 1e-2, ("logits", e)
    # fused noise+im2col path == noise_batch followed by the plain im2col path, bit for bit
    fused = clf.forward_logits(x0, 0, 8, 0.5, 42)
    assert torch.equal(fused, logits)


@pytest.mark.parametrize("mode", [mo.MODE_VIT_HEAD, mo.MODE_ENCODE_IMG])
def test_smooth_matches_oracle_on_identical_noise(mode):
    """Smooth.certify / predict on the GPU vs the line-for-line CPU restatement of the reference's Smooth fed with the
    SAME noisy inputs (the GPU's own draws, exported) and the same weights.  Votes must agree on every sample whose
    fp32 top-2 logit margin is decisive; label / abstain / radius given the counts are then exact."""
    K = 10
    clf, p16, params, cfg = tiny_pair(mode, num_classes=K, max_batch=16)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    sigma, seed, n0, n, alpha, bs = 0.25, 5, 24, 40, 0.05, 16

    def gpu_noise(first, num, shape):                       # N(0,1) draws exactly as the GPU made them
        nb = cg.noise_batch(torch.zeros(shape, device=DEV), first, num, 1.0, seed)
        return nb.cpu().numpy()

    oracle = so.SmoothOracle(mo.make_classifier(p16, cfg), K, sigma, gpu_noise)
    s = cg.Smooth(clf, K, sigma, seed=seed)
    # per-sample agreement
    noisy = cg.noise_batch(x0, 0, n0 + n, sigma, seed)
    ref_logits = mo.forward_all(p16, noisy.cpu(), cfg)["logits"]
    got_logits = torch.cat([clf(noisy[i:i + 16]) for i in range(0, n0 + n, 16)]).cpu()
    top2 = ref_logits.topk(2, dim=1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 2e-2 * ref_logits.abs().max()
    agree = got_logits.argmax(1) == ref_logits.argmax(1)
    assert bool(agree[decisive].all()), (int((~agree).sum()), int(decisive.sum()))
    assert float(agree.float().mean()) >= 0.9
    # counts: GPU hot loop vs CPU hot loop
    c_sel = s._sample_noise(x0, n0, bs)
    c_est = s._sample_noise(x0, n, bs)
    oracle._cursor = 0
    o_sel = oracle._sample_noise(x0.cpu().numpy(), n0, bs)
    o_est = oracle._sample_noise(x0.cpu().numpy(), n, bs)
    assert c_sel.sum() == n0 and c_est.sum() == n
    flips = int(np.abs(c_sel - o_sel).sum() + np.abs(c_est - o_est).sum()) // 2
    assert flips <= int((~decisive).sum()), (c_sel, o_sel, c_est, o_est)
    # statistics on identical counts: exact decision, radius to 1e-9 (north star: 1e-3)
    lab, rad = s.certify_from_counts(c_sel, c_est, n, alpha)
    olab, orad = so.certify_from_counts(c_sel, c_est, n, alpha, sigma)
    assert lab == olab and abs(rad - orad) <= 1e-9
    assert s.predict_from_counts(c_est, alpha) == so.predict_from_counts(c_est, alpha)
    # the public API end to end, reproducible after reset()
    s.reset()
    out1 = s.certify(x0, n0, n, alpha, bs)
    s.reset()
    out2 = s.certify(x0, n0, n, alpha, 7)                   # ragged last batch, different batch size: same counts
    assert out1 == out2 == (lab, rad)
    assert isinstance(out1[0], int) and isinstance(out1[1], float)
    p = s.predict(x0, n, alpha, bs)
    assert p == cg.Smooth.ABSTAIN or 0 <= p < K


def test_fused_pair_pass_is_bit_identical_to_two_passes():
    K = 10
    clf, p16, params, cfg = tiny_pair(mo.MODE_ENCODE_IMG, num_classes=K, max_batch=16)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    for (fa, na, fb, nb, bs) in [(0, 24, 24, 40, 16), (5, 7, 100, 9, 16), (0, 0, 3, 5, 4), (3, 5, 0, 0, 4), (0, 16, 16, 16, 16)]:
        pair = clf.sample_counts_pair(x0, fa, na, fb, nb, bs, 0.25, 5)
        a = clf.sample_counts(x0, fa, na, bs, 0.25, 5)
        b = clf.sample_counts(x0, fb, nb, bs, 0.25, 5)
        assert torch.equal(pair[0], a) and torch.equal(pair[1], b), (fa, na, fb, nb, bs)
        assert int(pair[0].sum()) == na and int(pair[1].sum()) == nb


def test_generic_module_path_uses_hip_noise_and_vote():
    """A base classifier that is an ordinary callable on PyTorch-ROCm (e.g. MiniGPT-4 + Vicuna): noise and vote are HIP."""
    K = 5
    w = torch.randn(K, 3 * 8 * 8, device=DEV)

    class Lin:
        def eval(self):
            return self

        def __call__(self, b):
            return b.flatten(1) @ w.t()

    x = torch.randn(3, 8, 8, device=DEV)
    s = cg.Smooth(Lin(), K, 0.5, seed=3)
    counts = s._sample_noise(x, 50, 16)
    noisy = cg.noise_batch(x, 0, 50, 0.5, 3)
    ref = torch.bincount((noisy.flatten(1) @ w.t()).argmax(1), minlength=K).cpu().numpy()
    assert counts.tolist() == ref.tolist()


# ------------------------------------------------------------------ BASELINE.json full size (ViT-G, 224x224)
@pytest.fixture(scope="module")
def vitg():
    cfg = mo.Config(mode=mo.MODE_VIT_HEAD, num_classes=1000)
    clf = make_classifier(cfg, max_batch=100)
    clf.init_synthetic(seed=0)
    yield clf, cfg
    clf.close()


def test_vitg_synthetic_init_follows_reference_law(vitg):
    clf, cfg = vitg
    w = clf.get_weight("visual_encoder.blocks.7.mlp.fc2.weight")
    assert abs(w.std() - 0.02 / np.sqrt(16.0)) < 2e-4 and abs(w.mean()) < 1e-4       # / sqrt(2*layer_id), eva_vit.py:308-314
    assert abs(clf.get_weight("visual_encoder.blocks.7.attn.qkv.weight").std() - 0.02) < 2e-4
    assert np.all(clf.get_weight("visual_encoder.blocks.3.norm1.weight") == 1) and np.all(clf.get_weight("visual_encoder.blocks.3.mlp.fc1.bias") == 0)
    # the device generator is the oracle's counter-based table (same tensor ids), up to fp32 transcendental rounding
    tid = list(mo.param_shapes(cfg).keys()).index("visual_encoder.blocks.0.attn.proj.weight")
    z = philox.normal_stream(0, tid, 4096, hi_word=1) * np.float32(0.02 / np.sqrt(2.0))
    got = clf.get_weight("visual_encoder.blocks.0.attn.proj.weight")[:4096]
    assert np.abs(got - z.astype(np.float16).astype(np.float32)).max() <= 2e-5


def test_vitg_counts_invariant_to_batching_and_sharding(vitg):
    """Size-independent property at BASELINE size (N=100, sigma=0.5): counts depend only on the global sample indices."""
    clf, cfg = vitg
    x = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    full = clf.sample_counts(x, 0, 100, 100, 0.5, 42)
    assert int(full.sum()) == 100
    b32 = clf.sample_counts(x, 0, 100, 32, 0.5, 42)                       # ragged batches 32,32,32,4
    assert torch.equal(full, b32)
    parts = torch.zeros_like(full)
    for r in range(8):                                                    # the 8-GPU partition 13,13,13,13,12,12,12,12
        lo, hi = cg.shard_range(100, r, 8)
        clf.sample_counts(x, lo, hi - lo, 13, 0.5, 42, counts=parts)
    assert torch.equal(full, parts)
    other = clf.sample_counts(x, 100, 100, 100, 0.5, 42)                  # fresh indices: a different draw
    assert int(other.sum()) == 100
    pair = clf.sample_counts_pair(x, 0, 100, 100, 100, 100, 0.5, 42)      # certify's fused pass, batches span the ranges
    assert torch.equal(pair[0], full) and torch.equal(pair[1], other)
    logits = clf.forward_logits(x, 0, 100, 0.5, 42)
    assert torch.isfinite(logits).all()
    assert torch.equal(torch.bincount(logits.argmax(1), minlength=1000), full)


def test_vitg_n1000_sharded_125_per_gpu_equals_one_pass_and_oracle_statistics():
    """BASELINE configs[3] at its real size (ViT-G + head, 224x224, sigma = 0.5): `_sample_noise(N = 1000)` as the eight
    125-sample shards an 8-GPU job runs (one 125-sample classifier batch per rank) gives the histogram of one pass bit for bit; and
    `Smooth.predict` (smoothing.py:58-79) / `Smooth.certify` (:29-56) at N = 1000 take the decisions of the CPU statistics oracle on
    those counts, on the host path and with the statistics finished on the device.  About 4 000 forwards, ~2.5 s of GPU."""
    cfg = mo.Config(mode=mo.MODE_VIT_HEAD, num_classes=1000)
    clf = make_classifier(cfg, max_batch=125)
    try:
        clf.init_synthetic(seed=0)
        x = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
        sigma, seed = 0.5, 42
        full = clf.sample_counts(x, 0, 1000, 125, sigma, seed)
        assert int(full.sum()) == 1000
        parts = torch.zeros_like(full)
        for r in range(8):
            lo, hi = cg.shard_range(1000, r, 8)
            assert hi - lo == 125
            clf.sample_counts(x, lo, hi - lo, 125, sigma, seed, counts=parts)
        assert torch.equal(full, parts)
        ragged = clf.sample_counts(x, 0, 1000, 99, sigma, seed)                 # 10 batches of 99 + one of 10
        assert torch.equal(full, ragged)
        est = clf.sample_counts(x, 1000, 1000, 125, sigma, seed)               # certify's estimation range
        # the mirrored 8-way partition of the estimation range (Smooth._sample_noise_pair) is a partition too
        parts_b = torch.zeros_like(est)
        for r in range(8):
            lo, hi = cg.shard_range(1000, r, 8, mirrored=True)
            clf.sample_counts(x, 1000 + lo, hi - lo, 125, sigma, seed, counts=parts_b)
        assert torch.equal(est, parts_b)
        c_sel, c_est = full.cpu().numpy(), est.cpu().numpy()
        for alpha in (0.001, 0.05):
            for dev_stats in (False, True):
                s = cg.Smooth(clf, 1000, sigma, seed=seed, device_stats=dev_stats)
                got = s.predict(x, 1000, alpha, 125)
                want = so.predict_from_counts(c_sel, alpha)
                assert got == want and ((got == cg.Smooth.ABSTAIN and type(got) is int) or isinstance(got, np.int64)), (alpha, dev_stats)
                s.reset()
                lab, rad = s.certify(x, 1000, 1000, alpha, 125)
                olab, orad = so.certify_from_counts(c_sel, c_est, 1000, alpha, sigma)
                assert lab == olab and abs(rad - orad) <= 1e-9, (alpha, dev_stats, (lab, rad), (olab, orad))
    finally:
        clf.close()


def test_vitg_profile_reports_times_flops_and_the_in_kernel_clock(vitg):
    """The measurement hooks behind bench.py's roofline: with profiling on, every GEMM launch is timed with HIP events on the launch
    stream and the 256-row kernels report the shader clock they ran at (s_memtime / s_memrealtime per workgroup).  The counts are
    unchanged by profiling; reading kind 0 drains the log and the clock sums."""
    clf, cfg = vitg
    x = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    plain = clf.sample_counts(x, 0, 100, 100, 0.5, 42)
    clf.profile_read(0)
    clf.profile(True)
    try:
        prof = clf.sample_counts(x, 0, 100, 100, 0.5, 42)
        ms, fl, n = clf.profile_read(1)                                      # fc1 + GELU: one launch per block
        assert n == 39 and abs(fl - 39 * 2.0 * 100 * 257 * 6144 * 1408) < 1e6 and 0 < ms < 1000
        ghz = {k: clf.profile_clock(k) for k in (0, 1, 2, 3, 4)}
        assert all(0.8 < v < 2.6 for v in ghz.values()), ghz              # MI355X: 2.4 GHz peak, lower under MFMA load
        tflops = fl / (ms * 1e-3) / 1e12
        assert tflops < 2500.0 * ghz[1] / 2.4 + 1e-9                         # no kernel beats 1 024 FLOP/clk/SIMD at the clock it ran at
        clf.profile_read(0)
        assert clf.profile_clock(0) == 0.0 and clf.profile_read(0)[2] == 0
    finally:
        clf.profile(False)
    assert torch.equal(plain, prof)


def test_vitg_image_sharded_certify_equals_sample_sharded_and_one_by_one(vitg):
    """SURVEY.md 8(e) names two partitions of the path: samples over ranks (`certify` / `certify_many` + the vote all-reduce) and
    whole images over ranks (`certify_images`, no vote collective).  At ViT-G size on one rank the three routes -- the one-by-one
    loop of the reference (smoothing.py:29-56 per image), the fused several-images pass, the image-sharded pass -- return the same
    (label, radius) list from the same cursor; and the image shards an 8-rank job would run (one image each, full 20 + 30 draws)
    reproduce their rows of it."""
    clf, cfg = vitg
    xs = torch.stack([torch.from_numpy(mo.synthetic_image(cfg, seed=1234 + i)).to(DEV) for i in range(4)])
    n0, n, alpha = 20, 30, 0.05
    s = cg.Smooth(clf, 1000, 0.5, seed=42)
    loop = [s.certify(xs[i], n0, n, alpha, 100) for i in range(4)]
    s.reset()
    many = s.certify_many(xs, n0, n, alpha, 100)
    s.reset()
    by_image = s.certify_images(xs, n0, n, alpha, 100)
    assert loop == many == by_image and s._next_sample == 4 * (n0 + n)
    for i in (3, 1):                                                      # a rank's shard: image i alone, at image i's cursor
        s.reset(i * (n0 + n))
        assert s.certify_images(xs[i:i + 1], n0, n, alpha, 100) == [loop[i]]
    s.reset(1000)
    one_by_one = [s.predict(xs[i], 40, alpha, 100) for i in range(4)]     # smoothing.py:58-79 per image
    s.reset(1000)
    together = s.predict_images(xs, 40, alpha, 100)
    assert together == one_by_one and [type(v) for v in together] == [type(v) for v in one_by_one]


def test_vitg_matches_oracle_on_two_samples(vitg):
    """Full-size numerical check: 2 noisy samples through ViT-G on the GPU vs the fp32 CPU oracle with the device's weights."""
    clf, cfg = vitg
    x = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    noisy = cg.noise_batch(x, 0, 2, 0.5, 42)
    logits = clf(noisy).cpu()
    vit_gpu = clf.activation("vit_out", 2).cpu()
    params = {n: torch.from_numpy(clf.get_weight(n)).reshape(s) for n, s in mo.param_shapes(cfg).items()}
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("CGPT_CPU_THREADS", "16"))))   # the box's CPU share
    ref = mo.forward_all(params, noisy.cpu(), cfg)
    e_vit = rel_err(vit_gpu, ref["vit_out"])
    e_log = rel_err(logits, ref["logits"])
    assert e_vit <= 2e-2 and e_log <= 2e-2, (e_vit, e_log)


def test_predict_and_n1000_sharded_125_per_gpu_match_single_pass():
    """BASELINE config 4 shape on the tiny model: N=1000 split 125 per rank over 8 ranks == one pass; Smooth.predict
    decisions on the GPU counts == the oracle's (smoothing.py:58-79)."""
    K = 10
    clf, p16, params, cfg = tiny_pair(mo.MODE_VIT_HEAD, num_classes=K, max_batch=125)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    full = clf.sample_counts(x0, 0, 1000, 125, 0.25, 9)
    parts = torch.zeros_like(full)
    for r in range(8):
        lo, hi = cg.shard_range(1000, r, 8)
        assert hi - lo == 125
        clf.sample_counts(x0, lo, hi - lo, 125, 0.25, 9, counts=parts)
    assert torch.equal(full, parts) and int(full.sum()) == 1000
    s = cg.Smooth(clf, K, 0.25, seed=9)
    for alpha in (0.001, 0.05, 0.5):
        s.reset()
        got = s.predict(x0, 1000, alpha, 125)
        assert got == so.predict_from_counts(full.cpu().numpy(), alpha)
        # return types of smoothing.py:77,79: the int constant ABSTAIN, else an element of an int64 ndarray
        assert (got == cg.Smooth.ABSTAIN and type(got) is int) or isinstance(got, np.int64)
        sd = cg.Smooth(clf, K, 0.25, seed=9, device_stats=True)          # binomial test on the GPU: same decision
        got_dev = sd.predict(x0, 1000, alpha, 125)
        assert got_dev == got and type(got_dev) is type(got)
    # certify at N=1000 through the fused pass == oracle statistics on the same counts
    s.reset()
    lab, rad = s.certify(x0, 1000, 1000, 0.001, 125)
    est = clf.sample_counts(x0, 1000, 1000, 125, 0.25, 9)
    olab, orad = so.certify_from_counts(full.cpu().numpy(), est.cpu().numpy(), 1000, 0.001, 0.25)
    assert lab == olab and abs(rad - orad) <= 1e-9


def test_edge_cases_and_errors():
    import ctypes as C
    K = 10
    clf, p16, params, cfg = tiny_pair(mo.MODE_VIT_HEAD, num_classes=K, max_batch=4)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    assert int(clf.sample_counts(x0, 0, 0, 4, 0.5, 1).sum()) == 0                      # empty range
    one = clf.sample_counts(x0, 7, 1, 4, 0.5, 1)                                       # single sample
    assert int(one.sum()) == 1
    a = clf.sample_counts(x0, 0, 11, 1, 0.5, 1)                                        # batch_size 1, ragged count
    b = clf.sample_counts(x0, 0, 11, 4, 0.5, 1)
    assert torch.equal(a, b)
    big = clf.sample_counts(x0, 0, 11, 1000, 0.5, 1)                                   # batch_size > max_batch is clamped by the binding
    assert torch.equal(big, b)
    counts = torch.zeros(K, dtype=torch.int64, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert clf._L.cgpt_sample_counts(clf._h, C.c_void_p(x0.data_ptr()), 0, 4, 5, 0.5, 1, C.c_void_p(counts.data_ptr()), st) == 1   # > max_batch
    assert clf._L.cgpt_sample_counts(clf._h, C.c_void_p(x0.data_ptr()), 0, -1, 4, 0.5, 1, C.c_void_p(counts.data_ptr()), st) == 1  # negative count
    assert clf._L.cgpt_sample_counts(clf._h, None, 0, 4, 4, 0.5, 1, C.c_void_p(counts.data_ptr()), st) == 1                         # null image
    with pytest.raises(ValueError):
        clf.sample_counts(torch.zeros(3, 8, 8, device=DEV), 0, 1, 1, 0.5, 1)             # wrong image shape
    with pytest.raises(TypeError):
        clf.sample_counts(x0.cpu(), 0, 1, 1, 0.5, 1)                                     # host tensor: no CPU path
    # sigma = 0: every sample sees the clean image -> unanimous vote
    z = clf.sample_counts(x0, 0, 9, 4, 0.0, 1)
    assert int(z.max()) == 9
    clean = clf(x0[None])
    assert int(z.argmax()) == int(clean.argmax())


def test_larger_image_streaming_attention_and_pos_embed_interpolation():
    """Section 8(f) rank 3: images whose token count exceeds the in-LDS attention limit (280x280 -> T = 401 here;
    448x448 -> 1025 in the reference, minigpt4.py:32) and a checkpoint pos_embed from a smaller grid
    (interpolate_pos_embed, eva_vit.py:383-404)."""
    import dataclasses
    cfg = dataclasses.replace(mo.tiny_config(mode=mo.MODE_ENCODE_IMG, num_classes=10), img_size=280)
    assert cfg.tokens == 401
    params = mo.init_params(cfg, 5)
    # pretend the checkpoint was trained at 56x56 (4x4 grid): hand the small pos_embed to the loader
    small = mo.init_params(mo.tiny_config(mode=mo.MODE_ENCODE_IMG, num_classes=10), 5)["visual_encoder.pos_embed"]
    state = dict(params)
    state["visual_encoder.pos_embed"] = small
    clf = make_classifier(cfg, max_batch=3)
    clf.load_state_dict(state)
    params["visual_encoder.pos_embed"] = mo.interpolate_pos_embed(small, cfg.tokens - 1)
    assert np.abs(clf.get_weight("visual_encoder.pos_embed") - params["visual_encoder.pos_embed"].numpy().reshape(-1)).max() <= 1e-6
    p16 = mo.round_fp16_weights(params)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    noisy = cg.noise_batch(x0, 0, 3, 0.5, 42)
    logits = clf(noisy)
    ref = mo.forward_all(p16, noisy.cpu(), cfg)
    for what in ("vit_out", "ln_vision", "qformer", "llama"):
        e = rel_err(clf.activation(what, 3), ref[what])
        assert e <= 1e-2, (what, e)
    assert rel_err(logits, ref["logits"]) <= 1e-2
    assert torch.equal(clf.forward_logits(x0, 0, 3, 0.5, 42), logits)


def test_device_side_statistics_match_reference_goldens(stats_golden):
    """cgpt_certify_device / cgpt_predict_device (one wavefront: shuffle arg-max, float64 Clopper-Pearson / binomial test /
    Phi^-1 on the GPU) against the goldens emitted by the reference's own smoothing.py: decisions exact, radius <= 1e-9."""
    s = cg.Smooth(None, 2, 1.0)
    worst = 0.0
    for c in stats_golden["certify"][::2]:
        s.sigma = c["sigma"]
        sel = torch.tensor(c["counts_sel"], dtype=torch.int64, device=DEV)
        est = torch.tensor(c["counts_est"], dtype=torch.int64, device=DEV)
        lab, rad = s._finalize_device(sel, est, c["n"], c["alpha"], predict=False)
        assert lab == c["label"], c
        worst = max(worst, abs(rad - c["radius"]))
    assert worst <= 1e-9, worst
    for c in stats_golden["predict"][::2]:
        cnt = torch.tensor(c["counts"], dtype=torch.int64, device=DEV)
        assert s._finalize_device(None, cnt, 1, c["alpha"], predict=True) == c["label"], c
    # end to end: certify with device_stats == certify with host statistics
    K = 10
    clf, p16, params, cfg = tiny_pair(mo.MODE_VIT_HEAD, num_classes=K, max_batch=16)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    a = cg.Smooth(clf, K, 0.25, seed=5).certify(x0, 24, 40, 0.05, 16)
    b = cg.Smooth(clf, K, 0.25, seed=5, device_stats=True).certify(x0, 24, 40, 0.05, 16)
    assert a[0] == b[0] and abs(a[1] - b[1]) <= 1e-12


# ------------------------------------------------------------------------------------------------ RGF attack
def test_rgf_step_kernel_matches_oracle_bit_for_bit():
    """cgpt_rgf_step vs oracle/rgf_oracle.rgf_step on the directions the device itself generates (exported through
    cgpt_noise_batch with x = 0, sigma = 1), and the direction stream vs the oracle's Philox restatement."""
    from oracle import rgf_oracle as ro
    shape = (3, 28, 28)
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(shape, generator=g)
    x_adv = x + 0.1 * torch.randn(shape, generator=g)
    for first, q, lr, eps in [(0, 1, 0.05, 0.25), (7, 3, -0.02, 0.08), (100, 32, 0.3, 0.2), (5, 2, 0.05, 0.0)]:
        coeffs = (torch.randn(q, generator=g) * 2).numpy().astype(np.float32)
        if q == 3:
            coeffs[1] = 0.0
        got = cg.rgf_step(x_adv.to(DEV), x.to(DEV), first, coeffs, lr, eps, 77).cpu().numpy()
        dirs = cg.noise_batch(torch.zeros(shape, device=DEV), first, q, 1.0, 77).cpu().numpy()
        ref = ro.rgf_step(x_adv.numpy(), x.numpy(), dirs, coeffs, lr, eps)
        assert np.array_equal(got, ref), (first, q, float(np.abs(got - ref).max()))
        assert float(np.abs(got - x.numpy()).max()) <= eps + 1e-7
        # the oracle's own direction stream is the same stream (float32 Box-Muller: compared to 2e-6)
        od = np.stack([ro.direction(77, first + i, shape) for i in range(q)])
        assert float(np.abs(od - dirs).max()) <= 2e-6
    # all-zero coefficients: sign(0) = 0, the image only gets clamped
    same = cg.rgf_step(x_adv.to(DEV), x.to(DEV), 0, np.zeros(2, np.float32), 0.5, 10.0, 77).cpu().numpy()
    assert np.array_equal(same, x_adv.numpy())


def test_rgf_attack_loop_matches_oracle_loop_and_is_reproducible():
    """RGFAttack (HIP noise / classifier / vote / update kernels + the host schedule) against oracle/rgf_oracle.attack, the
    numpy restatement of the schedule, with the vote shares of both coming from the GPU Smooth (so the comparison isolates the
    loop and the update rule; the classifier itself is covered by the tests above)."""
    from oracle import rgf_oracle as ro
    K = 10
    clf, p16, params, cfg = tiny_pair(mo.MODE_VIT_HEAD, num_classes=K, max_batch=16)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    sigma, seed, n, alpha, bs = 0.25, 5, 32, 0.05, 16
    s = cg.Smooth(clf, K, sigma, seed=seed)
    base_counts = s._sample_noise(x0, n, bs)
    target = int(np.argsort(base_counts)[-2]) if (base_counts > 0).sum() > 1 else (int(base_counts.argmax()) + 1) % K
    steps, q, delta, lr, eps, dseed = 4, 2, 0.5, 0.05, 0.2, 99

    def run():
        s.reset(1000)
        atk = cg.RGFAttack(s, steps=steps, num_dirs=q, delta=delta, lr=lr, eps=eps, dir_seed=dseed)
        return atk.attack(x0, target, n, alpha, bs, targeted=True)

    x_adv, label, hist = run()
    x_adv2, label2, hist2 = run()
    assert torch.equal(x_adv, x_adv2) and label == label2 and hist == hist2          # reproducible bit for bit
    assert len(hist) == steps + 1 and all(0.0 <= h <= 1.0 for h in hist)
    assert float((x_adv - x0).abs().max()) <= eps + 1e-6
    assert label == cg.Smooth.ABSTAIN or 0 <= label < K

    def share_fn(img, step):                                  # same sample indices per step as RGFAttack
        s.reset(1000 + step * n)
        c = s._sample_noise(torch.from_numpy(np.ascontiguousarray(img)).to(DEV), n, bs)
        return float(c[target]) / n

    def direction_fn(i):                                      # the device's own draws
        return cg.noise_batch(torch.zeros_like(x0), i, 1, 1.0, dseed)[0].cpu().numpy()

    o_adv, o_hist = ro.attack(share_fn, direction_fn, x0.cpu().numpy(), steps, q, delta, lr, eps, targeted=True)
    assert o_hist == hist, (o_hist, hist)
    assert np.array_equal(o_adv, x_adv.cpu().numpy())
    assert cg.RGFAttack(s, steps=8, num_dirs=1).forwards_per_image(100) == 1700


def test_attack_eval_agent_runs_the_rgf_scenario(tmp_path):
    """Plugin contract (launch.py:97-107) of the attack-evaluation agent on the tiny model: BASELINE configs[4] in miniature."""
    from certifiedgpt_amd.agents import registry, setup_agent
    from certifiedgpt_amd.agents import minigpt4_attack_agent  # noqa: F401  (import registers)
    K = 10
    clf, p16, params, cfg = tiny_pair(mo.MODE_VIT_HEAD, num_classes=K, max_batch=16)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    conf = {"run": {"agent": "image_text_attack_eval", "output_dir": str(tmp_path), "seed": 3,
                    "smoothing": {"sigma": 0.25, "n": 16, "alpha": 0.05, "batch_size": 16, "num_classes": K},
                    "attack": {"steps": 3, "num_dirs": 2, "delta": 0.5, "lr": 0.05, "eps": 0.2, "seed": 9}}}
    registry.register("configuration", conf)
    agent = setup_agent(conf)
    agent.classifier = clf
    agent.dataset = [(x0, 1), (x0 * 0.5, 4)]
    agent.run()
    res = agent.finalize()
    assert res["images"] == 2 and 0.0 <= res["attack_success_rate"] <= 1.0
    assert res["forwards_per_image"] == (3 * 3 + 1) * 16 + 16
    lines = open(tmp_path / "attack_eval.tsv").read().strip().splitlines()
    assert len(lines) == 3 and lines[0].split("\t")[:3] == ["idx", "label", "target"]
    assert lines[1].split("\t")[2] == "2" and lines[2].split("\t")[2] == "5"


def test_certify_many_is_bit_identical_to_consecutive_certify_calls():
    """Several images per fused pass (cgpt_sample_counts_images): same sample indices, same counts, same decisions as the
    per-image calls -- for full groups, a ragged last group, and a per-image share larger than the batch capacity."""
    K = 10
    clf, p16, params, cfg = tiny_pair(mo.MODE_ENCODE_IMG, num_classes=K, max_batch=16)
    g = torch.Generator(device="cpu").manual_seed(11)
    xs = torch.stack([torch.from_numpy(mo.synthetic_image(cfg)) * (0.5 + 0.25 * i) + 0.1 * torch.randn(3, cfg.img_size, cfg.img_size, generator=g)
                      for i in range(5)]).to(DEV)
    for (n0, n) in [(3, 4), (5, 3), (2, 6), (12, 20)]:            # 2 / 2 / 2 images per batch of 16; last: one image > capacity
        s1 = cg.Smooth(clf, K, 0.25, seed=5)
        s2 = cg.Smooth(clf, K, 0.25, seed=5)
        seq = [s1.certify(xs[i], n0, n, 0.05, 16) for i in range(5)]
        many = s2.certify_many(xs, n0, n, 0.05, 16)
        assert many == seq, (n0, n, many, seq)
        assert s1._next_sample == s2._next_sample == 5 * (n0 + n)
    # raw counts: [G, 2, K] table vs the pair pass per image
    tab = clf.sample_counts_images(xs, 7, 3, 100, 4, 11, 0.25, 5)
    for i in range(5):
        pair = clf.sample_counts_pair(xs[i], 7 + 11 * i, 3, 100 + 11 * i, 4, 16, 0.25, 5)
        assert torch.equal(tab[i], pair), i
    assert int(tab.sum()) == 5 * 7


def test_sample_counts_can_be_captured_in_a_hip_graph():
    """Nothing is allocated, freed or synchronised inside cgpt_sample_counts* and every launch goes to the caller's stream
    (include/cgpt.h), so a caller may capture a whole `_sample_noise` into a hipGraph (here through torch.cuda.CUDAGraph) and
    replay it: the replay re-draws the SAME sample indices on whatever image the captured buffer then holds."""
    K = 10
    clf, p16, params, cfg = tiny_pair(mo.MODE_ENCODE_IMG, num_classes=K, max_batch=16)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    x_static = x0.clone()
    want = clf.sample_counts(x_static, 3, 40, 16, 0.25, 5).clone()       # eager (also warms every per-device launch cache)
    torch.cuda.synchronize()
    counts = torch.zeros(K, dtype=torch.int64, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            clf.sample_counts(x_static, 3, 40, 16, 0.25, 5, counts=counts)
    torch.cuda.current_stream().wait_stream(side)
    counts.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(counts, want)
    g.replay()                                                            # votes are ADDED into the histogram
    torch.cuda.synchronize()
    assert torch.equal(counts, 2 * want)
    x_static.copy_(x0 * 0.5 + 0.1)                                        # another image in the captured buffer
    counts.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(counts, clf.sample_counts(x_static, 3, 40, 16, 0.25, 5))
