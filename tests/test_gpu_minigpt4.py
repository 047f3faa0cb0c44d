"""GPU tests of the full-MiniGPT-4 path (BASELINE configs[2] shape on tiny dimensions): cgpt_encode_img / cgpt_encode_img_noisy
(minigpt4.py:121-149 as a first-class C-ABI product) and `MiniGPT4Classifier` (minigpt_base.py:374-448) as the base classifier of
`Smooth`, with a random-init tiny LlamaForCausalLM on PyTorch-ROCm standing in for the frozen Vicuna-7B (no weights in the
container; nothing is downloaded) and token-id strings as answers."""
import ctypes as C

import numpy as np
import pytest
import torch

import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
from certifiedgpt_amd.minigpt4 import MiniGPT4Classifier, prepare_texts
from certifiedgpt_amd.agents import registry, setup_agent
from certifiedgpt_amd.agents import minigpt4_certify_agent  # noqa: F401
from certifiedgpt_amd.agents.label_adapter import AnswerLabelMap
from oracle import model_oracle as mo, smooth_oracle as so, generate_oracle as go
from gpu_util import DEV, tiny_pair, rel_err
from toy_llm import ToyTokenizer, tiny_llama

pytestmark = pytest.mark.gpu
PROMPT = prepare_texts(["<Img><ImageHere></Img> [vqa] what is shown here"])[0]


def test_encode_img_entry_points_match_the_forward_and_the_oracle():
    clf, p16, params, cfg = tiny_pair(mo.MODE_ENCODE_IMG, max_batch=4)
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    noisy = cg.noise_batch(x0, 3, 4, 0.5, 42)
    clf(noisy)
    want = clf.activation("llama", 4)                                   # inputs_llama of the logits forward
    emb, atts = clf.encode_img(noisy)
    assert emb.shape == (4, cfg.qf_queries, cfg.proj_dim) and torch.equal(emb, want)
    assert atts.shape == (4, cfg.qf_queries) and atts.dtype == torch.long and bool((atts == 1).all())   # minigpt4.py:148
    assert rel_err(emb, mo.forward_all(p16, noisy.cpu(), cfg)["llama"]) <= 1e-2
    assert torch.equal(clf.encode_img_noisy(x0, 3, 4, 0.5, 42), want)   # fused noise: same bits
    # any num: more samples than max_batch are cut into batches internally; results do not depend on the cut
    many = clf.encode_img_noisy(x0, 0, 11, 0.5, 42)
    assert torch.equal(many[3:7], want) and torch.isfinite(many).all()
    assert clf.encode_img_noisy(x0, 0, 0, 0.5, 42).shape[0] == 0
    # a CGPT_MODE_VIT_HEAD handle has no Q-Former: CGPT_ERR_STATE, not a crash
    vit, _, _, vcfg = tiny_pair(mo.MODE_VIT_HEAD, max_batch=4)
    out = torch.empty(4, 8, 192, device=DEV)
    rc = vit._L.cgpt_encode_img(vit._h, C.c_void_p(noisy.data_ptr()), 4, C.c_void_p(out.data_ptr()), None)
    assert rc == 5 and b"CGPT_MODE_ENCODE_IMG" in vit._L.cgpt_last_error()


@pytest.fixture(scope="module")
def minigpt4():
    enc, p16, params, cfg = tiny_pair(mo.MODE_ENCODE_IMG, max_batch=8)
    llm = tiny_llama(hidden=cfg.proj_dim, dtype=torch.float16, device=DEV)     # frozen fp16 decoder, base_model.py:201-219
    return enc, llm, cfg


def _answers_by_oracle(enc, llm, x0, first, num, sigma, seed):
    """Reference-shaped path: noisy images -> encode_img -> statement-by-statement generate restatement."""
    noisy = cg.noise_batch(x0, first, num, sigma, seed)
    emb, _ = enc.encode_img(noisy)
    return go.generate(llm, ToyTokenizer(), emb.to(torch.float16), [PROMPT] * num, max_new_tokens=5)


def test_generate_classifier_certify_matches_reference_shaped_path(minigpt4):
    enc, llm, cfg = minigpt4
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    sigma, seed, n0, n, alpha = 2.0, 7, 16, 24, 0.05          # a large sigma, so that the toy decoder's answers vary
    ref_answers = _answers_by_oracle(enc, llm, x0, 0, n0 + n, sigma, seed)
    vocab = sorted(set(ref_answers))[:5]
    K = len(vocab) + 1
    clf = MiniGPT4Classifier(enc, llm, ToyTokenizer(), PROMPT, AnswerLabelMap(K, vocab), max_new_tokens=5)
    print("answers:", sorted(set(ref_answers)))
    # one-hot logits of caller-supplied images == labels of the reference-shaped answers
    noisy = cg.noise_batch(x0, 0, 8, sigma, seed)
    lab = [vocab.index(a) if a in vocab else K - 1 for a in ref_answers]
    assert clf(noisy).argmax(1).tolist() == lab[:8]
    # Smooth through the engine path (noise fused into encode_img, vote in HIP)
    s = cg.Smooth(clf, K, sigma, seed=seed, non_certifiable=(clf.label_map.other_id,))
    c_sel = s._sample_noise(x0, n0, 8)
    c_est = s._sample_noise(x0, n, 5)                                   # ragged batches
    assert c_sel.tolist() == np.bincount(lab[:n0], minlength=K).tolist()
    assert c_est.tolist() == np.bincount(lab[n0:], minlength=K).tolist()
    s.reset()
    got = s.certify(x0, n0, n, alpha, 8)
    want = so.certify_from_counts(c_sel, c_est, n, alpha, sigma)
    if want[0] == clf.label_map.other_id:
        want = (cg.Smooth.ABSTAIN, 0.0)
    assert got[0] == want[0] and abs(got[1] - want[1]) <= 1e-9
    s.reset()
    p = s.predict(x0, n0, alpha, 8)
    wp = so.predict_from_counts(c_sel, alpha)
    assert p == (cg.Smooth.ABSTAIN if wp == clf.label_map.other_id else wp)


def test_certify_agent_drives_the_generating_classifier(minigpt4, tmp_path):
    """`image_text_certify` (the plugin the reference leaves empty) end to end over full MiniGPT-4: HipClassifier encoder built
    from the agent's config, decoder + tokenizer injected (a real run gives model.generate.llama_model = <local Vicuna dir>)."""
    enc, llm, cfg = minigpt4
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    vocab = sorted(set(_answers_by_oracle(enc, llm, x0, 0, 12, 2.0, 0)))[:4]
    dims = {k: getattr(cfg, k) for k in ("img_size", "patch_size", "vit_dim", "vit_depth", "vit_heads", "vit_mlp", "qf_layers",
                                         "qf_dim", "qf_heads", "qf_ffn", "qf_queries", "qf_xattn_freq", "proj_dim")}
    torch.save({"model": {k: v for k, v in mo.init_params(cfg, 20251121).items()}}, tmp_path / "ckpt.pth")
    conf = {"run": {"agent": "image_text_certify", "output_dir": str(tmp_path), "seed": 0,
                    "smoothing": {"sigma": 2.0, "n0": 8, "n": 16, "alpha": 0.05, "batch_size": 8, "num_classes": len(vocab) + 1,
                                  "radii": [0.0, 0.1]}},
            "model": {"dims": dims, "weights": str(tmp_path / "ckpt.pth"),
                      "generate": {"prompt": PROMPT, "answers": vocab, "max_new_tokens": 5}},
            "data": {"num_images": 2, "seed": 3}}
    registry.register("configuration", conf)
    agent = setup_agent(conf)
    agent.tokenizer, agent.llama_model = ToyTokenizer(), llm
    agent.run()
    res = agent.finalize()
    assert res["images"] == 2 and len(agent.records) == 2
    assert isinstance(agent.model, MiniGPT4Classifier) and agent.model.label_map.frozen
    for r in agent.records:
        assert r["predict"] == cg.Smooth.ABSTAIN or 0 <= r["predict"] < len(vocab)      # "other" is never certified
    lines = open(tmp_path / "certify.tsv").read().strip().splitlines()
    assert len(lines) == 3


def test_graph_decode_gives_the_answers_of_hf_generate(minigpt4):
    """decode="graph": prefill + every greedy step through the model's own forward, captured once per (batch, prompt length) into one
    hipGraph and replayed -- against `llama_model.generate` with the reference's arguments (decode="hf"), answer by answer, over full
    and ragged batches (each batch size captures its own graph), and the certify that rides on it."""
    enc, llm, cfg = minigpt4
    x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
    sigma, seed = 2.0, 7
    lm = AnswerLabelMap(4, ())
    hf = MiniGPT4Classifier(enc, llm, ToyTokenizer(), PROMPT, lm, max_new_tokens=5, decode="hf")
    gr = MiniGPT4Classifier(enc, llm, ToyTokenizer(), PROMPT, lm, max_new_tokens=5, decode="graph")
    for first, num in ((0, 8), (8, 8), (16, 5), (21, 8), (29, 1)):
        emb = enc.encode_img_noisy(x0, first, num, sigma, seed)
        assert gr.generate_from_embeds(emb, PROMPT) == hf.generate_from_embeds(emb, PROMPT), (first, num)
    assert gr.decode_stats["graph_captures"] == 3 and gr.decode_stats["graph_replays"] == 5 and gr.decode_stats["hf_calls"] == 0
    # ragged prompts are not the Monte-Carlo case: they take the HF path
    emb = enc.encode_img_noisy(x0, 0, 2, sigma, seed)
    texts = prepare_texts(["<ImageHere> short", "<ImageHere> a longer question here"])
    assert gr.generate_from_embeds(emb, texts) == hf.generate_from_embeds(emb, texts) and gr.decode_stats["hf_calls"] == 1
    # certify through the pair pass (selection + estimation draws in the same batches) on both decoders
    answers = hf.generate_from_embeds(enc.encode_img_noisy(x0, 0, 40, sigma, seed), PROMPT)
    vocab = sorted(set(answers))[:5]
    K = len(vocab) + 1
    outs = []
    for decode in ("hf", "graph"):
        clf = MiniGPT4Classifier(enc, llm, ToyTokenizer(), PROMPT, AnswerLabelMap(K, vocab), max_new_tokens=5, decode=decode)
        s = cg.Smooth(clf, K, sigma, seed=seed, non_certifiable=(clf.label_map.other_id,))
        outs.append((s.certify(x0, 16, 24, 0.05, 8), list(clf.last_answers)))
        s.reset()
        two = (s._sample_noise(x0, 16, 8).tolist(), s._sample_noise(x0, 24, 8).tolist())
        s.reset()
        pair = s._sample_noise_pair(x0, 16, 24, 8)
        assert (pair[0].tolist(), pair[1].tolist()) == two
    assert outs[0] == outs[1] and outs[0][1] == answers


def test_prefill_linears_through_the_library_gemm_match_torch():
    """prefill_linear="cgpt": the decoder's bias-free fp16 linears run calls of >= 1024 rows through cgpt_linear_f16 (the ViT's MFMA GEMM)
    inside the graph decode.  A 2-layer decoder with 256-aligned widths: the routed prefill's logits equal torch's to fp16 accumulation
    noise, calls below the row threshold and callers outside `enabled()` are untouched, and the answers of a 64-row batch agree."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from certifiedgpt_amd.minigpt4 import _LinearRoute
    cfg = LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, max_position_embeddings=256, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    torch.manual_seed(0)
    llm = LlamaForCausalLM(cfg).to(device=DEV, dtype=torch.float16).eval()
    assert _LinearRoute.install(llm, min_rows=1024) == 2 * 7 + 1           # q, k, v, o, gate, up, down per layer + lm_head
    x = torch.randn(64, 20, 256, device=DEV, dtype=torch.float16) * 0.5     # 1 280 rows
    with torch.no_grad():
        ref = llm(inputs_embeds=x, logits_to_keep=1).logits.float()
        assert torch.equal(llm(inputs_embeds=x, logits_to_keep=1).logits.float(), ref)           # not enabled: torch's path, same bits
        with _LinearRoute.enabled():
            got = llm(inputs_embeds=x, logits_to_keep=1).logits.float()
            small = llm(inputs_embeds=x[:8], logits_to_keep=1).logits.float()                    # 160 rows: below the threshold
        assert torch.equal(small, llm(inputs_embeds=x[:8], logits_to_keep=1).logits.float())
    assert rel_err(got, ref) <= 5e-3, rel_err(got, ref)

    class Enc:
        max_batch = 64
    from certifiedgpt_amd.minigpt4 import WordHashTokenizer
    tok = WordHashTokenizer(512)
    a = MiniGPT4Classifier(Enc(), llm, tok, PROMPT, AnswerLabelMap(4, ()), max_new_tokens=4, decode="graph")
    b = MiniGPT4Classifier(Enc(), llm, tok, PROMPT, AnswerLabelMap(4, ()), max_new_tokens=4, decode="graph", prefill_linear="cgpt")
    assert b.routed_linears == 15 and a.routed_linears == 0                # one patch per module, counted again; `a` never enables it
    emb = torch.randn(64, 8, 256, device=DEV) * 0.5
    # answers: a row may leave the torch path's trajectory only at a step whose top-2 margin (on the torch path) is within the
    # logit difference of the two paths; every decisive row gives the identical answer (the rule of the Vicuna-width test)
    from certifiedgpt_amd.minigpt4 import greedy_decode_parity, fp16_ulp
    segs = a._segment_embeddings(PROMPT, emb.device)
    embs = torch.cat([segs[0].expand(64, -1, -1), emb.half(), segs[1].expand(64, -1, -1)], dim=1)
    with torch.no_grad():
        t_ref, l_ref = a.greedy_tokens(embs, return_logits=True)
        with _LinearRoute.enabled():
            t_got, l_got = b.greedy_tokens(embs, return_logits=True)
    fin = torch.isfinite(l_ref[:, 0])
    d_first = float((l_got[:, 0][fin] - l_ref[:, 0][fin]).abs().max())
    ulp = fp16_ulp(float(l_ref[torch.isfinite(l_ref)].abs().max()))
    assert d_first <= 4 * ulp, (d_first, ulp)
    v = greedy_decode_parity(t_ref, l_ref, t_got, max(2.0 * d_first, 2.0 * ulp))
    ans_a, ans_b = a.generate_from_embeds(emb, PROMPT), b.generate_from_embeds(emb, PROMPT)
    print("routed prefill vs torch:", v, "; answers equal on", sum(int(p == q) for p, q in zip(ans_a, ans_b)), "of 64 rows")
    assert not v["violations"] and v["decisive_identical"] == v["decisive"] and v["diverged"] <= 64 - v["decisive"], v
