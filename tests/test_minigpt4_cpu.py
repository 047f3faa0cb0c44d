"""Host logic of the MiniGPT-4 `generate` base classifier (certifiedgpt_amd/minigpt4.py) on the CPU: prompt splice, left
padding, greedy decode through a random-init tiny LlamaForCausalLM, decode clean-up, answer -> label -- against
tests/golden/generate_golden.{json,npz}, which oracle/gen_golden_generate.py produced by EXECUTING the reference's own
`MiniGPTBase.generate` / `get_context_emb` (minigpt_base.py:75-89,374-448) and `CONV_VISION_minigptv2` prompt template, and
against the statement-by-statement restatement oracle/generate_oracle.py (itself pinned by the same fixture).  The encoder is a stub
(encode_img needs the GPU); the GPU test tests/test_gpu_minigpt4.py runs the same classifier over cgpt_encode_img."""
import json
import os

import numpy as np
import pytest
import torch

from certifiedgpt_amd.minigpt4 import MiniGPT4Classifier, prepare_texts, clean_answer
from certifiedgpt_amd.agents.label_adapter import AnswerLabelMap
from oracle import generate_oracle as go
from toy_llm import ToyTokenizer, StubEncoder, tiny_llama

GOLD = os.path.join(os.path.dirname(__file__), "golden", "generate_golden")


@pytest.fixture(scope="module")
def golden():
    meta = json.load(open(GOLD + ".json"))
    arrays = np.load(GOLD + ".npz")
    llm = tiny_llama(hidden=meta["llama"]["hidden"], seed=123)          # the seed is irrelevant: the fixture carries the weights
    llm.load_state_dict({k[len("llama."):]: torch.from_numpy(arrays[k]) for k in arrays.files if k.startswith("llama.")})
    return meta, arrays, llm.eval()


def _gold_classifier(llm, max_new_tokens):
    return MiniGPT4Classifier(StubEncoder(), llm, ToyTokenizer(), "<ImageHere>", AnswerLabelMap(4, ()), max_new_tokens=max_new_tokens)


def test_reference_generate_goldens_pin_the_oracle_and_the_product(golden):
    """Every case the reference's own MiniGPTBase.generate produced (shared, ragged and single prompts; 6 and 20 new tokens)."""
    meta, arrays, llm = golden
    n = 0
    for case in meta["cases"]:
        if "max_new_tokens" not in case:
            continue
        emb = torch.from_numpy(arrays[case["embeds"]])
        assert prepare_texts(case["questions"]) == case["texts"]                       # conversation.py:130-137 by execution
        got_oracle = go.generate(llm, ToyTokenizer(), emb, case["texts"], max_new_tokens=case["max_new_tokens"])
        assert got_oracle == case["answers"], case["name"]
        clf = _gold_classifier(llm, case["max_new_tokens"])
        assert clf.generate_from_embeds(emb, case["texts"]) == case["answers"], case["name"]
        if len(set(case["texts"])) == 1:
            assert clf.generate_from_embeds(emb, case["texts"][0]) == case["answers"]  # ONE str shared by all rows
        n += 1
    assert n == 6


def test_graph_decode_loop_reproduces_the_reference_generate_goldens(golden):
    """`greedy_tokens` -- the fixed-length greedy loop that decode="graph" captures into one hipGraph (prefill + every step through the
    model's own forward, EOS suppressed on the first token, pad after EOS) -- gives the answers the reference's
    `MiniGPTBase.generate` gave for the shared-prompt cases (a CPU run of the same loop; the capture itself is a GPU test)."""
    meta, arrays, llm = golden
    n = 0
    for case in meta["cases"]:
        if "max_new_tokens" not in case or len(set(case["texts"])) != 1:
            continue
        clf = _gold_classifier(llm, case["max_new_tokens"])
        emb = torch.from_numpy(arrays[case["embeds"]])
        segs = clf._segment_embeddings(case["texts"][0], emb.device)
        embs = torch.cat([segs[0].expand(len(emb), -1, -1), emb, segs[1].expand(len(emb), -1, -1)], dim=1)
        with torch.no_grad():
            toks = clf.greedy_tokens(embs)
        assert toks.shape == (len(emb), case["max_new_tokens"])
        assert clf._decode_outputs(toks) == case["answers"], case["name"]
        n += 1
    assert n == 4


def test_pair_pass_equals_two_passes():
    """`sample_counts_pair` (selection + estimation draws of one certify in the same batches) is host logic around the encoder, the
    decoder and the vote; with a CPU stub encoder it must split a contiguous index range exactly like two `sample_counts` calls."""
    import certifiedgpt_amd.minigpt4 as m

    class Enc(StubEncoder):
        max_batch = 8

        def encode_img_noisy(self, x, first, num, sigma, seed):
            g = torch.Generator().manual_seed(1000)
            table = torch.randn(64, 3, generator=g)
            imgs = x[None] + sigma * table[first:first + num, :, None, None]
            return self.encode_img(imgs)[0]

    def cpu_vote(logits, counts):
        counts += torch.bincount(logits.argmax(1), minlength=counts.numel())

    old, m.vote = m.vote, cpu_vote
    try:
        llm = tiny_llama()
        probe = MiniGPT4Classifier(Enc(), llm, ToyTokenizer(), PROMPT, AnswerLabelMap(6, ()), max_new_tokens=3)
        x = torch.randn(3, 8, 8)
        answers = probe.generate_from_embeds(probe.encoder.encode_img_noisy(x, 0, 40, 2.0, 0), PROMPT)
        vocab = sorted(set(answers))[:5]
        clf = MiniGPT4Classifier(Enc(), llm, ToyTokenizer(), PROMPT, AnswerLabelMap(6, vocab), max_new_tokens=3)
        for na, nb, bs in ((16, 24, 7), (5, 3, 8), (0, 9, 4), (9, 0, 4)):
            pair = clf.sample_counts_pair(x, 0, na, na, nb, bs, 2.0, 0)
            assert len(clf.last_answers) == na + nb
            a = clf.sample_counts(x, 0, na, bs, 2.0, 0)
            b = clf.sample_counts(x, na, nb, bs, 2.0, 0)
            assert pair[0].tolist() == a.tolist() and pair[1].tolist() == b.tolist(), (na, nb, bs)
            assert int(pair.sum()) == na + nb
        gap = clf.sample_counts_pair(x, 0, 4, 10, 4, 8, 2.0, 0)                 # not contiguous: two passes
        assert gap[0].tolist() == clf.sample_counts(x, 0, 4, 8, 2.0, 0).tolist()
    finally:
        m.vote = old


def test_reference_context_embedding_goldens(golden):
    meta, arrays, llm = golden
    clf = _gold_classifier(llm, 6)
    for name in ("shared", "ragged", "single"):
        case = next(c for c in meta["cases"] if c["name"] == f"{name}_mnt6")
        emb = torch.from_numpy(arrays[f"emb.{name}"])
        want = torch.from_numpy(arrays[f"ctx.{name}"])
        with torch.no_grad():
            assert torch.equal(clf.get_context_emb(case["texts"][0], [emb[0][None]]), want)
            assert torch.equal(go.get_context_emb(llm.get_input_embeddings(), ToyTokenizer(), case["texts"][0], [emb[0][None]]), want)


def test_reference_decode_cleanup_goldens(golden):
    """minigpt_base.py:441-447 driven by the reference on prescribed token rows and prescribed decoded strings."""
    meta, arrays, llm = golden
    rows = next(c for c in meta["cases"] if c["name"] == "scripted_cleanup")
    strs = next(c for c in meta["cases"] if c["name"] == "scripted_strings")
    assert [clean_answer(s) for s in strs["decoded"]] == strs["answers"]

    class Scripted:
        def __init__(self, inner, out):
            self.inner, self.out = inner, out

        def get_input_embeddings(self):
            return self.inner.get_input_embeddings()

        def parameters(self):
            return self.inner.parameters()

        def generate(self, **kw):
            return torch.tensor(self.out, dtype=torch.long)

    emb = torch.from_numpy(arrays["emb.scripted"])
    clf = MiniGPT4Classifier(StubEncoder(), Scripted(llm, rows["rows"]), ToyTokenizer(), "<ImageHere>", AnswerLabelMap(4, ()))
    assert clf.generate_from_embeds(emb, rows["texts"]) == rows["answers"]
    assert go.generate(Scripted(llm, rows["rows"]), ToyTokenizer(), emb, rows["texts"]) == rows["answers"]


PROMPT = prepare_texts(["<Img><ImageHere></Img> [vqa] what is shown here"])[0]


def _classifier(num_classes=8, vocabulary=(), llm=None, generate_kwargs=None):
    return MiniGPT4Classifier(StubEncoder(), llm if llm is not None else tiny_llama(), ToyTokenizer(), PROMPT,
                              AnswerLabelMap(num_classes, vocabulary), max_new_tokens=6, generate_kwargs=generate_kwargs)


def test_prompt_template_and_cleanup():
    assert PROMPT == "<s>[INST] <Img><ImageHere></Img> [vqa] what is shown here [/INST]"     # conversation.py:130-137
    assert clean_answer("<s> [INST] q [/INST] a cat </s> junk") == "a cat"                   # minigpt_base.py:444-447
    assert clean_answer("plain") == "plain"


def test_shared_prompt_path_equals_reference_shaped_generate():
    clf = _classifier()
    images = torch.randn(5, 3, 8, 8)
    emb, atts = clf.encoder.encode_img(images)
    assert atts.shape == (5, 4) and bool((atts == 1).all())
    ref = go.generate(clf.llama_model, clf.llama_tokenizer, emb, [PROMPT] * 5, max_new_tokens=6)
    assert clf.generate_from_embeds(emb, PROMPT) == ref                 # segments embedded once and broadcast
    assert clf.generate_from_embeds(emb, [PROMPT] * 5) == ref
    assert clf.generate(images, PROMPT) == ref
    assert all(isinstance(a, str) for a in ref) and len(set(ref)) >= 1


def test_ragged_prompts_left_padding_equals_reference_shaped_generate():
    clf = _classifier()
    images = torch.randn(3, 3, 8, 8)
    emb, _ = clf.encoder.encode_img(images)
    texts = prepare_texts(["<ImageHere> short", "<Img><ImageHere></Img> a much longer question about the picture", "<ImageHere> mid size one"])
    ref = go.generate(clf.llama_model, clf.llama_tokenizer, emb, texts, max_new_tokens=6)
    assert clf.generate_from_embeds(emb, texts) == ref
    # context embedding of one sample: bos only on the first segment, image tokens spliced at the placeholder
    ctx = clf.get_context_emb(texts[0], [emb[0][None]])
    tok = ToyTokenizer()
    n0 = tok("<s>[INST]", add_special_tokens=True).input_ids.shape[1]
    n1 = tok("short [/INST]", add_special_tokens=False).input_ids.shape[1]
    assert ctx.shape == (1, n0 + 4 + n1, 64)
    assert torch.equal(ctx[0, n0:n0 + 4], emb[0])


def test_one_hot_logits_follow_the_frozen_vocabulary():
    clf = _classifier()
    images = torch.randn(6, 3, 8, 8)
    answers = clf.generate(images, PROMPT)
    vocab = sorted(set(answers))[:3]
    clf2 = _classifier(num_classes=5, vocabulary=vocab)
    assert clf2.label_map.frozen
    logits = clf2(images)
    assert logits.shape == (6, 5) and bool((logits.sum(1) == 1).all())
    want = [vocab.index(a) if a in vocab else clf2.label_map.other_id for a in answers]
    assert logits.argmax(1).tolist() == want
    assert clf2.last_answers == answers[-len(clf2.last_answers):]


def _eos_first_llm():
    """A tiny decoder whose very first generated token would be EOS (and whose later ones are not): lm_head row of EOS boosted along
    the direction of the prompt's last hidden state."""
    from toy_llm import EOS
    llm = tiny_llama(hidden=64, seed=7)
    clf = _classifier(llm=llm)
    images = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(3))
    emb, _ = clf.encoder.encode_img(images)
    segs = clf._segment_embeddings(PROMPT, emb.device)
    embs = torch.cat([segs[0].expand(4, -1, -1), emb, segs[1].expand(4, -1, -1)], dim=1)
    with torch.no_grad():
        h = llm.model(inputs_embeds=embs).last_hidden_state[:, -1, :]                 # [4, hidden]: what the first logits are taken from
        llm.lm_head.weight[EOS] += 50.0 * h.mean(0) / h.mean(0).norm()
        first = llm(inputs_embeds=embs).logits[:, -1, :].argmax(-1)
    assert bool((first == EOS).all())                                                 # without suppression every answer would be empty
    return llm, emb, embs


def test_first_token_is_never_eos_on_either_decode_path_whatever_transformers_is_installed():
    """the reference's `min_length = 1` (minigpt_base.py:385) means "at least one GENERATED token" under its pinned
    transformers 4.30.0; the installed 5.15 would turn it into 0 for an inputs_embeds call.  Both decode paths ship the reference's
    semantics: HF `generate` gets it as min_new_tokens, the greedy loop masks EOS on the first step -- same tokens, no empty answer."""
    from toy_llm import EOS
    llm, emb, embs = _eos_first_llm()
    clf = _classifier(llm=llm)
    kw = clf.hf_generate_kwargs()
    assert kw.get("min_new_tokens") == 1 and kw["min_length"] == 0 and clf.min_new_tokens() == 1
    with torch.no_grad():
        hf = llm.generate(inputs_embeds=embs, attention_mask=torch.ones(embs.shape[:2], dtype=torch.int), max_new_tokens=6, **kw)
        loop = clf.greedy_tokens(embs)
    assert bool((hf[:, 0] != EOS).all()) and bool((loop[:, 0] != EOS).all())
    assert torch.equal(loop[:, :hf.shape[1]], hf) or clf._decode_outputs(loop) == clf._decode_outputs(hf)
    assert clf.generate_from_embeds(emb, PROMPT) == go.generate(llm, clf.llama_tokenizer, emb, [PROMPT] * 4, max_new_tokens=6)
    # min_length = 0 switches the suppression off on both paths (then the first token IS EOS)
    clf0 = _classifier(llm=llm, generate_kwargs={"min_length": 0})
    assert clf0.min_new_tokens() == 0 and "min_new_tokens" not in clf0.hf_generate_kwargs()
    with torch.no_grad():
        assert bool((clf0.greedy_tokens(embs)[:, 0] == EOS).all())


def test_checkpoint_min_length_cannot_split_the_decode_paths():
    """a checkpoint whose generation_config.json carries min_length > 0.  The call used to be SILENT on min_length (it
    travels as min_new_tokens), so `generate` applied the checkpoint's value while the greedy loop did not know it.  Now the HF call
    carries an explicit min_length = 0 beside min_new_tokens: both paths suppress EOS for exactly min_new_tokens() generated tokens,
    with the reference's min_length = 1 and with min_length = 0."""
    from toy_llm import EOS
    llm, emb, embs = _eos_first_llm()
    llm.generation_config.min_length = 3
    for gk, first_is_eos in (({}, False), ({"min_length": 0}, True)):
        clf = _classifier(llm=llm, generate_kwargs=gk)
        kw = clf.hf_generate_kwargs()
        assert kw["min_length"] == 0 and clf._greedy_defaults() is True
        with torch.no_grad():
            hf = llm.generate(inputs_embeds=embs, attention_mask=torch.ones(embs.shape[:2], dtype=torch.int), max_new_tokens=6, **kw)
            loop = clf.greedy_tokens(embs)
        assert bool((hf[:, 0] == EOS).all()) == first_is_eos and bool((loop[:, 0] == EOS).all()) == first_is_eos
        assert clf._decode_outputs(loop) == clf._decode_outputs(hf)


def test_checkpoint_generation_config_sends_the_call_down_the_hf_path():
    """`generate` also applies the checkpoint's own generation_config (generation_config.json).  A field there that
    creates a logits processor the greedy loop does not reproduce -- here no_repeat_ngram_size -- must disqualify the loop."""
    llm = tiny_llama(hidden=64, seed=7)
    clf = _classifier(llm=llm)
    assert clf._greedy_defaults() is True
    llm.generation_config.no_repeat_ngram_size = 2
    assert clf._greedy_defaults() is False
    llm.generation_config.no_repeat_ngram_size = 0
    llm.generation_config.repetition_penalty = 1.3                    # overridden by the call's own repetition_penalty = 1: neutral
    assert clf._greedy_defaults() is True
    clf_rp = _classifier(llm=llm, generate_kwargs={"repetition_penalty": 1.3})
    assert clf_rp._greedy_defaults() is False
    llm.generation_config.suppress_tokens = [5]
    assert clf._greedy_defaults() is False
