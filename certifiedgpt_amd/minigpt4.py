"""MiniGPT4Classifier -- full MiniGPT-4 (`generate`) as the base classifier of `Smooth` (BASELINE configs[2]).

Mirrors `MiniGPTBase.generate` (graphs/models/minigpt4/models/minigpt_base.py:374-448) and `get_context_emb` (:75-89):

    encode_img(images)            -> libcgpt.so  (cgpt_encode_img / cgpt_encode_img_noisy: HIP noise + ViT-G + Q-Former +
                                                  llama_proj straight into the buffer the LLM reads; minigpt4.py:121-149)
    prompt splice + left padding  -> this file, PyTorch-ROCm tensors (minigpt_base.py:401-416)
    llama_model.generate(...)     -> Hugging Face `generate` on PyTorch-ROCm, greedy, <= max_new_tokens (:418-431);
                                     the LLM is a frozen decoder loaded BY LOCAL PATH by the caller (base_model.py:181-247) --
                                     nothing is downloaded here
    decode + clean-up             -> this file (:441-448)
    answer -> class               -> agents/label_adapter.AnswerLabelMap (the "decoder-to-label" mapping, README.md:28-29)
    argmax / vote histogram       -> libcgpt.so (cgpt_vote: wavefront arg-max + 64-bit atomics)

The LLM decode is memory-bound PyTorch work and stays outside the MFMA-roofline claim; what this class adds
for the Monte-Carlo loop is that the N noisy copies of one image share ONE text prompt: the prompt segments are tokenised and
embedded once per call and broadcast, and because every row then has the same length there is no padding and the attention
mask is all ones (identical to what the reference's left-padding produces for equal lengths).

`Smooth` sees the engine interface (`sample_counts`), so noise generation stays fused into the patch-embed operand.
"""
import torch

from .classifier import vote

IMAGE_PLACEHOLDER = "<ImageHere>"


def prepare_texts(questions):
    """`MiniGPT4EvalAgent.prepare_texts` with CONV_VISION_minigptv2 (agents/minigpt4_eval_agent.py:265-271; conversation.py:130-137:
    system "", roles ("<s>[INST] ", " [/INST]"), sep ""): one user turn, assistant turn left open."""
    return ["<s>[INST] " + q + " [/INST]" for q in questions]


def clean_answer(text: str) -> str:
    """Post-processing of a decoded sequence, minigpt_base.py:444-447."""
    text = text.split("</s>")[0]
    text = text.replace("<s>", "")
    return text.split(r"[/INST]")[-1].strip()


class WordHashTokenizer:
    """Stand-in for LlamaTokenizer where no tokenizer model file exists (the container has no Vicuna directory and nothing is
    ever downloaded): words map to ids by a stable hash into [3, vocab_size), id k decodes to the word "w<k>" (token-id strings
    as answers).  It offers exactly the tokenizer surface `MiniGPTBase` uses: `__call__(text, return_tensors="pt",
    add_special_tokens=...)` -> object with `.input_ids` / `.to(device)`, and `decode(ids, skip_special_tokens=True)`.
    Used by the tests and by `bench.py --workload minigpt4` (synthetic decoder of the Vicuna-7B architecture)."""
    pad_token_id, bos_token_id, eos_token_id = 0, 1, 2

    class _Encoding:
        def __init__(self, ids):
            self.input_ids = ids
            self.attention_mask = torch.ones_like(ids)

        def to(self, device):
            self.input_ids = self.input_ids.to(device)
            self.attention_mask = self.attention_mask.to(device)
            return self

    def __init__(self, vocab_size=96):
        self.vocab_size = int(vocab_size)

    def _word_id(self, w):
        h = 0
        for ch in w:
            h = (h * 131 + ord(ch)) % 1000003
        return 3 + h % (self.vocab_size - 3)

    def __call__(self, text, return_tensors="pt", add_special_tokens=True, **_):
        ids = [self._word_id(w) for w in text.split()]
        if add_special_tokens:
            ids = [self.bos_token_id] + ids
        return self._Encoding(torch.tensor([ids], dtype=torch.long))

    def decode(self, ids, skip_special_tokens=True):
        out = []
        for t in ids.tolist():
            if t in (self.pad_token_id, self.bos_token_id):
                if not skip_special_tokens:
                    out.append("<s>" if t == self.bos_token_id else "<pad>")
            elif t == self.eos_token_id:
                out.append("</s>")                       # the reference splits on the literal stop sign, minigpt_base.py:445
            else:
                out.append(f"w{t}")
        return " ".join(out)


def greedy_decode_parity(hf_tokens, hf_scores, tokens, eps):
    """Compare a greedy decode `tokens` [B, n] with HF `generate`'s `hf_tokens` [B, <= n] row by row, given HF's per-step processed
    scores `hf_scores` [B, steps, V] (fp32).  Two greedy decoders of the SAME fp16 model can only part ways at a step where HF's own
    top-2 margin is within the logit noise of the two paths (another GEMM / attention kernel rounds the fp16 logits differently), after
    which the contexts differ and nothing more can be said about that row.  Returns a dict: `identical` rows, `decisive` rows (every
    step's margin > eps: these MUST be identical), `violations` = rows whose first divergence sits at a step with margin > eps."""
    B = hf_tokens.shape[0]
    S = min(hf_tokens.shape[1], tokens.shape[1], hf_scores.shape[1])
    top2 = hf_scores[:, :S].float().topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]                                  # [B, S]; inf where only one candidate is finite
    margin = torch.nan_to_num(margin, nan=float("inf"))
    neq = tokens[:, :S] != hf_tokens[:, :S]
    diverged = neq.any(dim=1)
    first = torch.where(diverged, neq.float().argmax(dim=1), torch.zeros(B, dtype=torch.long, device=neq.device))
    m_at = margin.gather(1, first[:, None])[:, 0]
    violations = (diverged & (m_at > eps)).nonzero().flatten().tolist()
    decisive = (margin > eps).all(dim=1)
    return {"identical": int((~diverged).sum()), "diverged": int(diverged.sum()), "decisive": int(decisive.sum()),
            "decisive_identical": int((decisive & ~diverged).sum()), "violations": violations,
            "max_margin_at_divergence": float(m_at[diverged].max()) if bool(diverged.any()) else 0.0}


def fp16_ulp(x):
    """Spacing of fp16 at magnitude x (normal range)."""
    import math
    return 2.0 ** (math.floor(math.log2(max(float(x), 6.2e-5))) - 10)


class _LinearRoute:
    """Routes the bias-free fp16 `nn.Linear` layers of a frozen decoder through libcgpt's MFMA GEMM (`cgpt_linear_f16`, the kernel
    of the ViT's qkv / proj / MLP) for calls with at least `min_rows` rows, i.e. the PREFILL of a Monte-Carlo batch (200 rows x 44
    tokens = 8 800 rows); decode steps (one row per copy) stay with the vendor library.  One patch per module; the patch is inert
    outside `enabled()`, and the switch is a context variable (per thread / per task), so another thread or another classifier that
    calls the same shared decoder meanwhile is not routed.  Eligibility (fp16, contiguous, on the device, no bias, K in 64s, N in 256s)
    is re-checked at every call: after a `.to()`, a dtype change or a weight swap (a LoRA merge) the original forward runs."""
    import contextvars as _cv
    _active = _cv.ContextVar("cgpt_linear_route", default=False)

    @staticmethod
    def _eligible(m):
        w = m.weight
        return (m.bias is None and w.dtype == torch.float16 and w.is_cuda and w.is_contiguous()
                and w.shape == (m.out_features, m.in_features) and m.in_features % 64 == 0 and m.out_features % 256 == 0)

    @classmethod
    def install(cls, model, min_rows=1024):
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        n = 0
        for m in model.modules():
            if not isinstance(m, torch.nn.Linear):
                continue
            if getattr(m, "_cgpt_routed", False):                       # patched by an earlier classifier on the same module
                n += 1
                continue
            if not cls._eligible(m):                                    # W must be readable for whole 256-row tiles, K in 64s
                continue
            orig = m.forward

            def fwd(x, m=m, orig=orig):
                rows = x.numel() // x.shape[-1] if x.shape[-1] else 0
                if (not cls._active.get() or rows < min_rows or x.dtype != torch.float16 or not x.is_cuda
                        or x.device != m.weight.device or not cls._eligible(m)):
                    return orig(x)
                K, N = m.in_features, m.out_features
                a = x.reshape(rows, K)
                if rows % 256 or not a.is_contiguous():                 # the kernel reads whole 256-row tiles: pad rows exist but are
                    pad = x.new_empty(((rows + 255) // 256 * 256, K))   # never stored from (their garbage stays in their own outputs)
                    pad[:rows].copy_(a)
                    a = pad
                out = x.new_empty((rows, N))
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                _lib.check(L.cgpt_linear_f16(C.c_void_p(a.data_ptr()), K, C.c_void_p(m.weight.data_ptr()), K, None,
                                             C.c_void_p(out.data_ptr()), N, None, N, rows, N, K, 0, st))
                return out.view(*x.shape[:-1], N)
            m.forward = fwd
            m._cgpt_routed = True
            n += 1
        return n

    @classmethod
    def enabled(cls):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            token = cls._active.set(True)
            try:
                yield
            finally:
                cls._active.reset(token)
        return ctx()


class MiniGPT4Classifier:
    """:param encoder: a `HipClassifier(mode="encode_img")` (or any object with `encode_img(images) -> (emb, atts)` and
               `encode_img_noisy(x, first_sample, num, sigma, seed) -> emb`, emb = [B, queries, llm_hidden])
       :param llama_model: a causal LM with `.generate(inputs_embeds=..., attention_mask=...)` and an input embedding table
               (`get_input_embeddings()`), e.g. `LlamaForCausalLM.from_pretrained(<local path>, torch_dtype=torch.float16)`
       :param llama_tokenizer: its tokenizer (`__call__(text, return_tensors="pt", add_special_tokens=...)`, `decode`)
       :param prompt: the text every noisy copy is asked, containing one `<ImageHere>`
       :param label_map: `AnswerLabelMap` -- frozen when torch.distributed is initialised (see label_adapter.py)
    """

    def __init__(self, encoder, llama_model, llama_tokenizer, prompt, label_map, max_new_tokens=20, max_batch=None,
                 generate_kwargs=None, decode="hf", prefill_linear="torch"):
        self.encoder = encoder
        self.llama_model = llama_model
        self.llama_tokenizer = llama_tokenizer
        self.prompt = prompt
        self.label_map = label_map
        self.num_classes = label_map.num_classes
        self.max_new_tokens = int(max_new_tokens)
        self.max_batch = int(max_batch or getattr(encoder, "max_batch", 16))
        # the reference's fixed generation arguments, minigpt_base.py:379-388,418-431
        self.generate_kwargs = dict(num_beams=1, min_length=1, top_p=0.9, repetition_penalty=1, length_penalty=1,
                                    temperature=1, do_sample=False)
        self.generate_kwargs.update(generate_kwargs or {})
        self.last_answers = []
        # decode = "hf": `llama_model.generate(...)` exactly as the reference calls it (minigpt_base.py:418-431).
        # decode = "graph": the same greedy decode -- the model's own forward for the prefill and for every step, a DynamicCache, EOS
        #   suppressed on the first `min_length` generated tokens (the reference's semantics, see hf_generate_kwargs), pad after EOS -- written as a fixed-length loop without host round trips and
        #   replayed from ONE hipGraph per (batch, prompt length); shared prompts on a HIP device only, everything else takes "hf".
        #   Parity with HF generate (tests/test_gpu_fullsize.py, Vicuna-7B widths, 200 rows x 20 tokens): run eagerly the loop gives HF's
        #   tokens and logits bit for bit; under graph capture other vendor kernels get picked, so logits move by fp16 rounding and a row
        #   may leave HF's trajectory at a step where HF's own top-2 margin is within that noise (near-tied logits of a random-init
        #   decoder) -- every decisive row gives the identical answer.
        # prefill_linear = "cgpt" (with decode = "graph"): the decoder's bias-free fp16 linears run the prefill's rows through this
        #   library's GEMM (see _LinearRoute); same arithmetic contract (fp16 operands, fp32 accumulation, fp16 result), other rounding
        #   order than the vendor library: first-token logits within 4 fp16 ulp, same decisive-row rule.
        assert decode in ("hf", "graph") and prefill_linear in ("torch", "cgpt")
        self.decode = decode
        self.prefill_linear = prefill_linear
        self.routed_linears = _LinearRoute.install(llama_model) if (decode == "graph" and prefill_linear == "cgpt") else 0
        import collections
        self._graphs = collections.OrderedDict()            # LRU of captured decode graphs, see _generate_graph
        self.max_graphs = 4
        self._eos_mask = None
        self.decode_stats = {"graph_replays": 0, "graph_captures": 0, "graph_evictions": 0, "hf_calls": 0, "hf_fallback_kwargs": 0}

    # ---- nn.Module-shaped surface used by Smooth (smoothing.py:42,71)
    def eval(self):
        if hasattr(self.llama_model, "eval"):
            self.llama_model.eval()
        return self

    @property
    def device(self):
        return next(self.llama_model.parameters()).device

    def embed_tokens(self, token_ids):
        """minigpt_base.py:366-371 (the LoRA-wrapped variant resolves to the same table)."""
        return self.llama_model.get_input_embeddings()(token_ids)

    # ---- minigpt_base.py:75-89
    def _segment_embeddings(self, prompt, device):
        segs = prompt.split(IMAGE_PLACEHOLDER)
        toks = [self.llama_tokenizer(seg, return_tensors="pt", add_special_tokens=(i == 0)).input_ids.to(device)
                for i, seg in enumerate(segs)]                         # only add bos to the first seg
        return [self.embed_tokens(t) for t in toks]

    def get_context_emb(self, prompt, img_list):
        seg_embs = self._segment_embeddings(prompt, img_list[0].device)
        assert len(seg_embs) == len(img_list) + 1, "Unmatched numbers of image placeholders and images."
        mixed = [emb for pair in zip(seg_embs[:-1], img_list) for emb in pair] + [seg_embs[-1]]
        return torch.cat(mixed, dim=1)

    # ---- minigpt_base.py:374-448 with encode_img's output given
    @torch.no_grad()
    def generate_from_embeds(self, img_embeds, texts):
        """img_embeds [B, queries, hidden] (any float dtype) + one prompt per row (or ONE str shared by all rows) -> list[str]."""
        dtype = self.embed_tokens(torch.zeros(1, 1, dtype=torch.long, device=img_embeds.device)).dtype
        img_embeds = img_embeds.to(dtype)
        B = img_embeds.shape[0]
        if isinstance(texts, str) or len(set(texts)) == 1:
            # shared prompt: segments embedded once and broadcast; equal lengths -> no padding, all-ones mask
            prompt = texts if isinstance(texts, str) else texts[0]
            segs = self._segment_embeddings(prompt, img_embeds.device)
            assert len(segs) == 2, "Unmatched numbers of image placeholders and images."
            embs = torch.cat([segs[0].expand(B, -1, -1), img_embeds, segs[1].expand(B, -1, -1)], dim=1)
            attn_mask = torch.ones(embs.shape[:2], dtype=torch.int, device=embs.device)
        else:
            assert len(texts) == B
            batch_embs = [self.get_context_emb(t, [img_embeds[i][None]]) for i, t in enumerate(texts)]
            max_len = max(e.shape[1] for e in batch_embs)
            embs = torch.zeros([B, max_len, batch_embs[0].shape[2]], dtype=dtype, device=img_embeds.device)
            attn_mask = torch.zeros([B, max_len], dtype=torch.int, device=img_embeds.device)
            for i, emb in enumerate(batch_embs):                       # left padding, minigpt_base.py:413-416
                embs[i, -emb.shape[1]:] = emb[0]
                attn_mask[i, -emb.shape[1]:] = 1
        shared = isinstance(texts, str) or len(set(texts)) == 1
        if self.decode == "graph" and shared and embs.is_cuda and self._greedy_defaults():
            outputs = self._generate_graph(embs)
        else:
            self.decode_stats["hf_calls"] += 1
            if self.decode == "graph" and shared and embs.is_cuda:
                self.decode_stats["hf_fallback_kwargs"] += 1            # generation arguments outside the greedy whitelist
            outputs = self.llama_model.generate(inputs_embeds=embs, attention_mask=attn_mask, max_new_tokens=self.max_new_tokens,
                                                **self.hf_generate_kwargs())
        return self._decode_outputs(outputs)

    def _decode_outputs(self, outputs):
        """minigpt_base.py:440-447."""
        answers = []
        for output_token in outputs:
            if output_token[0] == 0:
                output_token = output_token[1:]
            answers.append(clean_answer(self.llama_tokenizer.decode(output_token, skip_special_tokens=True)))
        return answers

    # ---- greedy decode without host round trips (decode = "graph")
    # generation arguments the fixed-length greedy loop reproduces, with the values at which it does (None = any value: without
    # sampling and with one beam HF ignores it).  Anything else -- eos_token_id, stopping_criteria, no_repeat_ngram_size,
    # bad_words_ids, min_new_tokens, max_length, logits_processor, ... -- takes the HF path, so the same classifier never decodes
    # differently by `decode=`.
    _GREEDY_NEUTRAL = {"num_beams": 1, "do_sample": False, "repetition_penalty": 1, "min_length": None, "top_p": None,
                       "temperature": None, "length_penalty": None, "top_k": None}
    # fields of the checkpoint's own `generation_config` (generation_config.json; `generate` merges it under the call's arguments) that
    # create a logits processor, a stopping rule or another decoding mode, with the values at which they do nothing
    _CONFIG_NEUTRAL = {"repetition_penalty": (None, 1, 1.0), "no_repeat_ngram_size": (None, 0), "encoder_no_repeat_ngram_size": (None, 0),
                       "encoder_repetition_penalty": (None, 1, 1.0), "bad_words_ids": (None,), "suppress_tokens": (None,),
                       "begin_suppress_tokens": (None,), "forced_bos_token_id": (None,), "forced_eos_token_id": (None,),
                       "sequence_bias": (None,), "min_new_tokens": (None, 0), "num_beams": (None, 1), "num_beam_groups": (None, 1),
                       "do_sample": (None, False), "penalty_alpha": (None, 0, 0.0), "exponential_decay_length_penalty": (None,),
                       "renormalize_logits": (None, False), "remove_invalid_values": (None, False), "guidance_scale": (None, 1, 1.0),
                       "watermarking_config": (None,), "prompt_lookup_num_tokens": (None,), "stop_strings": (None,),
                       "diversity_penalty": (None, 0, 0.0), "max_time": (None,), "dola_layers": (None,)}

    def _greedy_defaults(self):
        """May this call take the fixed-length greedy loop?  Only when every argument of the call AND every field of the model's own
        generation_config (which `generate` applies wherever the call is silent) is one the loop reproduces; otherwise the HF path
        runs, so the same classifier never decodes differently by `decode=`."""
        for key, val in self.generate_kwargs.items():
            if key not in self._GREEDY_NEUTRAL:
                return False
            want = self._GREEDY_NEUTRAL[key]
            if want is not None and val != want:
                return False
        gen = getattr(self.llama_model, "generation_config", None)
        if gen is not None:
            for key, neutral in self._CONFIG_NEUTRAL.items():
                if key in self.generate_kwargs:                        # the call's argument wins over the checkpoint's field
                    continue
                val = getattr(gen, key, None)
                if isinstance(val, (list, tuple, dict)) and len(val) == 0:
                    val = None
                if val not in neutral:
                    return False
        return True

    def min_new_tokens(self):
        """Generated tokens before which EOS is suppressed.  The reference passes `min_length = 1` with `inputs_embeds`
        (minigpt_base.py:385) to transformers 4.30.0 (its pin, docker/tpu-docker:32), which counts GENERATED tokens in that call: the
        first token is never EOS and an answer is never empty.  Later versions subtract the embedded prompt's length from min_length
        (5.15: `_prepare_generated_length`, max(min_length - inputs_embeds.shape[1], 0) = 0).  This class ships the REFERENCE's
        semantics on both decode paths, whatever transformers is installed."""
        return int(self.generate_kwargs.get("min_length", 1) or 0)

    def hf_generate_kwargs(self):
        """`generate_kwargs` as handed to the installed `generate`: the reference's `min_length` travels as `min_new_tokens`, which
        means "generated tokens" in every transformers version (and takes precedence over min_length where both exist), and
        `min_length` itself is handed over as an explicit 0."""
        kw = dict(self.generate_kwargs)
        ml = kw.pop("min_length", None)
        if ml and "min_new_tokens" not in kw:
            kw["min_new_tokens"] = int(ml)
        # never SILENT on min_length: where the call says nothing `generate` applies the checkpoint's own generation_config.min_length,
        # which the greedy loop (min_new_tokens() above) would not see -- the two decode paths of one classifier would then differ
        kw["min_length"] = 0
        return kw

    def greedy_tokens(self, embs, return_logits=False):
        """What `generate(inputs_embeds=embs, attention_mask=ones, max_new_tokens=n, do_sample=False, min_length=1, ...)` returns for
        prompts without padding, as a straight loop of the model's own forward calls: [B, n] token ids, pad_token_id after a row's
        EOS.  (HF additionally stops, and truncates, at the step where every row has finished; the extra columns here are pad ids,
        which decode to nothing.)  No data-dependent control flow: capturable as one graph."""
        from transformers import DynamicCache
        from transformers.cache_utils import DynamicLayer

        class PreallocLayer(DynamicLayer):
            """DynamicLayer whose keys / values are exact-length views of buffers allocated once for prompt + new tokens: same
            values as the concatenating layer, without re-copying the whole cache at every step (4 ms of an 18-ms step at 200 rows
            x 32 layers; profiles/r03/minigpt4_decode.txt)."""

            def __init__(self, max_len):
                super().__init__()
                self.max_len, self.used, self.buf_k, self.buf_v = max_len, 0, None, None

            def update(self, key_states, value_states, *args, **kwargs):
                if self.buf_k is None:
                    b, h, _, d = key_states.shape
                    self.buf_k = key_states.new_empty((b, h, self.max_len, d))
                    self.buf_v = value_states.new_empty((b, h, self.max_len, d))
                    self.dtype, self.device, self.is_initialized = key_states.dtype, key_states.device, True
                n = key_states.shape[-2]
                self.buf_k[:, :, self.used:self.used + n] = key_states
                self.buf_v[:, :, self.used:self.used + n] = value_states
                self.used += n
                self.keys, self.values = self.buf_k[:, :, :self.used], self.buf_v[:, :, :self.used]
                return self.keys, self.values

        llm = self.llama_model
        gen = getattr(llm, "generation_config", None)
        eos = getattr(gen, "eos_token_id", None)
        eos = getattr(llm.config, "eos_token_id", None) if eos is None else eos
        eos_ids = list(eos) if isinstance(eos, (list, tuple)) else ([eos] if eos is not None else [])
        pad = getattr(gen, "pad_token_id", None)
        pad = getattr(llm.config, "pad_token_id", None) if pad is None else pad
        pad = (eos_ids[0] if eos_ids else 0) if pad is None else pad
        B, L, _ = embs.shape
        cache = DynamicCache(config=llm.config)
        cache.layers = [PreallocLayer(L + self.max_new_tokens) for _ in cache.layers] or cache.layers
        pos = torch.arange(L, device=embs.device).unsqueeze(0)
        out = llm(inputs_embeds=embs, position_ids=pos, past_key_values=cache, use_cache=True, logits_to_keep=1)
        logits = out.logits[:, -1, :].float()
        min_new = self.min_new_tokens() if eos_ids else 0              # MinNewTokensLengthLogitsProcessor: no EOS on the first tokens
        if min_new > 0:
            key = (str(embs.device), logits.shape[-1])
            if self._eos_mask is None or self._eos_mask[0] != key:     # built outside any capture (the warm-up run comes first)
                m = torch.zeros(logits.shape[-1], dtype=torch.bool)
                m[eos_ids] = True
                self._eos_mask = (key, m.to(embs.device))
        unfinished = torch.ones(B, dtype=torch.long, device=embs.device)
        tokens, step_logits = [], []
        for i in range(self.max_new_tokens):
            if i < min_new:
                logits = logits.masked_fill(self._eos_mask[1], float("-inf"))
            if return_logits:
                step_logits.append(logits)
            nxt = logits.argmax(dim=-1)
            nxt = nxt * unfinished + pad * (1 - unfinished)
            tokens.append(nxt)
            for e in eos_ids:
                unfinished = unfinished * (nxt != e).long()
            if i + 1 == self.max_new_tokens:
                break
            pos = torch.full((1, 1), L + i, device=embs.device, dtype=torch.long)
            out = llm(input_ids=nxt[:, None], position_ids=pos, past_key_values=cache, use_cache=True, logits_to_keep=1)
            logits = out.logits[:, -1, :].float()
        if return_logits:                                              # tests: the per-step logits the argmax was taken on
            return torch.stack(tokens, dim=1), torch.stack(step_logits, dim=1)
        return torch.stack(tokens, dim=1)

    def _generate_graph(self, embs):
        """One captured hipGraph per (batch, prompt length, dtype, device, max_new_tokens, routed prefill), kept in an LRU of
        `max_graphs` entries: a graph pins its private pool -- the pre-allocated K / V of every layer plus activations, ~6.7 GB of K / V
        alone at 200 rows x 32 layers x 64 positions -- and ragged last batches or new prompt lengths add entries; the least recently
        used one is dropped (graph and static tensors freed) before a new capture."""
        key = (embs.shape[0], embs.shape[1], embs.dtype, embs.device.index, self.max_new_tokens, bool(self.routed_linears))
        entry = self._graphs.get(key)
        if entry is not None:
            self._graphs.move_to_end(key)
        if entry is None:
            while len(self._graphs) >= max(1, int(self.max_graphs)):
                _, old = self._graphs.popitem(last=False)
                del old
                self.decode_stats["graph_evictions"] += 1
                torch.cuda.empty_cache()                                # return the evicted graph's pool before capturing again
            static_in = embs.clone()
            side = torch.cuda.Stream(device=embs.device)
            side.wait_stream(torch.cuda.current_stream())
            import contextlib
            route = _LinearRoute.enabled() if self.routed_linears else contextlib.nullcontext()
            with route:
                with torch.cuda.stream(side):                          # warm-up outside the capture (lazy initialisations)
                    self.greedy_tokens(static_in)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = self.greedy_tokens(static_in)
            entry = (graph, static_in, static_out)
            self._graphs[key] = entry
            self.decode_stats["graph_captures"] += 1
        graph, static_in, static_out = entry
        static_in.copy_(embs)
        graph.replay()
        self.decode_stats["graph_replays"] += 1
        return static_out.clone()

    @torch.no_grad()
    def generate(self, images, texts, **_ignored):
        """`MiniGPTBase.generate(images, texts, ...)` -> list[str] (generation arguments are fixed at construction)."""
        img_embeds, _ = self.encoder.encode_img(images)
        return self.generate_from_embeds(img_embeds, texts)

    # ---- base classifier: [B,3,H,W] -> [B,num_classes] one-hot logits (smoothing.py:21,97)
    def _logits(self, answers, device):
        self.last_answers = answers
        return self.label_map.one_hot_logits(answers, device)

    def __call__(self, images):
        out = []
        for lo in range(0, images.shape[0], self.max_batch):
            out.append(self._logits(self.generate(images[lo:lo + self.max_batch], self.prompt), images.device))
        return torch.cat(out)

    def sample_counts_pair(self, x, first_a, num_a, first_b, num_b, batch_size, sigma, seed):
        """Selection + estimation draws of one `certify` (smoothing.py:44,48) in the same classifier batches: [2, num_classes] int64,
        identical to two `sample_counts` calls (a row's answer does not depend on its batch: tested)."""
        counts = torch.zeros((2, self.num_classes), dtype=torch.int64, device=x.device)
        if first_b != first_a + num_a:                                 # not one contiguous index range: two passes
            self.sample_counts(x, first_a, num_a, batch_size, sigma, seed, counts[0])
            self.sample_counts(x, first_b, num_b, batch_size, sigma, seed, counts[1])
            return counts
        bs = max(1, min(int(batch_size), self.max_batch))
        total, done, answers = num_a + num_b, 0, []
        while done < total:
            nb = min(bs, total - done)
            emb = self.encoder.encode_img_noisy(x, first_a + done, nb, float(sigma), int(seed))
            batch_answers = self.generate_from_embeds(emb, self.prompt)
            logits = self.label_map.one_hot_logits(batch_answers, x.device)
            na = max(0, min(nb, num_a - done))                         # rows of this batch that belong to the selection range
            if na > 0:
                vote(logits[:na], counts[0])
            if nb - na > 0:
                vote(logits[na:], counts[1])
            answers += batch_answers
            done += nb
        self.last_answers = answers
        return counts

    # ---- the `_sample_noise` engine (smoothing.py:81-99): noise fused into encode_img, vote in HIP
    def sample_counts(self, x, first_sample, num, batch_size, sigma, seed, counts=None):
        if counts is None:
            counts = torch.zeros(self.num_classes, dtype=torch.int64, device=x.device)
        bs = max(1, min(int(batch_size), self.max_batch))
        done = 0
        while done < num:
            nb = min(bs, num - done)
            emb = self.encoder.encode_img_noisy(x, first_sample + done, nb, float(sigma), int(seed))
            vote(self._logits(self.generate_from_embeds(emb, self.prompt), x.device), counts)
            done += nb
        return counts
