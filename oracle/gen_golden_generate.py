#!/usr/bin/env python3
"""Generate tests/golden/generate_golden.{json,npz} by executing the REFERENCE's own `MiniGPTBase.generate`,
`MiniGPTBase.get_context_emb` and `MiniGPTBase.embed_tokens` (graphs/models/minigpt4/models/minigpt_base.py:75-89,366-448) and the
reference's own prompt template (`CONV_VISION_minigptv2`, graphs/models/minigpt4/conversation/conversation.py:130-137, driven as
`MiniGPT4EvalAgent.prepare_texts` drives it, agents/minigpt4_eval_agent.py:265-271) on the CPU in fp32.

Runs ONLY in the build container (needs /root/reference); neither the stand-in modules below nor any reference file travels --
only the fixture (inputs + expected outputs) does.

    python oracle/gen_golden_generate.py

How the reference code is reached (the recipe of gen_golden_model.py):
  * conversation.py is loaded BY FILE PATH; it needs one stand-in (`common.registry.registry`, used only by `Chat`).
  * minigpt_base.py is loaded BY FILE PATH after registering stand-ins for names that are not on `generate`'s arithmetic:
    torch_xla (imported, never called on this path), `common.registry`, and
    `graphs.models.minigpt4.models.base_model.BaseModel` -- whose real module imports omegaconf / peft / torch_xla (absent here) --
    as an `nn.Module` with the two members `generate` touches: `device` (base_model.py:37-39) and `maybe_autocast`
    (base_model.py:132-142: `contextlib.nullcontext()` on a CPU device).
  * `MiniGPTBase.__init__` needs Vicuna + the ViT checkpoint (init_llm / init_vision_encoder), so the UNBOUND methods are called
    on a bare instance carrying what they read: `llama_model`, `llama_tokenizer`, `encode_img`.
  * The LLM is a random-init tiny `LlamaForCausalLM` (its weights go into the .npz so that no RNG stream is part of the contract)
    and the tokenizer is the toy word tokenizer of tests/toy_llm.py: stand-ins for Vicuna-7B + LlamaTokenizer, which are not in
    the container.  `encode_img` returns prescribed embeddings (its own parity: tests/test_gpu_fullsize.py).
  * The decode clean-up (:441-447) is additionally driven with a scripted LM whose `generate` returns prescribed token rows
    (leading pad id 0, the stop sign mid-sequence, tokens after it, an echoed "[/INST]"), which a random decoder never emits.
"""
import contextlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/graphs/models/minigpt4"
OUT = os.path.join(ROOT, "tests", "golden", "generate_golden")


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _BaseModel(nn.Module):
    """Stand-in for base_model.py:31-39,132-142 (see the module docstring)."""

    @property
    def device(self):
        return list(self.parameters())[-1].device

    def maybe_autocast(self, dtype=torch.float16):
        assert self.device.type == "cpu"
        return contextlib.nullcontext()


def load_reference():
    _module("common"); _module("common.registry", registry=types.SimpleNamespace())
    conv = _load("ref_conversation", os.path.join(REF, "conversation", "conversation.py"))
    _module("torch_xla"); _module("torch_xla.amp", autocast=None); _module("torch_xla.core")
    _module("torch_xla.core.xla_model")
    for n in ("graphs", "graphs.models", "graphs.models.minigpt4", "graphs.models.minigpt4.models",
              "graphs.models.minigpt4.conversation"):
        _module(n)
    _module("graphs.models.minigpt4.models.base_model", BaseModel=_BaseModel)
    sys.modules["graphs.models.minigpt4.conversation.conversation"] = conv
    base = _load("ref_minigpt_base", os.path.join(REF, "models", "minigpt_base.py"))
    return base, conv


def reference_prompts(conv, questions):
    """agents/minigpt4_eval_agent.py:265-271 with conv_temp = CONV_VISION_minigptv2 (:80)."""
    convs = [conv.CONV_VISION_minigptv2.copy() for _ in questions]
    for c, q in zip(convs, questions):
        c.append_message(c.roles[0], q)
        c.append_message(c.roles[1], None)
    return [c.get_prompt() for c in convs]


class ScriptedLM(nn.Module):
    """An LM whose `generate` returns prescribed rows: drives minigpt_base.py:441-447 through cases a random decoder never emits."""

    def __init__(self, embed, rows):
        super().__init__()
        self.base_model = types.SimpleNamespace(embed_tokens=embed)
        self.rows = rows
        self.p = nn.Parameter(torch.zeros(1))

    def generate(self, inputs_embeds=None, **kw):
        assert inputs_embeds.shape[0] == len(self.rows)
        return torch.tensor(self.rows, dtype=torch.long)


def main():
    from toy_llm import ToyTokenizer, tiny_llama, EOS
    base, conv = load_reference()
    MiniGPTBase = base.MiniGPTBase
    torch.manual_seed(0)
    llm = tiny_llama(hidden=64, seed=0)
    tok = ToyTokenizer()
    queries, hidden = 4, 64

    def bare(llama_model, embeds):
        obj = MiniGPTBase.__new__(MiniGPTBase)
        nn.Module.__init__(obj)
        obj.llama_model = llama_model
        obj.llama_tokenizer = tok
        obj.encode_img = lambda images: (embeds, torch.ones(embeds.shape[:-1], dtype=torch.long))
        return obj

    g = torch.Generator().manual_seed(20251121)
    cases, arrays = [], {}
    for k, v in llm.state_dict().items():
        arrays["llama." + k] = v.numpy()

    questions = {
        "shared": ["<Img><ImageHere></Img> [vqa] what is shown here"] * 5,
        "ragged": ["<ImageHere> short", "<Img><ImageHere></Img> a much longer question about the picture",
                   "<ImageHere> mid size one", "<Img><ImageHere></Img> [vqa] is there a dog"],
        "single": ["<Img><ImageHere></Img> [vqa] what colour is the bus"],
    }
    for name, qs in questions.items():
        texts = reference_prompts(conv, qs)
        emb = torch.randn(len(qs), queries, hidden, generator=g) * 0.5
        obj = bare(llm, emb)
        images = torch.zeros(len(qs), 3, 8, 8)                         # ignored by the prescribed encode_img
        for mnt in (6, 20):
            answers = MiniGPTBase.generate(obj, images, texts, max_new_tokens=mnt)
            cases.append({"name": f"{name}_mnt{mnt}", "questions": qs, "texts": texts, "embeds": f"emb.{name}",
                          "max_new_tokens": mnt, "answers": answers})
        ctx = MiniGPTBase.get_context_emb(obj, texts[0], [emb[0][None]])
        arrays[f"emb.{name}"] = emb.numpy()
        arrays[f"ctx.{name}"] = ctx.detach().numpy()
        print(name, [c["answers"] for c in cases if c["name"].startswith(name)])

    # decode clean-up through prescribed rows: leading pad 0 (:442-443), stop sign and what follows (:445), an echoed prompt (:447)
    inst = tok("[/INST]", add_special_tokens=False).input_ids[0, 0].item()
    rows = [[0, 17, 23, EOS, 40, 41], [1, 17, 23, 24, 25, EOS], [17, inst, 30, 31, EOS, 0], [EOS, 5, 6, 7, 8, 9],
            [0, 0, 12, 13, 14, 15]]
    texts = reference_prompts(conv, ["<ImageHere> q"] * len(rows))
    emb = torch.randn(len(rows), queries, hidden, generator=g)
    obj = bare(ScriptedLM(llm.get_input_embeddings(), rows), emb)
    answers = MiniGPTBase.generate(obj, torch.zeros(len(rows), 3, 8, 8), texts)
    cases.append({"name": "scripted_cleanup", "texts": texts, "rows": rows, "embeds": "emb.scripted", "answers": answers,
                  "decoded": [tok.decode(torch.tensor(r), skip_special_tokens=True) for r in rows]})
    arrays["emb.scripted"] = emb.numpy()
    print("scripted", answers)

    # ... and through prescribed DECODED strings (the toy tokenizer never decodes to "<s>" / "[/INST]" literals): row i decodes to
    # strings[i]; what :444-447 make of them is the golden for the product's clean_answer
    strings = ["<s> [INST] q [/INST] a cat </s> junk", "plain", "  two  words  ", "yes</s></s>", "<s><s>no", "a [/INST] b [/INST] c",
               "</s>", "[/INST]", "x<s>y</s>z[/INST]w", "Yes.\n"]

    class ScriptedTokenizer(ToyTokenizer):
        def decode(self, ids, skip_special_tokens=True):
            return strings[int(ids[0]) - 10]

    rows2 = [[10 + i, 3] for i in range(len(strings))]
    emb2 = torch.randn(len(rows2), queries, hidden, generator=g)
    obj = bare(ScriptedLM(llm.get_input_embeddings(), rows2), emb2)
    obj.llama_tokenizer = ScriptedTokenizer()
    cleaned = MiniGPTBase.generate(obj, torch.zeros(len(rows2), 3, 8, 8), reference_prompts(conv, ["<ImageHere> q"] * len(rows2)))
    cases.append({"name": "scripted_strings", "decoded": strings, "answers": cleaned})
    print("strings", cleaned)

    meta = {"generator": "oracle/gen_golden_generate.py (the reference's MiniGPTBase.generate / get_context_emb / "
                         "CONV_VISION_minigptv2, CPU fp32)",
            "transformers": __import__("transformers").__version__, "torch": torch.__version__,
            "llama": {"hidden": hidden, "vocab": tok.vocab_size, "queries": queries}, "cases": cases}
    with open(OUT + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(OUT + ".npz", **arrays)
    print("wrote", OUT + ".json", OUT + ".npz", os.path.getsize(OUT + ".npz"), "bytes")


if __name__ == "__main__":
    main()
