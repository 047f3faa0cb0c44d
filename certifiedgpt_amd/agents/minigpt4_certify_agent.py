"""Certify agent -- the reference ships `agents/minigpt4_certify_agent.py` as an EMPTY file, so the
shape here is inferred from the plugin contract (launch.py:97-107) and from the sibling eval agent
(agents/minigpt4_eval_agent.py:52-124): build the classifier, loop over (image, label) samples, call Smooth.certify,
log one line per sample (idx, label, predict, radius, correct, time -- the format of Cohen et al.'s certify.py that
`Smooth` comes from), then report certified accuracy at a few radii.

Config (a plain dict or any mapping; the reference's YAMLs define no smoothing keys, so these are build-side):
    run:   {agent: image_text_certify, output_dir: ..., seed: 0,
            smoothing: {sigma: 0.5, n0: 100, n: 100, alpha: 0.001, batch_size: 100, num_classes: 1000, radii: [0.25, 0.5, 1.0],
                        images_per_pass: 1,       # > 1: Smooth.certify_many (multi-GPU throughput mode)
                        shard: samples}}          # images: Smooth.certify_images / predict_images -- every rank takes whole images
            #                                       (all their draws) of each group of images_per_pass x world; no vote all-reduce
    model: {generate: {llama_model: <LOCAL dir>, prompt | question: ..., answers: [...], max_new_tokens: 20},   # optional: full
            #          MiniGPT-4 `generate` as the base classifier (certifiedgpt_amd/minigpt4.py); classes = the answer vocabulary
            mode: vit_head | encode_img, weights: <path to a torch state_dict saved with torch.save> | null, dims: {...},
            partial_weights: false}   # true: load what the checkpoint has on top of the synthetic initialisation
    data:  {num_images: 10, seed: 1234}      # synthetic CLIP-normalised images unless `dataset` is passed to the agent
"""
import json
import os
import time

import torch

from ..smoothing import Smooth
from .base import BaseAgent
from .registry import registry

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)    # processors/base_processor.py:18-20
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def synthetic_dataset(num_images, img_size, seed, device, num_classes):
    g = torch.Generator(device="cpu").manual_seed(seed)
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1)
    for i in range(num_images):
        u = torch.rand(3, img_size, img_size, generator=g)
        yield ((u - mean) / std).to(device), int(torch.randint(0, num_classes, (1,), generator=g))


class CertifyLoop:
    """Engine-agnostic loop: usable as a mixin with the reference's own BaseAgent."""

    def certify_dataset(self, sm_cfg, dataset, smooth, log_path=None, mode="certify"):
        self.records = []
        f = open(log_path, "w") if log_path else None
        header = "idx\tlabel\tpredict\tradius\tcorrect\ttime" if mode == "certify" else "idx\tlabel\tpredict\tcorrect\ttime"
        if f:
            print(header, file=f, flush=True)
        def emit(idx, label, pred, radius, dt):
            if isinstance(label, (list, tuple)):
                # a VQA sample: `label` is the list of human answers; the prediction is the answer text of the certified class
                # ("" when abstaining) and `correct` its VQA accuracy in [0, 1] (vqa_eval.py:211-247; label_adapter.vqa_accuracy)
                from .label_adapter import vqa_accuracy
                answers = getattr(getattr(smooth.base_classifier, "label_map", None), "answers", [])
                text = answers[int(pred)] if 0 <= int(pred) < len(answers) else ""
                rec = dict(idx=idx, label=-1, predict=int(pred), radius=float(radius), correct=vqa_accuracy(text, label), time=dt,
                           answer=text)
                self.records.append(rec)
                if f:
                    print(f"{idx}\t{'|'.join(sorted(set(label)))}\t{text or 'ABSTAIN'}\t{radius:.6f}\t{rec['correct']:.3f}\t{dt:.3f}", file=f, flush=True)
                return
            rec = dict(idx=idx, label=int(label), predict=int(pred), radius=float(radius), correct=int(pred == label), time=dt)
            self.records.append(rec)
            if f:
                if mode == "certify":
                    print(f"{idx}\t{label}\t{pred}\t{radius:.6f}\t{rec['correct']}\t{dt:.3f}", file=f, flush=True)
                else:
                    print(f"{idx}\t{label}\t{pred}\t{rec['correct']}\t{dt:.3f}", file=f, flush=True)

        shard = sm_cfg.get("shard", "samples")
        if shard not in ("samples", "images"):
            raise ValueError(f"run.smoothing.shard must be 'samples' or 'images', not {shard!r}")
        # Grouping of images.  shard = samples (certify only): Smooth.certify_many runs the per-rank sample slices of `images_per_pass`
        # images in one classifier batch and one all-reduce -- the multi-GPU throughput mode.  shard = images (certify AND predict):
        # groups of images_per_pass x world images, whole images per rank.  Results equal the one-by-one loop either way.
        by_image = shard == "images"
        group = int(sm_cfg.get("images_per_pass", 1)) if (mode == "certify" or by_image) else 1
        if by_image:                                       # SURVEY.md 8(e), the zero-communication mode: whole images per rank
            from ..smoothing import _world
            group = max(group, 1) * _world(smooth.process_group)[1]
        pending = []

        def flush():
            if not pending:
                return
            t0 = time.perf_counter()
            # image-sharded: hand over the list -- Smooth stacks only this rank's own slice of it; sample-sharded: every rank needs all
            xs = [p[1] for p in pending] if by_image else torch.stack([p[1] for p in pending])
            if mode != "certify":                          # predict agent, image-sharded
                outs = [(lab, 0.0) for lab in smooth.predict_images(xs, sm_cfg["n"], sm_cfg["alpha"], sm_cfg["batch_size"])]
            else:
                many = smooth.certify_images if by_image else smooth.certify_many
                outs = many(xs, sm_cfg["n0"], sm_cfg["n"], sm_cfg["alpha"], sm_cfg["batch_size"])
            dt = (time.perf_counter() - t0) / len(pending)
            for (idx, _, label), (pred, radius) in zip(pending, outs):
                emit(idx, label, pred, radius, dt)
            pending.clear()

        for idx, (x, label) in enumerate(dataset):
            if group > 1 or by_image:
                pending.append((idx, x, label))
                if len(pending) == group:
                    flush()
                continue
            t0 = time.perf_counter()
            if mode == "certify":
                pred, radius = smooth.certify(x, sm_cfg["n0"], sm_cfg["n"], sm_cfg["alpha"], sm_cfg["batch_size"])
            else:
                pred, radius = smooth.predict(x, sm_cfg["n"], sm_cfg["alpha"], sm_cfg["batch_size"]), 0.0
            emit(idx, label, pred, radius, time.perf_counter() - t0)
        flush()
        if f:
            f.close()
        return self.records

    def summary(self, radii=(0.0, 0.25, 0.5, 1.0)):
        n = max(len(self.records), 1)
        out = {"images": len(self.records),
               "abstain_rate": sum(r["predict"] == Smooth.ABSTAIN for r in self.records) / n,
               "accuracy": sum(r["correct"] for r in self.records) / n,
               "images_per_s": len(self.records) / max(sum(r["time"] for r in self.records), 1e-9)}
        for r0 in radii:   # certified accuracy at radius r: correct and certified radius >= r (README.md:52-59)
            out[f"certified_acc@{r0}"] = sum(r["correct"] * (r["radius"] >= r0) for r in self.records) / n
        return out


def prepare_state_dict(state, expected_names, allow_partial=False):
    """Checkpoint -> {library weight name: tensor}, accepting the layouts the reference saves / loads:
      * a wrapper dict with the weights under "model" (BLIP-2 / Q-Former / MiniGPT-4 checkpoints, base_model.py:59,261) or
        "model_state_dict" (this build's agents), or a bare state_dict;
      * raw EVA-ViT checkpoints, whose keys have no "visual_encoder." prefix (eva_vit.py:445-456 loads them into the
        encoder module itself).
    Keys the library does not know (the text branch of the Q-Former, the LLM, LoRA weights) are dropped.  Every weight of
    the library is zero until loaded, so a name mismatch would silently leave a layer at zero -- and a ViT whose LayerNorm
    gains are zero votes the head bias on every noisy sample, i.e. certifies the maximum radius for a model that never
    loaded.  Unless allow_partial is set (then the caller must have initialised the weights first), any library weight the
    checkpoint does not cover is an error."""
    for key in ("model", "model_state_dict", "state_dict"):
        if isinstance(state, dict) and key in state and isinstance(state[key], dict):
            state = state[key]
            break
    expected = set(expected_names)
    out = {}
    for k, v in state.items():
        if k in expected:
            out[k] = v
        elif "visual_encoder." + k in expected:
            out["visual_encoder." + k] = v
    missing = sorted(expected - set(out))
    if missing and not allow_partial:
        raise KeyError(f"checkpoint covers {len(out)} of {len(expected)} weights; missing e.g. {missing[:4]} "
                       f"(set model.partial_weights: true to load on top of the synthetic initialisation)")
    return out, missing


def build_generating_classifier(encoder, gen_cfg, num_classes, tokenizer=None, llama_model=None):
    """Full MiniGPT-4 as the base classifier (BASELINE configs[2]): `encoder` (HipClassifier, mode encode_img) + a frozen
    causal LM on PyTorch-ROCm, loaded BY LOCAL PATH only (base_model.py:181-247 loads `llama_model` the same way; nothing is
    ever fetched), + the answer vocabulary that defines the classes.
        gen_cfg: {llama_model: <local dir>, prompt: "... <ImageHere> ...", answers: [...], max_new_tokens: 20,
                  decode: "hf" | "graph", prefill_linear: "torch" | "cgpt"}   (decode / prefill_linear: see MiniGPT4Classifier)"""
    from ..minigpt4 import MiniGPT4Classifier, prepare_texts
    from .label_adapter import AnswerLabelMap
    if llama_model is None:
        from transformers import AutoModelForCausalLM
        llama_model = AutoModelForCausalLM.from_pretrained(gen_cfg["llama_model"], torch_dtype=torch.float16,
                                                           local_files_only=True).to(encoder.device).eval()
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(gen_cfg.get("llama_tokenizer", gen_cfg["llama_model"]), local_files_only=True,
                                                  use_fast=False)
    for p_ in llama_model.parameters():                      # frozen decoder, base_model.py:236-238
        p_.requires_grad = False
    prompt = gen_cfg.get("prompt") or prepare_texts(["<Img><ImageHere></Img> " + gen_cfg.get("question", "")])[0]
    answers = list(gen_cfg.get("answers", ()) or ())
    from .label_adapter import normalize_answer
    distinct = len({normalize_answer(a) for a in answers})
    if distinct == 0:
        raise ValueError("model.generate.answers is empty: every generated answer would map to the non-certifiable \"other\" class "
                         "and every certify / predict call would return ABSTAIN")
    if distinct > num_classes - 1:
        raise ValueError(f"model.generate.answers has {distinct} distinct normalised answers but run.smoothing.num_classes = "
                         f"{num_classes} leaves room for {num_classes - 1} (the last class id is \"other\"): raise num_classes "
                         "or shorten the vocabulary -- truncating it would move real answers into the non-certifiable bucket")
    label_map = AnswerLabelMap(num_classes, answers, frozen=True)
    return MiniGPT4Classifier(encoder, llama_model, tokenizer, prompt, label_map,
                              max_new_tokens=int(gen_cfg.get("max_new_tokens", 20)), max_batch=encoder.max_batch,
                              decode=gen_cfg.get("decode", "hf"), prefill_linear=gen_cfg.get("prefill_linear", "torch"))


def build_classifier(model_cfg, num_classes, max_batch, device_index, tokenizer=None, llama_model=None):
    from ..classifier import HipClassifier
    generating = "generate" in model_cfg
    clf = HipClassifier(mode="encode_img" if generating else model_cfg.get("mode", "vit_head"), num_classes=num_classes,
                        max_batch=max_batch, device=device_index, **model_cfg.get("dims", {}))
    path = model_cfg.get("weights")
    if path:
        partial = bool(model_cfg.get("partial_weights", False))
        if partial:                                   # e.g. a pretrained encoder + the build-side head left synthetic
            clf.init_synthetic(seed=int(model_cfg.get("seed", 0)))
        state = torch.load(path, map_location="cpu", weights_only=True)
        # the build-side label head is not part of MiniGPT-4: a generating classifier never evaluates it (its classes are
        # the answer vocabulary), so it is neither required from nor loaded out of the checkpoint
        names = [n for n in clf.weight_names() if not (generating and n.startswith("head."))]
        state, _ = prepare_state_dict(state, names, allow_partial=partial)
        clf.load_state_dict(state, strict=False)
    else:
        clf.init_synthetic(seed=int(model_cfg.get("seed", 0)))
    if generating:
        return build_generating_classifier(clf, model_cfg["generate"], num_classes, tokenizer, llama_model)
    return clf


@registry.register_agent("image_text_certify")
class MiniGPT4CertifyAgent(BaseAgent, CertifyLoop):
    mode = "certify"

    def __init__(self, dataset=None, classifier=None, tokenizer=None, llama_model=None):
        super().__init__()
        self.dataset = dataset
        self.classifier = classifier
        self.tokenizer = tokenizer              # optional overrides of what model.generate.{llama_tokenizer,llama_model} would load
        self.llama_model = llama_model
        self.result = None

    def run(self):
        cfg = self.config
        sm = dict(cfg["run"]["smoothing"])
        local = int(os.environ.get("LOCAL_RANK", "0"))
        clf = self.classifier or build_classifier(cfg.get("model", {}), sm["num_classes"], sm["batch_size"], local,
                                                  self.tokenizer, self.llama_model)
        self._model = clf
        self._device = getattr(clf, "device", None)
        # a generating classifier's "other" bucket (answers outside the vocabulary) is not a class of the certificate
        label_map = getattr(clf, "label_map", None)
        smooth = Smooth(clf, sm["num_classes"], sm["sigma"], seed=int(cfg["run"].get("seed", 0)),
                        non_certifiable=(label_map.other_id,) if label_map is not None else ())
        data = self.dataset
        if data is None:
            d = cfg.get("data", {})
            img = getattr(getattr(clf, "encoder", clf), "chw", (3, 224, 224))[1]
            data = synthetic_dataset(int(d.get("num_images", 10)), img, int(d.get("seed", 1234)), self._device, sm["num_classes"])
        out_dir = cfg["run"].get("output_dir")
        log = None
        if out_dir and int(os.environ.get("RANK", "0")) == 0:
            os.makedirs(out_dir, exist_ok=True)
            log = os.path.join(out_dir, f"{self.mode}.tsv")
        self.certify_dataset(sm, data, smooth, log, mode=self.mode)
        self.result = self.summary(tuple(sm.get("radii", (0.0, 0.25, 0.5, 1.0))))
        if log:
            with open(os.path.join(out_dir, f"{self.mode}_summary.json"), "w") as f:
                json.dump(self.result, f)

    def finalize(self):
        return self.result
