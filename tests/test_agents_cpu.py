"""Plugin contract of the certify / predict agents (reference: launch.py:97-107, agents/__init__.py:14-21,
common/registry.py:54-80) exercised on the CPU with a stand-in engine -- host logic only."""
import json

import numpy as np
import pytest
import torch

from certifiedgpt_amd.agents import registry, setup_agent, BaseAgent
from certifiedgpt_amd.agents import minigpt4_certify_agent, minigpt4_predict_agent  # noqa: F401  (import registers, launch.py:97-99)


class Engine:
    """sample_counts protocol: class 2 always wins 90 %."""
    num = 5
    chw = (3, 8, 8)
    device = "cpu"

    def eval(self):
        return self

    def sample_counts(self, x, first_sample, num, batch_size, sigma, seed):
        idx = np.arange(first_sample, first_sample + num)
        labels = np.where(idx % 10 == 0, 1, 2)
        return torch.from_numpy(np.bincount(labels, minlength=self.num).astype(np.int64))


def _config(tmp_path, agent):
    return {"run": {"agent": agent, "output_dir": str(tmp_path), "seed": 0,
                    "smoothing": {"sigma": 0.5, "n0": 100, "n": 100, "alpha": 0.001, "batch_size": 50, "num_classes": 5,
                                  "radii": [0.0, 0.25, 0.5]}}}


@pytest.mark.parametrize("agent_name", ["image_text_certify", "image_text_predict"])
def test_agent_contract(tmp_path, agent_name):
    cfg = _config(tmp_path, agent_name)
    registry.register("configuration", cfg)
    cls = registry.get_agent_class(agent_name)
    assert issubclass(cls, BaseAgent)
    agent = setup_agent(cfg)                       # agents/__init__.py:14-21
    agent.classifier = Engine()
    agent.dataset = [(torch.zeros(3, 8, 8), 2), (torch.zeros(3, 8, 8), 1), (torch.zeros(3, 8, 8), 2)]
    agent.run()
    res = agent.finalize()
    assert res["images"] == 3 and res["abstain_rate"] == 0.0 and abs(res["accuracy"] - 2 / 3) < 1e-12
    mode = "certify" if agent_name.endswith("certify") else "predict"
    lines = open(tmp_path / f"{mode}.tsv").read().strip().splitlines()
    assert lines[0].split("\t")[:3] == ["idx", "label", "predict"] and len(lines) == 4
    if mode == "certify":
        # nA = 90 of 100 at sigma .5, alpha .001 -> R = 0.3782577... (SURVEY.md 8(c) known answer)
        assert abs(float(lines[1].split("\t")[3]) - 0.378258) < 1e-6
        assert abs(res["certified_acc@0.25"] - 2 / 3) < 1e-12 and res["certified_acc@0.5"] == 0.0
    assert json.load(open(tmp_path / f"{mode}_summary.json"))["images"] == 3


def test_grouped_certify_gives_the_same_records(tmp_path):
    """images_per_pass > 1 routes through Smooth.certify_many; an engine without the several-images protocol falls back to
    consecutive certify calls, so the log is the same as the one-by-one loop."""
    outs = []
    for group in (1, 2):
        cfg = _config(tmp_path / f"g{group}", "image_text_certify")
        cfg["run"]["smoothing"]["images_per_pass"] = group
        registry.register("configuration", cfg)
        agent = setup_agent(cfg)
        agent.classifier = Engine()
        agent.dataset = [(torch.zeros(3, 8, 8), 2), (torch.zeros(3, 8, 8), 1), (torch.zeros(3, 8, 8), 2)]
        agent.run()
        outs.append([(r["idx"], r["label"], r["predict"], r["radius"]) for r in agent.records])
    assert outs[0] == outs[1] and len(outs[0]) == 3


def test_image_sharded_mode_gives_the_same_records_and_bad_mode_is_refused(tmp_path):
    """run.smoothing.shard = images routes through Smooth.certify_images (whole images per rank; one rank here): the log
    equals the one-by-one loop's.  An unknown mode is an error, not a silent default."""
    outs = []
    for shard in ("samples", "images"):
        cfg = _config(tmp_path / shard, "image_text_certify")
        cfg["run"]["smoothing"].update(images_per_pass=2, shard=shard)
        registry.register("configuration", cfg)
        agent = setup_agent(cfg)
        agent.classifier = Engine()
        agent.dataset = [(torch.zeros(3, 8, 8), 2), (torch.zeros(3, 8, 8), 1), (torch.zeros(3, 8, 8), 2)]
        agent.run()
        outs.append([(r["idx"], r["label"], r["predict"], r["radius"]) for r in agent.records])
    assert outs[0] == outs[1] and len(outs[0]) == 3
    preds = []
    for shard in ("samples", "images"):                      # the predict agent: Smooth.predict one by one | Smooth.predict_images
        cfg = _config(tmp_path / ("p" + shard), "image_text_predict")
        cfg["run"]["smoothing"].update(images_per_pass=2, shard=shard)
        registry.register("configuration", cfg)
        agent = setup_agent(cfg)
        agent.classifier = Engine()
        agent.dataset = [(torch.zeros(3, 8, 8), 2), (torch.zeros(3, 8, 8), 1), (torch.zeros(3, 8, 8), 2)]
        agent.run()
        preds.append([(r["idx"], r["label"], r["predict"]) for r in agent.records])
    assert preds[0] == preds[1] and [p[2] for p in preds[0]] == [2, 2, 2]
    cfg = _config(tmp_path / "bad", "image_text_certify")
    cfg["run"]["smoothing"]["shard"] = "pixels"
    registry.register("configuration", cfg)
    agent = setup_agent(cfg)
    agent.classifier = Engine()
    agent.dataset = [(torch.zeros(3, 8, 8), 2)]
    with pytest.raises(ValueError):
        agent.run()


def test_duplicate_registration_rejected():
    with pytest.raises(KeyError):
        @registry.register_agent("image_text_certify")
        class Again(BaseAgent):
            pass


def test_label_adapter_matches_reference_normaliser_goldens():
    import os
    from conftest import GOLDEN
    from certifiedgpt_amd.agents.label_adapter import normalize_answer, AnswerLabelMap
    cases = json.load(open(os.path.join(GOLDEN, "label_adapter_golden.json")))["cases"]
    assert len(cases) >= 30
    for c in cases:
        assert normalize_answer(c["answer"]) == c["normalized"], c
    m = AnswerLabelMap(4)
    assert [m(t) for t in ("Yes", "yes.", "No", "a cat", "the cat", "dog", "bird")] == [0, 0, 1, 2, 2, 3, 3]   # cap -> "other"
    logits = m.one_hot_logits(["yes", "no"])
    assert logits.shape == (2, 4) and logits[0, 0] == 1 and logits[1, 1] == 1


def test_label_map_frozen_vocabulary_and_other_bucket():
    from certifiedgpt_amd.agents.label_adapter import AnswerLabelMap
    m = AnswerLabelMap(4, ["yes", "no"])                      # a vocabulary given up front freezes the map
    assert m.frozen and m.other_id == 3
    assert [m(t) for t in ("Yes.", "NO", "maybe", "a maybe")] == [0, 1, 3, 3] and m.answers == ["yes", "no"]
    g = AnswerLabelMap(3)                                      # exploratory: grows, then overflows into "other"
    assert [g(t) for t in ("b", "a", "b", "c")] == [0, 1, 0, 2] and not g.frozen
    assert g.freeze()("zzz") == 2


def test_checkpoint_with_wrong_names_is_an_error_not_a_silent_zero_model():
    from certifiedgpt_amd.agents.minigpt4_certify_agent import prepare_state_dict
    names = ["visual_encoder.cls_token", "visual_encoder.blocks.0.norm1.weight", "ln_vision.weight", "head.weight"]
    full = {n: torch.zeros(1) for n in names}
    out, missing = prepare_state_dict({"model": dict(full, **{"llama_model.x": torch.zeros(1)})}, names)     # BLIP-2 style wrapper
    assert sorted(out) == sorted(names) and not missing
    out, _ = prepare_state_dict({"model_state_dict": full}, names)
    assert sorted(out) == sorted(names)
    # raw EVA checkpoint: no "visual_encoder." prefix (eva_vit.py:445-456)
    raw = {"cls_token": torch.zeros(1), "blocks.0.norm1.weight": torch.zeros(1), "ln_vision.weight": torch.zeros(1), "head.weight": torch.zeros(1)}
    out, _ = prepare_state_dict(raw, names)
    assert sorted(out) == sorted(names)
    with pytest.raises(KeyError, match="missing"):
        prepare_state_dict({"encoder.cls": torch.zeros(1), "ln_vision.weight": torch.zeros(1)}, names)
    out, missing = prepare_state_dict({"ln_vision.weight": torch.zeros(1)}, names, allow_partial=True)
    assert list(out) == ["ln_vision.weight"] and len(missing) == 3


def test_smooth_abstains_on_non_certifiable_top_class():
    from certifiedgpt_amd import Smooth
    s = Smooth(Engine(), 5, 0.5, non_certifiable=(2,))        # class 2 wins 90 % of the votes but is the "other" bucket
    assert s.certify(torch.zeros(3, 8, 8), 100, 100, 0.001, 50) == (Smooth.ABSTAIN, 0.0)
    assert s.predict(torch.zeros(3, 8, 8), 100, 0.001, 50) == Smooth.ABSTAIN
    t = Smooth(Engine(), 5, 0.5)
    lab, rad = t.certify(torch.zeros(3, 8, 8), 100, 100, 0.001, 50)
    assert lab == 2 and abs(rad - 0.3782577025559939) < 1e-9
    p = t.predict(torch.zeros(3, 8, 8), 100, 0.001, 50)
    assert p == 2 and isinstance(p, np.int64)                 # smoothing.py:79 returns an int64 ndarray element


def test_vqa_accuracy_matches_the_references_evaluation_loop():
    import os
    from conftest import GOLDEN
    from certifiedgpt_amd.agents.label_adapter import vqa_accuracy
    cases = json.load(open(os.path.join(GOLDEN, "label_adapter_golden.json")))["vqa_accuracy"]
    assert len(cases) >= 25
    for c in cases:
        assert abs(vqa_accuracy(c["answer"], c["gt_answers"]) - c["accuracy"]) <= 1e-6, c


def test_certify_agent_scores_vqa_samples_by_answer_text(tmp_path):
    """A dataset item whose label is a list of human answers is scored with the VQA accuracy of the certified class's answer text."""
    from certifiedgpt_amd.agents.label_adapter import AnswerLabelMap

    class TextEngine(Engine):
        label_map = AnswerLabelMap(5, ["a cat", "no", "2 dogs", "yes"])          # class 2 ("2 dogs") wins 90 % of the votes

    cfg = _config(tmp_path, "image_text_certify")
    registry.register("configuration", cfg)
    agent = setup_agent(cfg)
    agent.classifier = TextEngine()
    gts = ["two dogs"] * 2 + ["2 dogs"] * 2 + ["dogs"] * 6
    agent.dataset = [(torch.zeros(3, 8, 8), gts), (torch.zeros(3, 8, 8), ["yes"] * 10)]
    agent.run()
    recs = agent.records
    assert recs[0]["answer"] == "2 dogs" and abs(recs[0]["correct"] - 0.6) < 1e-9    # "2 dogs" matches 2 of the processed answers
    assert recs[1]["correct"] == 0.0
    res = agent.finalize()
    assert abs(res["accuracy"] - 0.3) < 1e-9 and abs(res["certified_acc@0.25"] - 0.3) < 1e-9


def test_generating_classifier_rejects_an_empty_or_oversized_vocabulary():
    """An empty vocabulary would make every call ABSTAIN silently; one longer than num_classes - 1 would be truncated silently."""
    import pytest
    from certifiedgpt_amd.agents.minigpt4_certify_agent import build_generating_classifier

    class Enc:
        max_batch = 4

    class LM:
        def parameters(self):
            return iter(())

    for answers, K, word in (((), 5, "empty"), (None, 5, "empty"), (["a cat", "dog", "the dog", "two", "2", "bus"], 4, "distinct")):
        with pytest.raises(ValueError) as e:
            build_generating_classifier(Enc(), {"prompt": "<ImageHere> q", "answers": answers}, K, tokenizer=object(), llama_model=LM())
        assert word in str(e.value)
    clf = build_generating_classifier(Enc(), {"prompt": "<ImageHere> q", "answers": ["a dog", "dog", "two", "2"]}, 3,
                                      tokenizer=object(), llama_model=LM())          # 2 distinct normalised answers + other
    assert clf.label_map.frozen and clf.label_map.answers == ["dog", "2"]
