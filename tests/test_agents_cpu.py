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
