"""Black-box attack evaluation agent -- BASELINE configs[4] ("8-step RGF perturbation x smoothed predict N=100").

The reference describes this stage in prose only (README.md:62-64,108-120: targeted black-box attack success rate against
the smoothed model) and ships no code or config for it, so both the schedule (certifiedgpt_amd/rgf.py) and the keys below
are build-side.  Per (image, label): smoothed prediction of the clean image, RGF attack towards a target class
(default: (label + 1) mod K), smoothed prediction of the adversarial image; the summary reports the attack success rate
(adversarial prediction == target), the flip rate and the throughput.

    run: {agent: image_text_attack_eval, output_dir: ..., seed: 0,
          smoothing: {sigma: 0.5, n: 100, alpha: 0.001, batch_size: 100, num_classes: 1000},
          attack: {steps: 8, num_dirs: 1, delta: 0.5, lr: 0.05, eps: 0.25, seed: 1234, targeted: true}}
"""
import json
import os
import time

from ..smoothing import Smooth
from .base import BaseAgent
from .minigpt4_certify_agent import build_classifier, synthetic_dataset
from .registry import registry


@registry.register_agent("image_text_attack_eval")
class MiniGPT4AttackEvalAgent(BaseAgent):
    def __init__(self, dataset=None, classifier=None):
        super().__init__()
        self.dataset, self.classifier = dataset, classifier
        self.records, self.result = [], None

    def run(self):
        from ..rgf import RGFAttack
        cfg = self.config
        sm = dict(cfg["run"]["smoothing"])
        at = dict(cfg["run"].get("attack", {}))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        clf = self.classifier or build_classifier(cfg.get("model", {}), sm["num_classes"], sm["batch_size"], local)
        self._model, self._device = clf, getattr(clf, "device", None)
        K = sm["num_classes"]
        smooth = Smooth(clf, K, sm["sigma"], seed=int(cfg["run"].get("seed", 0)))
        attack = RGFAttack(smooth, steps=int(at.get("steps", 8)), num_dirs=int(at.get("num_dirs", 1)), delta=float(at.get("delta", 0.5)),
                           lr=float(at.get("lr", 0.05)), eps=float(at.get("eps", 0.25)), dir_seed=int(at.get("seed", 1234)))
        targeted = bool(at.get("targeted", True))
        data = self.dataset
        if data is None:
            d = cfg.get("data", {})
            img = getattr(clf, "chw", (3, 224, 224))[1]
            data = synthetic_dataset(int(d.get("num_images", 4)), img, int(d.get("seed", 1234)), self._device, K)
        out_dir = cfg["run"].get("output_dir")
        f = None
        if out_dir and int(os.environ.get("RANK", "0")) == 0:
            os.makedirs(out_dir, exist_ok=True)
            f = open(os.path.join(out_dir, "attack_eval.tsv"), "w")
            print("idx\tlabel\ttarget\tclean_predict\tadv_predict\tsuccess\tshare_first\tshare_last\ttime", file=f, flush=True)
        for idx, (x, label) in enumerate(data):
            t0 = time.perf_counter()
            clean = smooth.predict(x, sm["n"], sm["alpha"], sm["batch_size"])
            target = (int(label) + 1) % K if targeted else int(label)
            _, adv, hist = attack.attack(x, target, sm["n"], sm["alpha"], sm["batch_size"], targeted=targeted)
            dt = time.perf_counter() - t0
            success = int(adv == target) if targeted else int(adv != label and adv != Smooth.ABSTAIN)
            rec = dict(idx=idx, label=int(label), target=target, clean_predict=int(clean), adv_predict=int(adv), success=success,
                       share_first=hist[0], share_last=hist[-1], time=dt)
            self.records.append(rec)
            if f:
                print("\t".join(str(rec[k]) if not isinstance(rec[k], float) else f"{rec[k]:.4f}" for k in
                                ("idx", "label", "target", "clean_predict", "adv_predict", "success", "share_first", "share_last", "time")),
                      file=f, flush=True)
        if f:
            f.close()
        n = max(len(self.records), 1)
        fw = attack.forwards_per_image(sm["n"]) + sm["n"]
        total_t = max(sum(r["time"] for r in self.records), 1e-9)
        self.result = {"images": len(self.records), "attack_success_rate": sum(r["success"] for r in self.records) / n,
                       "flip_rate": sum(r["adv_predict"] != r["clean_predict"] for r in self.records) / n,
                       "images_per_s": len(self.records) / total_t, "forwards_per_image": fw,
                       "forwards_per_s": len(self.records) * fw / total_t}
        if out_dir and int(os.environ.get("RANK", "0")) == 0:
            with open(os.path.join(out_dir, "attack_eval_summary.json"), "w") as g:
                json.dump(self.result, g)

    def finalize(self):
        return self.result
