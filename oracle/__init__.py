"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path (leodesouza/certifiedGPT:
randomized_smoothing/smoothing.py, graphs/models/minigpt4/models/{eva_vit,
Qformer,minigpt4,base_model}.py).  Nothing in the product package
(`certifiedgpt_amd/`) imports, links or executes anything under `oracle/`.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may use it, and only as the checker / the timed CPU baseline.

Pinning (see DESIGN.md "Oracle"):
  * statistics: pinned against tests/golden/stats_golden.json, produced by
    executing the reference's own smoothing.py under scipy 1.7.1 /
    statsmodels 0.12.2 (oracle/gen_golden_stats.py).
  * model: pinned against tests/golden/model_golden.npz, produced by
    executing the reference's own eva_vit.py / Qformer.py classes loaded by
    file path (oracle/gen_golden_model.py).
"""
