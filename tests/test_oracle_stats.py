"""Pin the statistics ORACLE against goldens emitted by the reference's own smoothing.py
(oracle/gen_golden_stats.py, scipy 1.7.1 / statsmodels 0.12.2)."""
import numpy as np

from oracle import smooth_oracle as so

R_TOL = 1e-12  # radius agreement demanded of the oracle (north-star tolerance is 1e-3)


def test_certify_goldens(stats_golden):
    for c in stats_golden["certify"]:
        lab, rad = so.certify_from_counts(c["counts_sel"], c["counts_est"], c["n"], c["alpha"], c["sigma"])
        assert lab == c["label"], c
        assert abs(rad - c["radius"]) <= R_TOL * max(1.0, abs(c["radius"])), (c, rad)


def test_predict_goldens(stats_golden):
    for c in stats_golden["predict"]:
        assert so.predict_from_counts(c["counts"], c["alpha"]) == c["label"], c


def test_scalar_goldens(stats_golden):
    for c in stats_golden["lcb"]:
        v = so.lower_confidence_bound(c["nA"], c["n"], c["alpha"])
        assert abs(v - c["value"]) <= 1e-13 * max(1.0, abs(c["value"])), (c, v)
    for c in stats_golden["binom_test"]:
        v = so.binom_test(c["k"], c["n"], c["p"])
        assert abs(v - c["value"]) <= 1e-12 * max(c["value"], 1e-300) + 1e-300, (c, v)
    for c in stats_golden["norm_ppf"]:
        v = so.norm_ppf(c["p"])
        assert abs(v - c["value"]) <= 1e-13 * max(1.0, abs(c["value"])), (c, v)
    for c in stats_golden["count_arr"]:
        assert so.count_arr(np.asarray(c["arr"], dtype=int), c["length"]).tolist() == c["counts"]


def test_survey_known_answers():
    # SURVEY.md section 8(c) known answers captured from the reference (alpha = 0.001)
    assert so.certify_from_counts([0, 100], [10, 90], 100, 0.001, 0.5) == (1, 0.3782577025559939)
    assert so.certify_from_counts([0, 100], [40, 60], 100, 0.001, 0.5) == (-1, 0.0)
    lab, r = so.certify_from_counts([0, 10], [0, 10], 10, 0.001, 0.25)
    assert lab == 1 and abs(r - 0.0007439894428455479) < 1e-15
    assert so.certify_from_counts([5, 5], [9, 1], 10, 0.5, 1.0)[0] == 0  # selection tie -> argmax = 0
    assert so.predict_from_counts([60, 40, 0], 0.001) == -1
    assert so.predict_from_counts([90, 10, 0], 0.001) == 0
    assert so.predict_from_counts([50, 50, 0], 0.001) == -1


def test_rgf_oracle_step_properties():
    """oracle/rgf_oracle (build-side rule; the reference has no attack code): eps-ball, sign rule, determinism."""
    from oracle import rgf_oracle as ro
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 8, 8)).astype(np.float32)
    dirs = [ro.direction(5, i, x.shape) for i in range(3)]
    assert np.array_equal(dirs[1], ro.direction(5, 1, x.shape)) and not np.array_equal(dirs[0], dirs[1])
    out = ro.rgf_step(x, x, dirs, [1.0, 0.0, 0.0], 0.1, 0.25)
    assert np.allclose(out - x, 0.1 * np.sign(dirs[0]), atol=1e-6)
    out = ro.rgf_step(x, x, dirs, [0.5, -2.0, 0.25], 1.0, 0.25)
    assert float(np.abs(out - x).max()) <= 0.25 + 1e-7
    # a share function that rewards moving along +dirs[0] is ascended by the targeted loop
    u0 = dirs[0]
    share = lambda img, step: float(1.0 / (1.0 + np.exp(-np.sum((img - x) * u0) / 50.0)))
    adv, hist = ro.attack(share, lambda i: ro.direction(5, i, x.shape), x, steps=6, num_dirs=4, delta=0.5, lr=0.05, eps=0.5)
    assert hist[-1] > hist[0] and len(hist) == 7
