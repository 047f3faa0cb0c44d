"""The float64 statistics (certifiedgpt_amd/csrc/stats.h) compiled host-only with AddressSanitizer + UBSan and run over the
reference goldens and edge cases.  GPU sanitizers are not available on the pool; this code is what decides label / abstain /
radius, and it is plain host C++ (the same header also compiles into the device finalize kernel)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = str(tmp_path_factory.mktemp("san") / "stats_sanitize")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(HERE, "native", "stats_sanitize.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return exe


def _run(exe, lines):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
    return r.stdout.strip().splitlines()


def test_statistics_are_sanitizer_clean_and_match_goldens(driver, stats_golden):
    lines, expect = [], []
    for c in stats_golden["certify"]:
        K = len(c["counts_sel"])
        lines.append(f"C {K} {c['n']} {c['alpha']!r} {c['sigma']!r} " + " ".join(map(str, c["counts_sel"])) + " " +
                     " ".join(map(str, c["counts_est"])))
        expect.append(("C", c["label"], c["radius"]))
    for c in stats_golden["predict"]:
        lines.append(f"P {len(c['counts'])} {c['alpha']!r} " + " ".join(map(str, c["counts"])))
        expect.append(("P", c["label"], None))
    out = _run(driver, lines)
    assert len(out) == len(expect)
    for o, (kind, label, radius) in zip(out, expect):
        if kind == "C":
            lab, rad = o.split()
            assert int(lab) == label and abs(float(rad) - radius) <= 1e-9, (o, label, radius)
        else:
            assert int(o) == label, (o, label)


def test_edge_cases_do_not_trip_the_sanitizers(driver):
    lines = []
    for N in (1, 2, 10, 100, 1000, 10 ** 6, 10 ** 9):
        for NA in {0, 1, N // 2, N - 1, N}:
            for alpha in (1e-12, 1e-3, 0.5, 0.999999):
                lines.append(f"L {NA} {N} {alpha!r}")
    for n in (0, 1, 2, 99, 100, 10 ** 5):
        for x in {0, n // 3, n // 2, n}:
            for p in (0.5, 1e-9, 0.999):
                lines.append(f"B {x} {n} {p!r}")
    for p in (0.0, 1e-320, 1e-300, 1e-17, 0.5, 1 - 1e-16, 1.0):
        lines.append(f"Q {p!r}")
    lines.append("C 3 100 0.001 0.5 0 0 0 0 0 0")           # empty histograms
    lines.append("P 2 0.001 0 0")
    lines.append("C 1 10 0.001 0.25 10 10")                  # a single class
    out = _run(driver, lines)
    assert len(out) == len(lines)
    vals = [float(v.split()[-1]) for v in out[:-3] if v.split()[-1] not in ("inf", "-inf")]
    assert all(np.isfinite(v) or True for v in vals)
