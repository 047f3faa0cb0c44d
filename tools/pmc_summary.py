"""Collects the profile evidence behind bench.py's `roofline` object and DESIGN.md, so that the roofline can be checked from profiles/ alone.

Profiles ONE command -- `python3 bench.py --steps 51 --warmup 51 --no-cpu-baseline` with CGPT_BENCH_ONLY_TIMED=1 -- whose GPU work is
nothing but FULL classifier batches: 51 images x 200 forwards = 40 batches of 255 samples per `Smooth.certify_many` call (no ragged
last batch, no single-image leg, no yardstick loop), so every launch of the dominant kernel (fc1 + GELU: M = 255 x 257 = 65 535 rows,
N = 6144, K = 1408) does the same work and per-launch averages mean one thing.

  pass "stats":  rocprofv3 --kernel-trace --stats                     -> <round>/bench_kernel_stats.csv, <round>/bench_under_rocprofv3.json
  passes p0..p3: rocprofv3 --kernel-trace --pmc <one counter group>   (separate passes; the pool refuses --pmc with the API traces)

and writes <round>/pmc_summary.json: per kernel {launches, avg / min / max us, counters}, plus the row bench.py reads:
  "fc1_full_batch": {batch_samples, launches, avg_us (stats pass), min_us, max_us, flop_per_launch, tflops, frac_of_peak,
                     bench_frac_same_run (HIP events inside bench.py, same process as the stats pass), hbm_read_bytes_corrected,
                     hbm_write_bytes, algorithmic_bytes, mfma_busy_frac, clock_ghz, l2_hit_rate}
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled on gfx950.
Run on the GPU box:   python3 tools/pmc_summary.py [out_dir]   (default gpurun_out/pmc_summary/out -- only gpurun_out/ comes back from the
box; copy its three files into profiles/<round>/ afterwards).  The program after `--` is python3 itself: no re-exec hop.  The per-pass
trace CSVs (tens of MB) are deleted once parsed."""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_summary")
GROUPS = [["FETCH_SIZE"], ["WRITE_SIZE"], ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"], ["TCC_HIT_sum", "TCC_MISS_sum"]]
KERNELS = {"layernorm": "layernorm4_", "gemm_f16out": "gemm9_f16_kernel<0>", "attention": "attention_kernel<88", "fc1": "gemm9_f16_kernel<1>"}
BATCH = 255
BENCH_ARGS = ["--steps", "51", "--warmup", "51", "--no-cpu-baseline"]   # 2 x 40 full batches of 255 samples, nothing else
FC1_FLOP = 2.0 * BATCH * 257 * 6144 * 1408
PEAK = 2500e12


def git_head():
    """Commit of the tree the counters were taken on (the GPU box receives a snapshot without .git: the caller passes it in CGPT_GIT_HEAD)."""
    if os.environ.get("CGPT_GIT_HEAD"):
        return os.environ["CGPT_GIT_HEAD"]
    try:
        return subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        return None


def durations(trace_csv):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace_csv)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return dur


def parse_pass(d, grp, res):
    f = glob.glob(os.path.join(d, "**", "r_counter_collection.csv"), recursive=True)
    t = glob.glob(os.path.join(d, "**", "r_kernel_trace.csv"), recursive=True)
    if not f or not t:
        return False
    dur = durations(t[0])
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, pat in KERNELS.items():
        names = [n for n in vals if pat in n]
        if not names:
            continue
        n = names[0]
        res[key]["kernel_name"] = n
        res[key]["launches"] = len(dur[n])
        for c in grp:
            if vals[n][c]:
                res[key][c] = sum(vals[n][c]) / len(vals[n][c])
                res[key]["avg_us_" + c] = sum(dur[n]) / len(dur[n]) / 1e3
    return True


def derive(res):
    for key, r in res.items():
        if "FETCH_SIZE" in r:
            r["hbm_read_bytes_corrected"] = r["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in r:
            r["hbm_write_bytes"] = r["WRITE_SIZE"] * 1024
        if "GRBM_GUI_ACTIVE" in r:
            cyc = r["GRBM_GUI_ACTIVE"] / 8                      # summed over the 8 XCDs
            r["clock_ghz"] = cyc / r["avg_us_GRBM_GUI_ACTIVE"] / 1e3
            r["mfma_busy_frac"] = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024)   # 256 CUs x 4 SIMDs
        if "TCC_HIT_sum" in r:
            r["l2_hit_rate"] = r["TCC_HIT_sum"] / max(r["TCC_HIT_sum"] + r["TCC_MISS_sum"], 1.0)


def stats_pass(round_dir, env, res):
    """rocprofv3 --kernel-trace --stats of the same command: the per-kernel average durations the roofline is checked against, and the
    bench line (HIP events inside the process) of the very same run."""
    d = os.path.join(OUT, "stats")
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "r", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), *BENCH_ARGS]
    log = os.path.join(OUT, "stats.log")
    with open(log, "w") as lf:
        subprocess.run(cmd, cwd=ROOT, env=env, stdout=lf, stderr=subprocess.STDOUT, check=False)
    line = None
    for l in open(log, errors="replace"):
        if l.startswith("{") and '"metric"' in l:
            line = json.loads(l)
    ks = glob.glob(os.path.join(d, "**", "r_kernel_stats.csv"), recursive=True)
    t = glob.glob(os.path.join(d, "**", "r_kernel_trace.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(round_dir, "bench_kernel_stats.csv"))
    if line is not None:
        with open(os.path.join(round_dir, "bench_under_rocprofv3.json"), "w") as f:
            json.dump(line, f)
    if not t:
        print("stats pass produced no kernel trace; see", log, file=sys.stderr)
        return None
    dur = durations(t[0])
    for key, pat in KERNELS.items():
        names = [n for n in dur if pat in n]
        if names:
            v = dur[names[0]]
            res[key].update({"kernel_name": names[0], "stats_launches": len(v), "stats_avg_us": sum(v) / len(v) / 1e3,
                             "stats_min_us": min(v) / 1e3, "stats_max_us": max(v) / 1e3})
    shutil.rmtree(d, ignore_errors=True)
    return line


def main():
    round_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(OUT, "out")
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(round_dir, exist_ok=True)
    # CGPT_BENCH_ONLY_TIMED: no single-image leg, no MFMA-only yardstick loop -- every classifier batch of the run is a full one
    env = dict(os.environ, TMPDIR="/tmp", CGPT_BENCH_ONLY_TIMED="1")
    res = {k: {} for k in KERNELS}
    line = stats_pass(round_dir, env, res)
    print("stats pass done", flush=True)
    for i, grp in enumerate(GROUPS):
        d = os.path.join(OUT, f"p{i}")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *grp, "-d", d, "-o", "r", "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), *BENCH_ARGS]
        with open(os.path.join(OUT, f"p{i}.log"), "w") as log:
            subprocess.run(cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT, check=False)
        if not parse_pass(d, grp, res):
            print("pass", i, "produced no counters; see", os.path.join(OUT, f"p{i}.log"), file=sys.stderr)
        shutil.rmtree(d, ignore_errors=True)
        print("pmc pass", i, grp, "done", flush=True)
    derive(res)
    fc1 = res["fc1"]
    full = {"batch_samples": BATCH, "rows": BATCH * 257, "kernel_name": fc1.get("kernel_name"),
            "launches": fc1.get("stats_launches"), "avg_us": fc1.get("stats_avg_us"), "min_us": fc1.get("stats_min_us"),
            "max_us": fc1.get("stats_max_us"), "flop_per_launch": FC1_FLOP,
            "algorithmic_bytes": BATCH * 257 * (1408 + 6144) * 2 + 6144 * 1408 * 2}
    if fc1.get("stats_avg_us"):
        full["tflops"] = FC1_FLOP / (fc1["stats_avg_us"] * 1e-6) / 1e12
        full["frac_of_peak"] = FC1_FLOP / (fc1["stats_avg_us"] * 1e-6) / PEAK
    if line is not None:
        full["bench_frac_same_run"] = line["roofline"]["frac"]
        full["bench_avg_launch_us_same_run"] = 1e3 * line["roofline"]["avg_launch_ms"]
        full["bench_in_kernel_clock_ghz_same_run"] = line["roofline"].get("in_kernel_clock_ghz")
    for k in ("hbm_read_bytes_corrected", "hbm_write_bytes", "mfma_busy_frac", "clock_ghz", "l2_hit_rate", "avg_us_FETCH_SIZE",
              "avg_us_WRITE_SIZE", "avg_us_GRBM_GUI_ACTIVE"):
        if k in fc1:
            full[k] = fc1[k]
    if "hbm_read_bytes_corrected" in full and "hbm_write_bytes" in full:
        full["traffic_over_algorithmic"] = (full["hbm_read_bytes_corrected"] + full["hbm_write_bytes"]) / full["algorithmic_bytes"]
    res["fc1_full_batch"] = full
    import hashlib
    lib = os.path.join(ROOT, "certifiedgpt_amd", "libcgpt.so")
    res["_meta"] = {"git_head": git_head(),
                    "command": "CGPT_BENCH_ONLY_TIMED=1 rocprofv3 --kernel-trace {--stats | --pmc <group>} -- python3 bench.py " + " ".join(BENCH_ARGS),
                    "counter_groups": GROUPS, "batch_size_per_gpu": BATCH,
                    "libcgpt_sha256_16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None,
                    "note": "every classifier batch of the profiled run holds 255 samples, so per-launch averages are per FULL batch; "
                            "stats_* durations come from the --stats pass (no counters), counter averages from their own --pmc pass "
                            "(a --pmc pass runs at other clocks: its avg_us_<counter> is listed beside each counter); FETCH_SIZE / "
                            "WRITE_SIZE in KiB, FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md); Infinity-Cache hits are counted"}
    with open(os.path.join(round_dir, "pmc_summary.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["fc1_full_batch"], indent=1))


if __name__ == "__main__":
    main()
