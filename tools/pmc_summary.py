"""Collects the PMC evidence behind bench.py's `roofline.traffic` and DESIGN.md: runs the bench command under rocprofv3 --pmc in
SEPARATE passes (one counter group each, with --kernel-trace only: the pool refuses --pmc combined with the API traces) and
writes per-kernel averages to profiles/<round>/pmc_summary.json.  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE is doubled on gfx950.  Run on the GPU box:   python tools/pmc_summary.py [round_dir]"""
import collections, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_summary")
GROUPS = [["FETCH_SIZE"], ["WRITE_SIZE"], ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"], ["TCC_HIT_sum", "TCC_MISS_sum"]]
KERNELS = {"layernorm": "layernorm4_", "gemm_f16out": "gemm9_f16_kernel<0>", "attention": "attention_kernel<88", "fc1": "gemm9_f16_kernel<1>"}
BENCH_ARGS = ["--steps", "5", "--warmup", "1", "--no-cpu-baseline"]   # the default bench command (batches of 255, 255, 255, 235 samples + the 200-sample warm-up)


def git_head():
    """Commit of the tree the counters were taken on (the GPU box receives a snapshot without .git: the caller passes it in CGPT_GIT_HEAD)."""
    if os.environ.get("CGPT_GIT_HEAD"):
        return os.environ["CGPT_GIT_HEAD"]
    try:
        return subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        return None



def main():
    round_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04")
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", CGPT_BENCH_NO_SUSTAINED="1")   # (the MFMA-only yardstick loop is not part of the profiled work)
    res = {k: {} for k in KERNELS}
    for i, grp in enumerate(GROUPS):
        d = os.path.join(OUT, f"p{i}")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *grp, "-d", d, "-o", "r", "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), *BENCH_ARGS]
        with open(os.path.join(OUT, f"p{i}.log"), "w") as log:
            subprocess.run(cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT, check=False)
        f = glob.glob(os.path.join(d, "**", "r_counter_collection.csv"), recursive=True)
        t = glob.glob(os.path.join(d, "**", "r_kernel_trace.csv"), recursive=True)
        if not f or not t:
            print("pass", i, "produced no counters; see", os.path.join(OUT, f"p{i}.log"), file=sys.stderr)
            continue
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(t[0])):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
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
    os.makedirs(round_dir, exist_ok=True)
    import hashlib
    lib = os.path.join(ROOT, "certifiedgpt_amd", "libcgpt.so")
    res["_meta"] = {"git_head": git_head(), "command": "rocprofv3 --kernel-trace --pmc <group> -- python bench.py " + " ".join(BENCH_ARGS),
                    "counter_groups": GROUPS, "batch_size_per_gpu": 255,
                    "libcgpt_sha256_16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None,
                    "note": "per-launch averages over every launch of the kernel in the run; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 "
                            "(gfx950 correction, MI355X_MICROARCH.md); Infinity-Cache hits are counted"}
    with open(os.path.join(round_dir, "pmc_summary.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: {c: v for c, v in r.items() if c in ("launches", "hbm_read_bytes_corrected", "hbm_write_bytes", "clock_ghz", "mfma_busy_frac", "l2_hit_rate", "kernel_name")} for k, r in res.items() if k != "_meta"}, indent=1))


if __name__ == "__main__":
    main()
