"""Kernel-level evidence for ONE bench.py workload: a `rocprofv3 --kernel-trace --stats` table of the command plus the
PMC rows of its top kernels, next to the bench line of the very same run, so that a roofline fraction can be recomputed from the files.

    python3 tools/profile_workload.py <name> [--top N] [--no-pmc] [--flop SUBSTRING=FLOP_PER_LAUNCH ...] -- <bench.py arguments>

  pass "stats":  CGPT_BENCH_ONLY_TIMED=1 rocprofv3 --kernel-trace --stats -- python3 bench.py <arguments>
  passes p0..p3: the same command with --kernel-trace --pmc <one counter group> (separate passes: the pool refuses --pmc with API traces)

Writes gpurun_out/prof_<name>/out/ (copy into profiles/<round>/<name>/):
  bench_under_rocprofv3.json   the bench line of the stats pass (HIP events inside the process)
  kernel_stats.csv             rocprofv3's own per-kernel table of that pass
  summary.json                 per kernel, by share of device time: launches, avg / min / max us, share; for the top N also read / write
                               bytes per launch (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE; KiB -> bytes), MFMA-busy share, clock, L2 hit
                               rate, and -- where a FLOP count per launch was given with --flop -- TFLOP/s and the fraction of 2 500
CGPT_BENCH_ONLY_TIMED=1: no single-image leg and no yardstick loop reach the GPU, one stream synchronisation per classifier batch (the
profiler keeps a record per dispatch in flight).  The program after `--` is python3 itself: no re-exec hop.  Trace CSVs are deleted once parsed."""
import collections, csv, glob, hashlib, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["FETCH_SIZE"], ["WRITE_SIZE"], ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"], ["TCC_HIT_sum", "TCC_MISS_sum"]]
PEAK = 2500e12


def durations(trace_csv):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace_csv)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return dur


def run_pass(out, tag, flags, bench_args, env):
    d = os.path.join(out, tag)
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace", *flags, "-d", d, "-o", "r", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), *bench_args]
    log = os.path.join(out, tag + ".log")
    with open(log, "w") as lf:
        rc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=lf, stderr=subprocess.STDOUT, check=False).returncode
    return d, log, rc


def short(name):
    """Kernel names as rocprofv3 prints them can be hundreds of characters (torch's templates): keep what identifies them."""
    n = name.replace("cgpt::(anonymous namespace)::", "").replace("void ", "")
    return n if len(n) <= 110 else n[:107] + "..."


def main():
    argv = sys.argv[1:]
    if "--" not in argv or not argv or argv[0].startswith("-"):
        print(__doc__)
        sys.exit(2)
    cut = argv.index("--")
    mine, bench_args = argv[:cut], argv[cut + 1:]
    name, top, pmc, flops = mine[0], 4, True, {}
    i = 1
    while i < len(mine):
        if mine[i] == "--top":
            top = int(mine[i + 1]); i += 2
        elif mine[i] == "--no-pmc":
            pmc = False; i += 1
        elif mine[i] == "--flop":
            k, v = mine[i + 1].rsplit("=", 1)
            flops[k] = float(v); i += 2
        else:
            raise SystemExit("unknown option " + mine[i])
    out = os.path.join(ROOT, "gpurun_out", "prof_" + name)
    res_dir = os.path.join(out, "out")
    os.makedirs(res_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", CGPT_BENCH_ONLY_TIMED="1")

    d, log, rc = run_pass(out, "stats", ["--stats"], bench_args, env)
    line = None
    for l in open(log, errors="replace"):
        if l.startswith("{") and '"metric"' in l:
            line = json.loads(l)
    ks = glob.glob(os.path.join(d, "**", "r_kernel_stats.csv"), recursive=True)
    t = glob.glob(os.path.join(d, "**", "r_kernel_trace.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(res_dir, "kernel_stats.csv"))
    if line is not None:
        with open(os.path.join(res_dir, "bench_under_rocprofv3.json"), "w") as f:
            json.dump(line, f)
    if not t:
        print("stats pass produced no kernel trace (rc %d); see %s" % (rc, log), file=sys.stderr)
        sys.exit(1)
    dur = durations(t[0])
    shutil.rmtree(d, ignore_errors=True)
    total = float(sum(sum(v) for v in dur.values()))
    order = sorted(dur, key=lambda n: -sum(dur[n]))
    kernels = collections.OrderedDict()
    mine_first = [n for n in order if "cgpt" in n]                   # this library's kernels (some names come out mangled), by device time
    for n in order[:24] + [m for m in mine_first[:12] if m not in order[:24]]:   # (under a torch decode ours may not be among the first 24)
        v = dur[n]
        kernels[n] = {"kernel": short(n), "launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3,
                      "total_ms": sum(v) / 1e6, "share_of_device_time": sum(v) / total}
        # one kernel name can stand for several shapes (gemm9_f16_kernel<0> is qkv, proj, fc2 and the Q-Former's large linears): its
        # launches sorted by duration and cut where the next one is more than 1.25 x longer -- [launches, avg us, total ms] per group
        sv, groups, start = sorted(v), [], 0
        for j in range(1, len(sv) + 1):
            if j == len(sv) or sv[j] > 1.25 * sv[j - 1]:
                g = sv[start:j]
                groups.append({"launches": len(g), "avg_us": sum(g) / len(g) / 1e3, "total_ms": sum(g) / 1e6})
                start = j
        if len(groups) > 1:
            kernels[n]["duration_groups"] = groups[:12]
    print("stats pass done: %d kernels, %.1f ms of device time" % (len(dur), total / 1e6), flush=True)
    chosen = mine_first[:top]                                        # PMC rows

    if pmc:
        for gi, grp in enumerate(GROUPS):
            d, log, rc = run_pass(out, "p%d" % gi, ["--pmc", *grp], bench_args, env)
            f = glob.glob(os.path.join(d, "**", "r_counter_collection.csv"), recursive=True)
            tt = glob.glob(os.path.join(d, "**", "r_kernel_trace.csv"), recursive=True)
            if not f or not tt:
                print("pmc pass %d produced no counters (rc %d); see %s" % (gi, rc, log), file=sys.stderr)
                shutil.rmtree(d, ignore_errors=True)
                continue
            pd = durations(tt[0])
            vals = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f[0])):
                if r["Kernel_Name"] in chosen:
                    vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for n in chosen:
                for c in grp:
                    if vals[n][c]:
                        kernels[n][c] = sum(vals[n][c]) / len(vals[n][c])
                        kernels[n]["avg_us_in_pass_" + c] = sum(pd[n]) / len(pd[n]) / 1e3
            shutil.rmtree(d, ignore_errors=True)
            print("pmc pass", gi, grp, "done", flush=True)
    for n in chosen:
        r = kernels[n]
        if "FETCH_SIZE" in r:
            r["read_bytes_per_launch"] = r["FETCH_SIZE"] * 1024 * 2          # KiB, x2: gfx950 correction (MI355X_MICROARCH.md)
        if "WRITE_SIZE" in r:
            r["write_bytes_per_launch"] = r["WRITE_SIZE"] * 1024
        if "read_bytes_per_launch" in r and "write_bytes_per_launch" in r:
            r["bytes_per_s_stats_pass"] = (r["read_bytes_per_launch"] + r["write_bytes_per_launch"]) / (r["avg_us"] * 1e-6)
        if "GRBM_GUI_ACTIVE" in r:
            cyc = r["GRBM_GUI_ACTIVE"] / 8                                    # summed over the 8 XCDs
            r["clock_ghz_in_pass"] = cyc / r["avg_us_in_pass_GRBM_GUI_ACTIVE"] / 1e3
            r["mfma_busy_frac"] = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024)   # 256 CUs x 4 SIMDs
        if "TCC_HIT_sum" in r:
            r["l2_hit_rate"] = r["TCC_HIT_sum"] / max(r["TCC_HIT_sum"] + r.get("TCC_MISS_sum", 0.0), 1.0)
    for n, r in kernels.items():
        for sub, fl in flops.items():
            if sub in n:
                r["flop_per_launch_given"] = fl
                r["tflops"] = fl / (r["avg_us"] * 1e-6) / 1e12
                r["frac_of_2500"] = fl / (r["avg_us"] * 1e-6) / PEAK
    lib = os.path.join(ROOT, "certifiedgpt_amd", "libcgpt.so")
    summary = {"_meta": {"name": name, "git_head": os.environ.get("CGPT_GIT_HEAD"),
                         "command": "CGPT_BENCH_ONLY_TIMED=1 rocprofv3 --kernel-trace {--stats | --pmc <group>} -- python3 bench.py " + " ".join(bench_args),
                         "counter_groups": GROUPS if pmc else None, "device_time_ms_stats_pass": total / 1e6,
                         "libcgpt_sha256_16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None,
                         "note": "avg / min / max us and shares are from the --stats pass (no counters); counters are per-launch averages of "
                                 "their own --pmc pass, which runs at other clocks (avg_us_in_pass_<counter> beside each); FETCH_SIZE / "
                                 "WRITE_SIZE in KiB, FETCH_SIZE x2 (gfx950); Infinity-Cache hits are counted as traffic"},
               "bench_value": None if line is None else {k: line.get(k) for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup")},
               "bench_roofline": None if line is None else {k: line["roofline"].get(k) for k in ("kernel", "achieved", "frac", "launches",
                                                                                                  "avg_launch_ms", "flop_per_launch",
                                                                                                  "in_kernel_clock_ghz")},
               "kernels": list(kernels.values())}
    with open(os.path.join(res_dir, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    for r in list(kernels.values())[:10]:
        print("%5.1f %%  %7d x %9.1f us  %s%s" % (100 * r["share_of_device_time"], r["launches"], r["avg_us"], r["kernel"][:70],
                                                   "  mfma %.3f" % r["mfma_busy_frac"] if "mfma_busy_frac" in r else ""))


if __name__ == "__main__":
    main()
