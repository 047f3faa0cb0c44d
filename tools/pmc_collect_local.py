"""Rebuilds profiles/<round>/pmc_summary.json from the per-pass CSVs that `tools/pmc_summary.py` left under gpurun_out/pmc_summary/
(the GPU box returns only gpurun_out/): same parser, no rocprofv3 run.   python tools/pmc_collect_local.py [round_dir]"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_summary as ps


def main():
    round_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ps.ROOT, "profiles", "r01")
    res = {k: {} for k in ps.KERNELS}
    for i, grp in enumerate(ps.GROUPS):
        d = os.path.join(ps.OUT, f"p{i}")
        f = glob.glob(os.path.join(d, "**", "r_counter_collection.csv"), recursive=True)
        t = glob.glob(os.path.join(d, "**", "r_kernel_trace.csv"), recursive=True)
        if not f or not t:
            print("pass", i, "missing", file=sys.stderr)
            continue
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(t[0])):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for key, pat in ps.KERNELS.items():
            names = [n for n in vals if pat in n]
            if not names:
                continue
            n = names[0]
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
            cyc = r["GRBM_GUI_ACTIVE"] / 8
            r["clock_ghz"] = cyc / r["avg_us_GRBM_GUI_ACTIVE"] / 1e3
            r["mfma_busy_frac"] = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024)
        if "TCC_HIT_sum" in r:
            r["l2_hit_rate"] = r["TCC_HIT_sum"] / max(r["TCC_HIT_sum"] + r["TCC_MISS_sum"], 1.0)
    with open(os.path.join(round_dir, "pmc_summary.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: {c: round(v, 4) for c, v in r.items() if c in ("launches", "hbm_read_bytes_corrected", "hbm_write_bytes", "clock_ghz", "mfma_busy_frac", "l2_hit_rate", "avg_us_FETCH_SIZE")} for k, r in res.items()}))


if __name__ == "__main__":
    main()
