"""Idle time between consecutive kernels of a `rocprofv3 --kernel-trace` run of bench.py: reads the *_kernel_trace.csv under the given
directory and reports, for the longest back-to-back stretch of forward kernels, the sum of kernel durations, the sum of the gaps between
one kernel's end and the next one's start, and the gap distribution.  Usage (GPU box):
    rocprofv3 --kernel-trace -d gpurun_out/gaps -o gaps --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline
    python tools/kernel_gaps.py gpurun_out/gaps"""
import csv, glob, os, sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gaps"
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
print(f"{len(rows)} kernels in {len(files)} file(s)")
# stretches: consecutive kernels whose gap is < 200 us (a host-side pause ends a stretch)
best, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][1] > 200_000:
        if len(cur) > len(best): best = cur
        cur = []
    cur.append(r)
if len(cur) > len(best): best = cur
busy = sum(e - s for s, e, _ in best)
gaps = [best[i + 1][0] - best[i][1] for i in range(len(best) - 1)]
span = best[-1][1] - best[0][0]
pos = [g for g in gaps if g > 0]
print(f"longest stretch: {len(best)} kernels over {span / 1e6:.2f} ms; kernels {busy / 1e6:.2f} ms ({100 * busy / span:.2f} %), "
      f"gaps {sum(pos) / 1e6:.2f} ms ({100 * sum(pos) / span:.2f} %), overlapping starts {sum(1 for g in gaps if g <= 0)}")
pos.sort()
if pos:
    q = lambda p: pos[min(len(pos) - 1, int(p * len(pos)))] / 1e3
    print(f"gap us: median {q(.5):.2f}  p90 {q(.9):.2f}  p99 {q(.99):.2f}  max {pos[-1] / 1e3:.2f}")
by = {}
for i, g in enumerate(gaps):
    if g > 0:
        k = best[i][2].split("(")[0][-40:] + " -> " + best[i + 1][2].split("(")[0][-40:]
        a = by.setdefault(k, [0, 0]); a[0] += g; a[1] += 1
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {t / 1e6:8.3f} ms  {n:6d} x {t / n / 1e3:6.2f} us   {k}")
