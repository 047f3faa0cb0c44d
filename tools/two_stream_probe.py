"""Probe (round 5): can the non-GEMM kernels of a forward -- LayerNorm (HBM-bound), attention (issue-bound): 19 % of the time, both far from
the power limit the GEMMs sit at -- run in the SHADOW of the GEMMs?  Two independent classifier batches are kept in flight on two HIP
streams (two handles, two host threads); the persistent 256 x 256 GEMM is told to leave some CUs free (cgpt_set_option "gemm_grid"), so the
other stream's LayerNorm / attention workgroups find a place while a GEMM runs (a GEMM workgroup fills its CU's register file: nothing
co-resides with it).  Same images, same sample indices, same results in every configuration (checked); certified images / s.

    python tools/two_stream_probe.py [images_per_stream=51] [rounds=2]
NOTE: this probe drives two handles on the SAME device from two host threads and flips the process-global "gemm_grid"
option between runs.  include/cgpt.h supports one thread per handle on DIFFERENT devices only; the probe is safe because (a) a
complete certify_many has run from the main thread (`one_stream()`, the first warm-up call) before any worker thread starts, so every
per-DEVICE first-launch cache of the launchers (LDS attribute, CU count) is already set and only read afterwards, (b) a handle owns its
workspace and weights (nothing is
shared between handles but read-only globals), and (c) the option is set only between runs, with both streams synchronised.  Its
conclusion ("overlap across batches: closed", profiles/r05/two_stream_probe.txt) rests on that warm-up; it is not a supported deployment.
"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import certifiedgpt_amd as cg
from certifiedgpt_amd import _lib
import bench


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 51
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    xs = bench.synthetic_images(2 * G, dev)
    clfs = []
    for _ in range(2):
        c = cg.HipClassifier(mode="vit_head", num_classes=1000, max_batch=255, device=0)
        c.init_synthetic(seed=0)
        clfs.append(c)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    L = cg.lib()

    def one_stream():
        s = cg.Smooth(clfs[0], 1000, 0.5, seed=42)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = s.certify_many(xs[:G], 100, 100, 0.001, 255) + s.certify_many(xs[G:], 100, 100, 0.001, 255)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    def two_streams():
        res, bar = [None, None], threading.Barrier(3)

        def work(i):
            torch.cuda.set_device(0)
            s = cg.Smooth(clfs[i], 1000, 0.5, seed=42)
            s.reset(i * G * 200)                                   # the cursor positions of the one-stream run
            with torch.cuda.stream(streams[i]):
                bar.wait()
                res[i] = s.certify_many(xs[i * G:(i + 1) * G], 100, 100, 0.001, 255)
                streams[i].synchronize()
            bar.wait()
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        torch.cuda.synchronize()
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t in th:
            t.join()
        return dt, res[0] + res[1]

    one_stream(); two_streams()                                    # warm-up of both paths
    ref = None
    for r in range(rounds):
        for name, grid, fn in [("one stream, GEMM on 256 CUs", 0, one_stream), ("two streams, GEMM on 256 CUs", 0, two_streams),
                               ("two streams, GEMM on 240 CUs", 240, two_streams), ("two streams, GEMM on 224 CUs", 224, two_streams),
                               ("two streams, GEMM on 208 CUs", 208, two_streams), ("two streams, GEMM on 192 CUs", 192, two_streams),
                               ("two streams, GEMM on 160 CUs", 160, two_streams), ("one stream, GEMM on 224 CUs", 224, one_stream)]:
            _lib.check(L.cgpt_set_option(b"gemm_grid", grid))
            dt, out = fn()
            ref = ref or out
            same = out == ref
            print(f"round {r}  {name:32s} {2 * G / dt:7.3f} certified images/s   {1e3 * dt / (2 * G):7.2f} ms / image   results identical: {same}", flush=True)
    _lib.check(L.cgpt_set_option(b"gemm_grid", 0))
    for c in clfs:
        c.close()


if __name__ == "__main__":
    main()
