"""Per-rank cost of one certify at the 1/2/4/8-GPU shard sizes, measured on ONE GPU (the driver runs the real N-GPU bench).
A rank of a G-GPU job runs its shard_range share of the n0 = 100 and n = 100 draws (25 samples at G = 8) as one fused pass; this times exactly that pass (the all-reduce of
8 KB and the statistics are not included) and prints the implied strong-scaling ceiling.   python tools/shard_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import certifiedgpt_amd as cg
import bench

def main():
    dev = torch.device("cuda", 0)
    x = bench.synthetic_images(1, dev)[0]
    clf = cg.HipClassifier(mode="vit_head", num_classes=1000, max_batch=255, device=0)
    clf.init_synthetic(seed=0)
    base = gbase = None
    for world in (1, 2, 4, 8):
        a, b = cg.shard_range(100, 0, world), cg.shard_range(100, 0, world, mirrored=True)
        na, nb = a[1] - a[0], b[1] - b[0]                      # rank 0's share; every rank has the same total
        for _ in range(2):
            clf.sample_counts_pair(x, 0, na, 100, nb, na + nb, 0.5, 42)
        torch.cuda.synchronize()
        clf.profile_read(0); clf.profile(True)
        t0 = time.perf_counter(); it = 4 * world
        for _ in range(it):
            clf.sample_counts_pair(x, 0, na, 100, nb, na + nb, 0.5, 42)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / it * 1e3
        clf.profile(False)
        gms, gfl, gn = clf.profile_read(0)
        base = base or ms
        print(f"world {world}: {na + nb:3d} samples/rank  {ms:8.2f} ms/certify-shard  gemms {gms / it:7.2f} ms ({gfl / gms / 1e9:5.0f} TF)"
              f"  non-gemm {ms - gms / it:6.2f} ms   speed-up ceiling {base / ms:4.2f}x", flush=True)
        # Smooth.certify_many as bench.py uses it: 51 images per call, rows cut into 255-sample batches
        G = 51
        xs = torch.stack([x * (1.0 - 0.001 * i) for i in range(G)])
        clf.sample_counts_images(xs[:8], 0, na, 100, nb, 200, 0.5, 42)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        clf.sample_counts_images(xs, 0, na, 100, nb, 200, 0.5, 42)
        torch.cuda.synchronize()
        msg = (time.perf_counter() - t0) * 1e3 / G
        gbase = gbase or msg
        print(f"         certify_many, 51 images, 255-sample batches: {msg:8.2f} ms/certify-shard   speed-up ceiling {gbase / msg:4.2f}x"
              f"   (vs one image per pass on 1 GPU: {base / msg:4.2f}x)", flush=True)
    clf.close()

if __name__ == "__main__":
    main()
