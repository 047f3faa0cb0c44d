#!/usr/bin/env python3
"""bench.py -- certified images / s on MI355X (BASELINE.json metric).

One "step" = one `Smooth.certify(x, n0=100, n=100, alpha=0.001, batch_size)` of one synthetic 224x224 image at
sigma = 0.5 through EVA-ViT-G (random-init weights of that architecture) + ln_vision(CLS) + Linear(1408->1000) head
(BASELINE.json configs[1]); i.e. n0 + n = 200 classifier forwards (reference smoothing.py:44,48).
With N GPUs the Monte-Carlo samples of every `_sample_noise` are sharded over the ranks and the int64 vote histograms are
summed with one RCCL all-reduce (strong scaling: the work per certified image is fixed).  The n0 selection draws and the n
estimation draws are independent, so `certify` runs them in the same classifier batches; the K images of a run go through
`Smooth.certify_many`, which cuts the (image, sample) rows of a group of images -- on N GPUs each rank's (n0+n)/N-draw slice of
every image -- into 255-sample classifier batches that are not aligned to image boundaries and sums all histograms of the group
with one all-reduce (same sample indices, counts, labels and radii as K separate `certify` calls; tested).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline] [--cpu-budget-s S] [--cpu-baseline-headline]
    (extra, NON-headline data points: --workload encode_img | rgf | minigpt4, --img-size 448, --n 1000 / --n0 K)
    environment: CGPT_BENCH_FORCE_NCCL=1 with --gpus 1 = the whole multi-rank code path over RCCL with a world of ONE rank (first contact
    with the collective on a one-GPU box); CGPT_BENCH_ONLY_TIMED=1 = profiling runs (tools/pmc_summary.py): nothing but the timed region's
    batches reaches the GPU, one stream synchronisation per classifier batch.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment is a self-contained N-rank run: this process -- before it touches
the GPU -- starts `python -m torch.distributed.run --nproc-per-node N` on this file as a CHILD process (one rank per GPU, as the
reference's launcher starts one process per device, launch.py:110-120), relays its output and exits with its code.  A rank whose
WORLD_SIZE differs from --gpus, or a node with fewer than N devices, is an error (exit code 2), never a silent 1-GPU measurement.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the MLP fc1 GEMM with the GELU epilogue,
gemm9_f16_kernel<EPI_F16_GELU, true>, the two-phase quadrant kernel): achieved = 2*M*6144*1408 FLOP per launch / its mean launch duration measured with HIP
events on the launch stream inside the timed region; `roofline.vit_gemms` lists the four ViT GEMM shapes the same way.
`cpu_baseline` times ONE WHOLE Smooth.certify of BASELINE configs[0] (n0 = n = 10, sigma = 0.25) on the CPU oracle (oracle/, a port
of the reference's Smooth + ViT-G forward in fp32 PyTorch) on the host cores of rank 0, and `parity` compares its (label, radius) and
per-sample votes with the same call on the GPU.  At N = 1 the headline config ITSELF follows on the oracle (200 fp32 ViT-G forwards,
~105 s on 16 threads) whenever its projected time -- 10 x the configs[0] leg just measured -- fits --cpu-budget-s (default 300; 0 =
never, --cpu-baseline-headline = always): then `cpu_baseline.value` is that like-for-like figure (`headline_certify_s`), else the
configs[0] leg scaled to 200 forwards.  At N > 1 rank 0 runs the configs[0] leg after the timed region while the other ranks wait at
the final barrier, so every line carries `cpu_baseline` and `parity`.  `single_image_certify_ms` is the reference-shaped call.
A multi-rank line explains itself: `ranks` holds, per rank, the ms of the timed region spent in its classifier passes, in the vote
all-reduce (device and host) and in the host statistics, the classifier batches it ran (`batch_samples`, from the library's own log)
and the batches the plan predicts (`planned_batches`); `ranks.scaling_inputs` puts the slowest rank's shares per certified image
next to each other.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N0, N, ALPHA, SIGMA, NUM_CLASSES = 100, 100, 0.001, 0.5, 1000
F_VIT = 520_719_260_160          # FLOP per 224^2 sample, ViT-G GEMMs + attention (SURVEY.md 8(d) / BASELINE.md section 2)
MFMA_PEAK_TFLOPS = 2500.0        # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md chip table


def synthetic_images(count, device, img=224):
    """x = (u - mean)/std, u ~ U[0,1): CLIP-normalised space, where the reference adds its noise
    (processors/base_processor.py:18-20)."""
    g = torch.Generator(device="cpu").manual_seed(1234)
    u = torch.rand(count, 3, img, img, generator=g)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    return ((u - mean) / std).to(device)


def host_cores():
    """Threads of the CPU leg.  The GPU box gives one GPU's share of the host (16 cores per GPU on this pool) while
    os.cpu_count() / the affinity mask report the whole machine; oversubscribing the share makes the fp32 forward many times
    slower, so the default is capped at 16 (CGPT_CPU_THREADS overrides; the count used is reported as `cores`)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, int(os.environ.get("CGPT_CPU_THREADS", "16"))))


def cpu_headline_leg(clf, x, cores):
    """Like for like with the headline: ONE whole `Smooth.certify(x, n0=100, n=100, alpha=0.001, batch_size=10)` at sigma = 0.5 on the
    CPU oracle (smoothing.py:29-56 restated, fp32 ViT-G + head, this run's weights, image and the GPU's own noise draws), timed; the
    same call on the GPU for the comparison of the two results.  200 fp32 ViT-G forwards: ~100 s on 16 cores (--cpu-baseline-headline)."""
    import numpy as np
    import certifiedgpt_amd as cg
    from oracle import model_oracle as mo, smooth_oracle as so
    seed = 4242
    cfg = mo.Config(mode=mo.MODE_VIT_HEAD, num_classes=NUM_CLASSES)
    params = {name: torch.from_numpy(clf.get_weight(name)).reshape(shape) for name, shape in mo.param_shapes(cfg).items()}
    torch.set_num_threads(cores)
    gpu_label, gpu_radius = cg.Smooth(clf, NUM_CLASSES, SIGMA, seed=seed).certify(x, N0, N, ALPHA, 100)
    draws = cg.noise_batch(torch.zeros_like(x), 0, N0 + N, 1.0, seed).cpu().numpy()
    done = {"n": 0}

    def classifier(batch):
        out = mo.forward_all(params, torch.from_numpy(np.ascontiguousarray(batch)), cfg)["logits"].numpy()
        done["n"] += len(batch)
        print(f"bench.py: cpu headline leg {done['n']}/{N0 + N} forwards", file=sys.stderr, flush=True)   # a progress line every ~5 s
        return out

    oracle = so.SmoothOracle(classifier, NUM_CLASSES, SIGMA, lambda first, num, shape: draws[first:first + num])
    t0 = time.perf_counter()
    cpu_label, cpu_radius = oracle.certify(x.cpu().numpy(), N0, N, ALPHA, 10)
    cpu_s = time.perf_counter() - t0
    return {"headline_certify_s": cpu_s, "headline_images_per_s": 1.0 / cpu_s,
            "headline_sample": f"one whole Smooth.certify of the headline config itself (n0={N0}, n={N}, sigma={SIGMA}, alpha={ALPHA}: "
                               f"{N0 + N} fp32 ViT-G forwards in batches of 10 on {cores} threads), same weights / image / noise draws as "
                               f"the GPU call it is compared with",
            "headline_result_cpu": [int(cpu_label), float(cpu_radius)], "headline_result_gpu": [int(gpu_label), float(gpu_radius)],
            "headline_label_equal": int(cpu_label) == int(gpu_label), "headline_abs_dR": abs(float(cpu_radius) - float(gpu_radius))}


def rank_share(n_sel, n_est, r, world):
    """Draws of one image on rank r: its slice of the n0 selection draws + its (mirrored) slice of the n estimation draws."""
    import certifiedgpt_amd as cg
    a, b = cg.shard_range(n_sel, r, world), cg.shard_range(n_est, r, world, mirrored=True)
    return (a[1] - a[0]) + (b[1] - b[0])


def planned_batches(world, lo, hi, n_sel, n_est, per_gpu, group):
    """Samples per classifier batch, per rank, for the images [lo, hi) of a run: what `run(lo, hi)` below makes the library do.  Pure
    host arithmetic (certifiedgpt_amd.batch_plan), so the GEMM shapes of an N-GPU run can be read -- and tested -- without N GPUs:
    --gpus 8 --steps 20 (n0 = n = 100, 255-sample engine, up to 51 images per call) gives every rank 20 x 25 rows = [255, 245]."""
    import certifiedgpt_amd as cg
    plan = []
    for r in range(world):
        share, sizes, i = rank_share(n_sel, n_est, r, world), [], lo
        while i < hi:
            g = min(group, hi - i)
            sizes += cg.batch_plan(g, share, per_gpu)       # one image: the fused pair pass cuts its rows the same way
            i += g
        plan.append(sizes)
    return plan


def cpu_baseline_and_parity(clf, x, process_group=None):
    """BASELINE configs[0] on both sides, in this run: ONE whole `Smooth.certify(x, n0=10, n=10, alpha=0.001, batch_size=10)`
    at sigma = 0.25 (a) timed on the host cores with the CPU oracle (oracle/smooth_oracle.py around the fp32 PyTorch-CPU
    ViT-G of oracle/model_oracle.py: the restatement of the reference's smoothing.py:29-56 + eva_vit.py, same weights --
    downloaded from the device -- same image, and the GPU's own noise draws, exported), and (b) run on the GPU through the
    product path.  Returns (cpu_baseline, parity): the timed CPU figure scaled to the headline unit, and the comparison of the
    two results (label, abstain, radius, per-sample argmax agreement).  process_group: `certifiedgpt_amd.LOCAL_ONLY` in a multi-rank run,
    so that the GPU side of this leg draws all its samples here and ends in no collective (the other ranks are waiting)."""
    import numpy as np
    import certifiedgpt_amd as cg
    from oracle import model_oracle as mo, smooth_oracle as so
    n0 = n = 10
    sigma, alpha, seed = 0.25, 0.001, 42
    cfg = mo.Config(mode=mo.MODE_VIT_HEAD, num_classes=NUM_CLASSES)
    params = {name: torch.from_numpy(clf.get_weight(name)).reshape(shape) for name, shape in mo.param_shapes(cfg).items()}
    cores = host_cores()
    torch.set_num_threads(cores)
    # GPU side first (also exports the N(0,1) draws the CPU side must see)
    smooth = cg.Smooth(clf, NUM_CLASSES, sigma, seed=seed, process_group=process_group, force_collective=False)
    gpu_label, gpu_radius = smooth.certify(x, n0, n, alpha, 10)
    gpu_logits = clf.forward_logits(x, 0, n0 + n, sigma, seed).cpu()
    draws = cg.noise_batch(torch.zeros_like(x), 0, n0 + n, 1.0, seed).cpu().numpy()

    cpu_logits = []

    def classifier(batch):
        out = mo.forward_all(params, torch.from_numpy(np.ascontiguousarray(batch)), cfg)["logits"]
        cpu_logits.append(out)
        return out.numpy()

    oracle = so.SmoothOracle(classifier, NUM_CLASSES, sigma, lambda first, num, shape: draws[first:first + num])
    t0 = time.perf_counter()
    cpu_label, cpu_radius = oracle.certify(x.cpu().numpy(), n0, n, alpha, 10)
    cpu_s = time.perf_counter() - t0
    cpu_logits = torch.cat(cpu_logits)
    agree = int((cpu_logits.argmax(1) == gpu_logits.argmax(1)).sum())
    top2 = cpu_logits.topk(2, dim=1).values
    margin = ((top2[:, 0] - top2[:, 1]) / cpu_logits.abs().max()).tolist()
    fwd_per_s = (n0 + n) / cpu_s
    baseline = {"value": fwd_per_s / (N0 + N), "unit": "certified images/s", "cores": cores, "threads": torch.get_num_threads(),
                "kind": "port",
                "sample": f"one whole Smooth.certify of BASELINE configs[0] (n0={n0}, n={n}, sigma={sigma}, alpha={alpha}: {n0 + n} ViT-G "
                          f"forwards in batches of 10, fp32 PyTorch-CPU oracle) timed at {cpu_s:.1f} s; value = its forwards/s scaled "
                          f"to the headline's n0+n={N0 + N} forwards per certified image",
                "config0_certify_s": cpu_s, "config0_images_per_s": 1.0 / cpu_s, "forwards_per_s": fwd_per_s}
    parity = {"config": f"BASELINE configs[0]: Smooth.certify n0={n0} n={n} sigma={sigma} alpha={alpha}, ViT-G + head, same weights / image / noise draws",
              "gpu": [int(gpu_label), float(gpu_radius)], "cpu_oracle": [int(cpu_label), float(cpu_radius)],
              "label_equal": int(gpu_label) == int(cpu_label), "abs_dR": abs(float(gpu_radius) - float(cpu_radius)),
              "argmax_agreement": f"{agree}/{n0 + n}", "min_fp32_top2_margin_rel": min(margin),
              "logits_rel_err": float((gpu_logits - cpu_logits).abs().max() / cpu_logits.abs().max())}
    return baseline, parity


def launch_ranks(n, argv):
    """Parent of a self-contained N-rank run: start the ranks as a child process and relay its exit code.  Nothing in this
    process has touched the GPU (torch.cuda.device_count() does not initialise it)."""
    import socket
    import subprocess
    rehearsal = bool(os.environ.get("CGPT_BENCH_ONE_GPU_REHEARSAL"))
    have = torch.cuda.device_count()
    if have < n and not rehearsal:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible on this node; refusing to measure a smaller configuration",
              file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # the CPU leg on the headline config itself (n0 = n = 100, sigma = 0.5: ~100 s of host time) beside the default configs[0] leg
    ap.add_argument("--cpu-baseline-headline", action="store_true")
    # ... which also runs by default at --gpus 1 when its projected time (10 x the configs[0] leg) fits this many seconds; 0 = never
    ap.add_argument("--cpu-budget-s", type=float, default=300.0)
    # extra data points (NOT the headline line): BASELINE configs[2] without the Vicuna decode, and the reference's own 448^2 size
    ap.add_argument("--workload", choices=["vit_head", "encode_img", "rgf", "minigpt4"], default="vit_head")
    ap.add_argument("--img-size", type=int, default=224)
    ap.add_argument("--n", type=int, default=N)      # BASELINE configs[3]: --n 1000 (sharded 125 / GPU at 8 GPUs)
    ap.add_argument("--n0", type=int, default=N0)
    ap.add_argument("--decode", choices=["graph", "hf"], default="graph")   # --workload minigpt4: greedy decode as one hipGraph | HF generate
    ap.add_argument("--prefill-linear", choices=["cgpt", "torch"], default="cgpt")   # graph decode: the prefill's linears through this library's GEMM
    args = ap.parse_args()
    headline = args.workload == "vit_head" and args.img_size == 224 and args.n == N and args.n0 == N0
    n_est, n_sel = args.n, args.n0
    rgf = args.workload == "rgf"                     # BASELINE configs[4]: 8-step RGF x smoothed predict(N) on ViT-G + head
    gen = args.workload == "minigpt4"                # BASELINE configs[2]: full MiniGPT-4, Vicuna-7B-shaped decode on PyTorch-ROCm
    mode = "vit_head" if rgf else ("encode_img" if gen else args.workload)

    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:     # self-contained N-rank run: the ranks are a child process
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}: refusing to measure a different configuration", file=sys.stderr)
        sys.exit(2)
    rehearsal = bool(os.environ.get("CGPT_BENCH_ONE_GPU_REHEARSAL"))   # the N > 1 code path on a one-GPU box: all ranks on cuda:0, gloo
    if rehearsal:
        local = 0
    elif torch.cuda.device_count() <= local:
        print(f"bench.py: rank {rank} wants cuda:{local} but {torch.cuda.device_count()} device(s) are visible", file=sys.stderr)
        sys.exit(2)
    import torch.distributed as dist
    backend = None
    # CGPT_BENCH_FORCE_NCCL=1 on ONE GPU: the whole multi-rank code path -- process group "nccl" (= RCCL), the all-reduce of the CUDA
    # int64 vote histograms, dist.barrier(device_ids=...), the `ranks` report -- with a world of one rank, so that a one-GPU box
    # executes on hardware what an 8-GPU run executes (a SUM over one rank changes nothing; the line says summed_ranks = 1).
    forced = world == 1 and os.environ.get("CGPT_BENCH_FORCE_NCCL", "") not in ("", "0")
    collective = world > 1 or forced
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        torch.cuda.set_device(local)
        backend = "gloo" if rehearsal else "nccl"            # "nccl" IS RCCL on ROCm
        dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus and dist.get_backend() == backend
    else:
        torch.cuda.set_device(local)
    # rank 0's CPU / parity leg draws all of its samples on its own GPU and ends in no collective while the other ranks wait at the
    # final barrier: a Smooth that is told to stay local (no new process group: nothing here that an 8-GPU run executes for the first time)

    import certifiedgpt_amd as cg
    if os.environ.get("CGPT_GEMM_KERNEL"):          # A/B measurements only; default = the library's own choice
        from certifiedgpt_amd import _lib
        _lib.check(cg.lib().cgpt_set_option(b"gemm_kernel", int(os.environ["CGPT_GEMM_KERNEL"])))
    if os.environ.get("CGPT_GEMM_GROUP_M"):         # measurement only: tile rows per group of the block -> tile map
        from certifiedgpt_amd import _lib
        _lib.check(cg.lib().cgpt_set_option(b"gemm_group_m", int(os.environ["CGPT_GEMM_GROUP_M"])))
    if os.environ.get("CGPT_GEMM_ABLATE"):          # measurement-only switches of experimental code paths
        from certifiedgpt_amd import _lib
        _lib.check(cg.lib().cgpt_set_option(b"gemm_ablate", int(os.environ["CGPT_GEMM_ABLATE"])))
    if os.environ.get("CGPT_BENCH_ONLY_TIMED", "") not in ("", "0"):   # profiling runs: bounded number of dispatches in flight
        from certifiedgpt_amd import _lib
        # (CGPT_BENCH_NO_SYNC=1 / CGPT_BENCH_TRACE_BATCHES=1: the one diagnostic run of profiles/r06/pmc_sigsegv.txt)
        _lib.check(cg.lib().cgpt_set_option(b"sync_batches", 0 if os.environ.get("CGPT_BENCH_NO_SYNC") else 1))
        _lib.check(cg.lib().cgpt_set_option(b"trace_batches", 1 if os.environ.get("CGPT_BENCH_TRACE_BATCHES") else 0))
    dev = torch.device("cuda", local)
    # certify runs its n0 + n draws as ONE fused pass; the largest per-rank share of it is one batch
    share = max(rank_share(n_sel, n_est, r, world) for r in range(world))   # draws of one image per rank: 25 at 8 GPUs (13 + 12)
    # Classifier batch capacity.  Smooth.certify_many cuts the (image, sample) rows of a group of images into batches of this
    # size, not aligned to image boundaries, so it is chosen for the GEMMs: 255 samples x 257 tokens = 65 535 rows = 256 tile
    # rows of 256 -> every GEMM's tile count is a multiple of the 256 CUs (200 samples: 201 tile rows, 96 % tile efficiency).
    per_gpu = 200 if (rgf or gen or args.img_size != 224) else 255
    group = 1 if (rgf or gen) else 51                                 # images per certify_many call (51 x 200 = 40 x 255)
    if os.environ.get("CGPT_BENCH_BATCH"):                            # measurement only: "capacity,group"
        per_gpu, group = (int(v) for v in os.environ["CGPT_BENCH_BATCH"].split(","))
    clf = cg.HipClassifier(mode=mode, num_classes=NUM_CLASSES, max_batch=per_gpu, device=local, img_size=args.img_size)
    clf.init_synthetic(seed=0)                                         # identical weights on every rank
    images = synthetic_images(args.steps + args.warmup, dev, args.img_size)
    base = clf
    if gen:
        # BASELINE configs[2]: encode_img in HIP + a frozen decoder of the Vicuna-7B ARCHITECTURE (LlamaConfig 4096 / 32 layers / 32
        # heads / 11008 / vocab 32000, fp16) with random-init weights -- the checkpoint is not in the container and nothing is ever
        # downloaded -- greedy decode of 20 new tokens per noisy copy (the reference's max_new_tokens), HF `generate` on PyTorch-ROCm.
        # Tokenizer: the word-hash stand-in; classes: the answers of a first 100-sample batch (frozen), everything else "other".
        from transformers import LlamaConfig, LlamaForCausalLM
        from certifiedgpt_amd.minigpt4 import MiniGPT4Classifier, WordHashTokenizer, prepare_texts
        from certifiedgpt_amd.agents.label_adapter import AnswerLabelMap
        lcfg = LlamaConfig(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                           num_key_value_heads=32, max_position_embeddings=2048, pad_token_id=0, bos_token_id=1, eos_token_id=2)
        torch.manual_seed(0)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float16)
        try:
            with torch.device(dev):
                llm = LlamaForCausalLM(lcfg).eval()
        finally:
            torch.set_default_dtype(prev)
        for prm in llm.parameters():
            prm.requires_grad = False
        tok = WordHashTokenizer(32000)
        prompt = prepare_texts(["<Img><ImageHere></Img> [vqa] what is shown in the picture"])[0]
        probe = MiniGPT4Classifier(clf, llm, tok, prompt, AnswerLabelMap(NUM_CLASSES), max_new_tokens=20, max_batch=per_gpu)
        emb = clf.encode_img_noisy(images[0], 0, per_gpu, SIGMA, 42)
        vocab = sorted(set(probe.generate_from_embeds(emb, prompt)))[:NUM_CLASSES - 1]
        base = MiniGPT4Classifier(clf, llm, tok, prompt, AnswerLabelMap(NUM_CLASSES, vocab, frozen=True), max_new_tokens=20, max_batch=per_gpu,
                                  decode=args.decode, prefill_linear=args.prefill_linear)
    smooth = cg.Smooth(base, NUM_CLASSES, SIGMA, seed=42, non_certifiable=(base.label_map.other_id,) if gen else (),
                       force_collective=forced)
    torch.cuda.synchronize()

    def barrier():
        if collective:
            if backend == "nccl":
                dist.barrier(device_ids=[local])      # this rank's own GPU, explicitly
            else:
                dist.barrier()

    attack = cg.RGFAttack(smooth, steps=8, num_dirs=1, delta=0.5, lr=0.05, eps=0.25, dir_seed=1234) if rgf else None

    def step(img):
        if rgf:                                      # attack towards class 1, then the smoothed prediction of the result
            _, label, hist = attack.attack(img, 1, n_est, ALPHA, per_gpu, targeted=True)
            return label, hist[-1]
        return smooth.certify(img, n_sel, n_est, ALPHA, per_gpu)

    def run(lo, hi):
        out, i = [], lo
        while i < hi:
            g = min(group, hi - i)
            out += smooth.certify_many(images[i:i + g], n_sel, n_est, ALPHA, per_gpu) if g > 1 else [step(images[i])]
            i += g
        return out

    results = run(0, args.warmup)
    torch.cuda.synchronize()
    clf.profile_read(0)
    clf.profile(not os.environ.get("CGPT_BENCH_NO_PROFILE"))        # A/B only: what the HIP-event hooks around every GEMM cost
    if collective:
        smooth.collect_timing(True)                  # HIP events around each rank's classifier pass and around the all-reduce
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    results += run(args.warmup, args.warmup + args.steps)
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0                 # this rank's own time, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    clf.profile(False)
    ranks_report = None
    if collective:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what the line needs to explain itself: per rank, the time in its own classifier passes, in the collective, and in total;
        # and how many ranks the communicator really summed over (a SUM of ones through the same backend as the vote counts)
        tm = smooth.timing()
        smooth.collect_timing(False)
        # the classifier batches this rank ran in the timed region (library log since profile(True)), as [size, count] runs
        ran = clf.profile_batches()
        runs = []
        for v in ran:
            if runs and runs[-1][0] == v:
                runs[-1][1] += 1
            else:
                runs.append([v, 1])
        runs = runs[:16] + [[-1, 0]] * (16 - min(16, len(runs)))          # fixed width for the gather; -1 pads
        mine = torch.tensor([1e3 * t_rank, tm["compute_ms"], tm["allreduce_ms"], tm["allreduce_host_ms"], float(tm["calls"]),
                             tm["stats_host_ms"], float(len(ran))] + [float(v) for pair in runs for v in pair],
                            device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ones = torch.ones(1, device=dev, dtype=torch.int64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rows = [[float(v) for v in r.tolist()] for r in allr]
        plan = planned_batches(world, args.warmup, args.warmup + args.steps, n_sel, n_est, per_gpu, group) if not (rgf or gen) else None

        def runs_of(r):
            return [[int(r[7 + 2 * k]), int(r[8 + 2 * k])] for k in range(16) if r[7 + 2 * k] >= 0]

        def as_runs(sizes):
            out = []
            for v in sizes:
                if out and out[-1][0] == v:
                    out[-1][1] += 1
                else:
                    out.append([v, 1])
            return out
        slow = max(range(world), key=lambda i: rows[i][1])                # the rank with the longest classifier passes
        ranks_report = {"summed_ranks": int(ones.item()),
                        "per_rank_ms": [{"rank": i, "total": r[0], "classifier_passes": r[1], "all_reduce_device": r[2],
                                         "all_reduce_host": r[3], "host_statistics": r[5], "sample_noise_calls": int(r[4]),
                                         "classifier_batches": int(r[6]), "batch_samples": runs_of(r),
                                         "planned_batches": as_runs(plan[i])[:16] if plan is not None else None}
                                        for i, r in enumerate(rows)],
                        "batches_as_planned": (all(runs_of(r) == as_runs(plan[i])[:16] for i, r in enumerate(rows))
                                               if plan is not None else None),
                        # what a scaling number is made of: the slowest rank's ms per certified image in its own classifier passes, in the
                        # all-reduce on the device (includes waiting for the slowest rank), in the host statistics, and the rest of the
                        # step (launch gaps, the device->host copies, Python); speed-up over one GPU = that run's ms_per_step / this total
                        "scaling_inputs": {"slowest_rank": slow,
                                           "classifier_ms_per_image": rows[slow][1] / args.steps,
                                           "all_reduce_device_ms_per_image": rows[slow][2] / args.steps,
                                           "host_statistics_ms_per_image": rows[slow][5] / args.steps,
                                           "other_ms_per_image": (1e3 * elapsed - rows[slow][1] - rows[slow][2] - rows[slow][5]) / args.steps,
                                           "ms_per_step": 1e3 * elapsed / args.steps,
                                           "note": "projected ceilings on one GPU (DESIGN.md section 6, profiles/r04/shard_bench.txt): certify_many "
                                                   "8.0x at 8 GPUs (12.7 ms per image per rank), the reference-shaped per-image call 5.9x (17.65 ms)"},
                        "rank_total_ms_max": max(r[0] for r in rows), "rank_total_ms_min": min(r[0] for r in rows),
                        "classifier_ms_max": max(r[1] for r in rows), "classifier_ms_min": min(r[1] for r in rows),
                        "all_reduce_device_ms_max": max(r[2] for r in rows),
                        "note": "per rank over the timed region: HIP events around its own classifier passes and around dist.all_reduce of the "
                                "vote histograms (device time: includes waiting for the slowest rank), host time inside dist.all_reduce; "
                                "summed_ranks = SUM of ones through the same communicator"}

    # HBM-side traffic of the fc1 GEMM: rocprofv3 PMC (FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled as the
    # MI355X guide prescribes for gfx950) cannot be collected from inside this process; the committed summary of the same
    # command under profiles/ is reported when it matches this configuration (batch), else null.
    traffic, traffic_note, pmc_fc1 = None, "no PMC summary for this configuration", None
    FC1_KERNEL = "gemm9_f16_kernel<1>"                 # roofline.kernel as rocprofv3 names it
    try:
        import glob
        import hashlib
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json")))
        if not cands:
            raise FileNotFoundError("profiles/r*/pmc_summary.json")
        newest = cands[-1]                                             # the newest round directory that holds a summary
        with open(newest) as f:
            summ = json.load(f)
        # the row whose launches are all full batches of `per_gpu` samples (tools/pmc_summary.py profiles a run made of nothing else)
        pm, meta = summ.get("fc1_full_batch") or summ["fc1"], summ.get("_meta", {})
        rel = os.path.relpath(newest, ROOT)
        lib_sha = hashlib.sha256(open(cg._lib.LIB_PATH, "rb").read()).hexdigest()[:16]
        if not (world == 1 and headline):
            traffic_note = "PMC summary %s is for the single-GPU headline configuration" % rel
        elif FC1_KERNEL not in pm.get("kernel_name", ""):
            traffic_note = "PMC summary %s (commit %s) is for kernel %r, not %s: traffic withheld" % (
                rel, meta.get("git_head"), pm.get("kernel_name"), FC1_KERNEL)
        elif pm.get("batch_samples", meta.get("batch_size_per_gpu")) != per_gpu:
            traffic_note = "PMC summary %s was taken at batch %s, this run uses %d: traffic withheld" % (
                rel, pm.get("batch_samples", meta.get("batch_size_per_gpu")), per_gpu)
        else:
            traffic = pm["hbm_read_bytes_corrected"] + pm["hbm_write_bytes"]
            same = meta.get("libcgpt_sha256_16") == lib_sha
            # MFMA busy fraction and shader clock of the same kernel in the profiled run: busy x clock / 2.4 GHz reproduces `frac`
            pmc_fc1 = {"mfma_busy_frac": pm.get("mfma_busy_frac"), "clock_ghz": pm.get("clock_ghz"), "l2_hit_rate": pm.get("l2_hit_rate"),
                       "source": rel, "same_build": same, "launches": pm.get("launches"), "avg_us_unprofiled_pass": pm.get("avg_us"),
                       "frac_from_profile": pm.get("frac_of_peak"),
                       "note": "mfma_busy_frac and clock_ghz belong to the PROFILED run (a --pmc pass holds other clocks than this run): their "
                               "product / 2.4 GHz is that pass's frac, not this run's; this run's own pair is in_kernel_clock_ghz and "
                               "implied_mfma_busy_frac"}
            traffic_note = ("bytes per launch of %s from %s, taken at commit %s (%s this run's libcgpt.so; rocprofv3 --pmc FETCH_SIZE / "
                            "WRITE_SIZE in separate passes, FETCH_SIZE x2 gfx950 correction; Infinity-Cache hits are counted): read %.0f MB + "
                            "write %.0f MB vs algorithmic %.0f MB (A %.0f + W %.0f + out %.0f); counters and algorithmic bytes are both for "
                            "full %d-sample batches" % (
                                FC1_KERNEL, rel, meta.get("git_head"),
                                "the same build as" if same else "a DIFFERENT build than",
                                pm["hbm_read_bytes_corrected"] / 1e6, pm["hbm_write_bytes"] / 1e6,
                                (per_gpu * 257 * (1408 + 6144) * 2 + 6144 * 1408 * 2) / 1e6, per_gpu * 257 * 1408 * 2 / 1e6,
                                6144 * 1408 * 2 / 1e6, per_gpu * 257 * 6144 * 2 / 1e6, per_gpu))
    except Exception as e:
        traffic_note = "no usable PMC summary under profiles/ (%r)" % (e,)

    fc1_ms, fc1_flops, fc1_n = clf.profile_read(1)
    fc1_clock = clf.profile_clock(1)                  # in-kernel: sum of the workgroups' shader cycles / 100-MHz ticks, timed region only
    by_kind = {}
    for kind, name in ((2, "qkv"), (3, "proj"), (1, "fc1_gelu"), (4, "fc2")):
        ms, fl, cnt = clf.profile_read(kind)
        if cnt:
            by_kind[name] = {"avg_launch_us": 1e3 * ms / cnt, "tflops": fl / (ms * 1e-3) / 1e12, "launches": cnt,
                             "in_kernel_clock_ghz": clf.profile_clock(kind)}
    all_clock = clf.profile_clock(0)
    all_ms, all_flops, all_n = clf.profile_read(0)

    # The reference-shaped call, outside the timed region: ONE Smooth.certify(x, n0, n, alpha, batch_size = n0 + n) per image
    # (smoothing.py:29-56), i.e. no grouping of images; every rank takes part (its shard + the all-reduce).
    single_ms = None
    only_timed = os.environ.get("CGPT_BENCH_ONLY_TIMED", "") not in ("", "0")   # profiling runs (tools/pmc_summary.py): nothing but the
    if not rgf and not only_timed:                                              # timed region's full batches reaches the GPU
        reps = 3
        smooth.certify(images[0], n_sel, n_est, ALPHA, n_sel + n_est)
        torch.cuda.synchronize()
        barrier()
        ts = time.perf_counter()
        for i in range(reps):
            smooth.certify(images[i % len(images)], n_sel, n_est, ALPHA, n_sel + n_est)
        torch.cuda.synchronize()
        barrier()
        single_ms = 1e3 * (time.perf_counter() - ts) / reps

    # SURVEY.md 8(e) asks for both partitions: the same timed images once more, IMAGE-sharded (Smooth.certify_images: every rank
    # certifies whole images with all their draws, no vote all-reduce; 16 bytes per image come back).  Same sample indices as the
    # timed region, so the (label, radius) list must be the same list.  Outside the timed region; reported beside the headline.
    image_sharded = None
    if collective and not rgf and not gen:
        sm2 = cg.Smooth(base, NUM_CLASSES, SIGMA, seed=42, force_collective=forced)
        sm2.certify_images(images[:min(world, len(images))], n_sel, n_est, ALPHA, per_gpu)      # untimed: first use of this path
        sm2.reset(args.warmup * (n_sel + n_est))                                             # the cursor of the timed region
        torch.cuda.synchronize()
        barrier()
        ts = time.perf_counter()
        out2 = sm2.certify_images(images[args.warmup:args.warmup + args.steps], n_sel, n_est, ALPHA, per_gpu)
        torch.cuda.synchronize()
        barrier()
        t2 = torch.tensor([time.perf_counter() - ts], device=dev, dtype=torch.float64)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        el2 = float(t2.item())
        image_sharded = {"value": args.steps / el2, "unit": "certified images/s", "ms_per_step": 1e3 * el2 / args.steps,
                         "images_per_rank_max": -(-args.steps // world),
                         "equals_sample_sharded": [(int(l), float(r)) for l, r in out2] ==
                                                  [(int(l), float(r)) for l, r in results[args.warmup:]],
                         "note": "the other partition of SURVEY.md 8(e): whole images per rank (all n0 + n draws of each), no vote "
                                 "all-reduce, one 16-byte-per-image exchange of results; the timed region's images and sample indices, "
                                 "barrier + synchronize on both sides, MAX over ranks; `value` above stays the sample-sharded mode"}

    # The yardstick beside the data-sheet peak: the dense fp16 MFMA rate THIS device sustains on random operands with nothing else
    # running (cgpt_mfma_sustained: MFMA-only loop, ~2 s, clock settled first), measured after everything else, rank 0 of a 1-GPU run only.
    sustained = None
    if world == 1 and rank == 0 and headline and not os.environ.get("CGPT_BENCH_NO_SUSTAINED") and not only_timed:
        import ctypes as C
        tf, ghz = C.c_double(), C.c_double()
        if cg.lib().cgpt_mfma_sustained(2.0, C.byref(tf), C.byref(ghz)) == 0:
            sustained = {"mfma_only_tflops": tf.value, "clock_ghz": ghz.value,
                         "note": "v_mfma_f32_16x16x32_f16 back to back from registers on random fp16 operands, no LDS, no memory: what this device "
                                 "sustains when it does nothing but MFMAs (the 2 500 of the data sheet is 1 024 SIMDs x 1 024 FLOP/clk at 2.4 GHz); "
                                 "measured in this run, after the timed region (profiles/r04/mfma_sustained.txt)"}

    gen_report = None
    if gen:
        # where a certified image's time goes (one 200-row batch, HIP events), and what the graph decode changes against HF generate
        def timed(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps, out
        x0 = images[0]
        enc_ms, emb = timed(lambda: clf.encode_img_noisy(x0, 0, per_gpu, SIGMA, 42))
        segs = base._segment_embeddings(prompt, dev)
        embs = torch.cat([segs[0].expand(per_gpu, -1, -1), emb.to(segs[0].dtype), segs[1].expand(per_gpu, -1, -1)], dim=1)
        hf = MiniGPT4Classifier(clf, llm, tok, prompt, base.label_map, max_new_tokens=20, max_batch=per_gpu, decode="hf")
        gr = MiniGPT4Classifier(clf, llm, tok, prompt, base.label_map, max_new_tokens=20, max_batch=per_gpu, decode="graph")
        gc = MiniGPT4Classifier(clf, llm, tok, prompt, base.label_map, max_new_tokens=20, max_batch=per_gpu, decode="graph", prefill_linear="cgpt")
        hf_ms, a_hf = timed(lambda: hf.generate_from_embeds(emb, prompt), reps=2)
        gr_ms, a_gr = timed(lambda: gr.generate_from_embeds(emb, prompt), reps=2)
        gc_ms, a_gc = timed(lambda: gc.generate_from_embeds(emb, prompt), reps=2)
        from certifiedgpt_amd.minigpt4 import _LinearRoute, greedy_decode_parity, fp16_ulp
        # parity of the measured path with the reference's call, row by row: HF generate's tokens and processed scores on this batch;
        # a row may leave HF's trajectory only at a step whose HF top-2 margin is within the logit noise of the paths (eps = 4 fp16 ulp
        # of the largest logit; tests/test_gpu_fullsize.py measures the noise itself): every DECISIVE row must give the identical answer
        with torch.no_grad():
            ho = llm.generate(inputs_embeds=embs, attention_mask=torch.ones(embs.shape[:2], dtype=torch.int, device=dev),
                              max_new_tokens=20, output_scores=True, return_dict_in_generate=True, **base.hf_generate_kwargs())
            h_tok, h_sc = ho.sequences, torch.stack(ho.scores, dim=1).float()
            eps = 4.0 * fp16_ulp(float(h_sc[torch.isfinite(h_sc)].abs().max()))
            par_gr = greedy_decode_parity(h_tok, h_sc, gr._generate_graph(embs), eps)
            par_gc = greedy_decode_parity(h_tok, h_sc, gc._generate_graph(embs), eps)
            del ho, h_sc
        with torch.no_grad():
            pre_ms, _ = timed(lambda: llm(inputs_embeds=embs, use_cache=True, logits_to_keep=1).logits, reps=2)
            with _LinearRoute.enabled():
                pre_c_ms, _ = timed(lambda: llm(inputs_embeds=embs, use_cache=True, logits_to_keep=1).logits, reps=2)
        gen_report = {"rows_per_batch": per_gpu, "prompt_tokens_incl_32_image_tokens": int(embs.shape[1]),
                      "encode_img_ms": enc_ms, "hf_generate_ms": hf_ms, "graph_decode_ms": gr_ms, "prefill_alone_ms": pre_ms,
                      "per_token_step_ms_hf": (hf_ms - pre_ms) / 19.0, "per_token_step_ms_graph": (gr_ms - pre_ms) / 19.0,
                      "graph_decode_cgpt_prefill_ms": gc_ms, "prefill_alone_cgpt_linears_ms": pre_c_ms, "routed_linears": gc.routed_linears,
                      "answers_identical_rows": sum(int(a == b) for a, b in zip(a_hf, a_gr)),
                      "answers_identical_rows_cgpt_prefill_vs_hf": sum(int(a == b) for a, b in zip(a_hf, a_gc)),
                      "answers_identical_decisive": {"eps_logit": eps, "graph": f"{par_gr['decisive_identical']}/{par_gr['decisive']}",
                                                     "graph_cgpt_prefill": f"{par_gc['decisive_identical']}/{par_gc['decisive']}",
                                                     "rows_leaving_hf_at_a_decisive_step": len(par_gr["violations"]) + len(par_gc["violations"]),
                                                     "note": "decisive = every step's HF top-2 logit margin > eps; other rows may flip on fp16 "
                                                             "rounding of near-tied logits (random-init decoder)"},
                      "decode": args.decode,
                      "prefill_linear": args.prefill_linear,
                      "decode_stats": base.decode_stats}

    if rank == 0:
        value = args.steps / elapsed
        fc1_tflops = fc1_flops / (fc1_ms * 1e-3) / 1e12 if fc1_ms > 0 else 0.0
        all_tflops = all_flops / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0
        line = {
            "metric": "certified images/sec (N=100, sigma=0.5)", "value": value, "unit": "certified images/s",
            "n_gpus": world,
            "rccl_ranks": (ranks_report["summed_ranks"] if backend == "nccl" else 0) if collective else 1,
            "collective_backend": backend or "none (single rank)",
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "EVA-ViT-G encoder + ln_vision(CLS) + Linear head, random-init weights, 224x224 synthetic "
                                   "image, Smooth.certify n0=100 n=100 alpha=0.001 sigma=0.5 (BASELINE configs[1])",
                       "n0": N0, "n": N, "alpha": ALPHA, "sigma": SIGMA, "num_classes": NUM_CLASSES,
                       "batch_size_per_gpu": per_gpu, "forwards_per_image": N0 + N,
                       "images_per_pass": group,
                       "parallelism": (f"sample-sharded x{world}: every rank draws its slice of each image's n0 and n samples; up to "
                                       f"{group} images per Smooth.certify_many call (their rows cut into {per_gpu}-sample classifier "
                                       f"batches, one int64[G,2,{NUM_CLASSES}] all-reduce per call)")},
            "forwards_per_s": value * (N0 + N),
            "vit_tflops_end_to_end": value * (N0 + N) * F_VIT / 1e12,
            "roofline": {"bound": "mfma", "kernel": "gemm9_f16_kernel<1> = <EPI_F16_GELU> (ViT MLP fc1 + GELU, M=batch*257, N=6144, K=1408)",
                         "achieved": fc1_tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fc1_tflops / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                         "in_kernel_clock_ghz": fc1_clock,
                         "implied_mfma_busy_frac": (fc1_tflops / MFMA_PEAK_TFLOPS) * 2.4 / fc1_clock if fc1_clock else None,
                         "in_kernel_clock_note": "shader clock this kernel held in THIS run: d(s_memtime) / d(s_memrealtime) x 100 MHz summed over all "
                                                 "workgroups of its launches in the timed region (the data sheet's peak assumes 2.4 GHz); "
                                                 "implied_mfma_busy_frac = frac x 2.4 / this clock = the share of THIS run's cycles in which the "
                                                 "matrix pipes would have to be issuing to deliver `achieved`",
                         "launches": fc1_n, "avg_launch_ms": fc1_ms / max(fc1_n, 1),
                         "flop_per_launch": fc1_flops / max(fc1_n, 1),
                         "all_gemms": {"achieved": all_tflops, "frac": all_tflops / MFMA_PEAK_TFLOPS, "launches": all_n,
                                       "total_ms": all_ms, "in_kernel_clock_ghz": all_clock},
                         "vit_gemms": by_kind,
                         "pmc": pmc_fc1, "device_sustained": sustained},
            "results_sample": [[int(l), float(r)] for l, r in results[-3:]],
            "single_image_certify_ms": single_ms,
            "single_image_certify_note": "one reference-shaped Smooth.certify(x, n0, n, alpha, batch_size=n0+n) per image, no grouping "
                                         "of images (the headline value sends groups of images through Smooth.certify_many)",
        }
        if sustained:
            sustained["frac_of_sustained"] = fc1_tflops / sustained["mfma_only_tflops"]
            sustained["all_gemms_frac_of_sustained"] = all_tflops / sustained["mfma_only_tflops"]
        if ranks_report is not None:
            line["ranks"] = ranks_report
        if image_sharded is not None:
            line["image_sharded"] = image_sharded
        if not headline:
            T = (args.img_size // 14) ** 2 + 1
            what = (f"8-step RGF attack (1 direction per step) + smoothed predict, {attack.forwards_per_image(n_est)} forwards per image"
                    if rgf else f"Smooth.certify n0={n_sel} n={n_est} alpha=0.001 sigma=0.5")
            if gen:
                what += (" over full MiniGPT-4: encode_img in HIP + random-init decoder of the Vicuna-7B architecture (fp16, greedy decode on "
                         "PyTorch-ROCm: %s; 20 new tokens per noisy copy, batches of %d), answers -> classes by a frozen vocabulary of %d "
                         "answers (BASELINE configs[2])" % (("prefill + 19 steps replayed from one hipGraph" + (", the prefill's linears through "
                                                             "libcgpt's GEMM" if args.prefill_linear == "cgpt" else "")) if args.decode == "graph" else
                                                            "HF generate", per_gpu, len(base.label_map.answers)))
                line["minigpt4_phases"] = gen_report
            line["config"]["workload"] = (f"NON-HEADLINE data point: mode={mode}, image {args.img_size}x{args.img_size} (T={T}), "
                                          f"random-init weights, {what}")
            line["config"].update({"n0": n_sel, "n": n_est})
            fw = attack.forwards_per_image(n_est) if rgf else n_sel + n_est
            line["config"]["forwards_per_image"] = fw
            line["forwards_per_s"] = value * fw
            if rgf:
                line["metric"], line["unit"] = "attacked images/sec (8-step RGF, N=%d, sigma=0.5)" % n_est, "attacked images/s"
                line["parity_note"] = ("parity unpinned: the reference has no attack code (README.md:62-64,108-120 is prose); the schedule is this "
                                       "build's own rule, pinned bit-exactly against oracle/rgf_oracle.py only (tests/test_gpu_fullsize.py)")
            line.pop("vit_tflops_end_to_end", None)
            line["roofline"]["traffic"] = None
            line["roofline"]["flop_per_launch"] = fc1_flops / max(fc1_n, 1)
        if not args.no_cpu_baseline and headline:
            try:
                cb, line["parity"] = cpu_baseline_and_parity(clf, images[0], process_group=cg.LOCAL_ONLY if world > 1 else None)
                line["cpu_baseline"] = cb
                # the like-for-like leg: the headline config itself on the oracle, when it fits the budget (single rank only: at N > 1
                # the other ranks are waiting at the barrier below)
                projected = cb["config0_certify_s"] * (N0 + N) / 20.0
                want = args.cpu_baseline_headline or (world == 1 and 0 < projected <= args.cpu_budget_s)
                cb["headline_leg"] = ("ran" if want else "skipped") + (": projected %.0f s (10 x the configs[0] leg) against --cpu-budget-s %.0f"
                                                                      % (projected, args.cpu_budget_s)) + \
                                     ("" if world == 1 else "; multi-rank run: configs[0] leg only")
                if want:
                    cb.update(cpu_headline_leg(clf, images[0], cb["cores"]))
                    cb["value_config0_scaled"] = cb["value"]
                    cb["value"] = cb["headline_images_per_s"]
                    cb["sample"] = cb["headline_sample"] + " (like for like with `value`); config0_* = the BASELINE configs[0] leg: " + cb["sample"]
                    line["gpu_over_cpu_headline"] = value / cb["headline_images_per_s"]
                line["gpu_over_cpu"] = value / cb["value"]
            except Exception as e:  # the CPU leg must never void the GPU measurement
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if collective:
        barrier()                                    # ranks > 0 wait here while rank 0 runs the CPU leg
    clf.close()
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
