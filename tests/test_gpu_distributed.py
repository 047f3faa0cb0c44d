"""2-rank end-to-end test of the sharded GPU path on ONE GPU: each rank owns a HipClassifier on cuda:0, `Smooth.certify` /
`predict` shard the sample range, gloo all-reduces the CUDA histograms.  (The 8-GPU RCCL run is the driver's; this
covers everything but the transport.)  Ranks must agree with each other and with the single-process result."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import certifiedgpt_amd as cg
    from oracle import model_oracle as mo
    from gpu_util import tiny_pair, DEV
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        K = 10
        clf, p16, params, cfg = tiny_pair(mo.MODE_ENCODE_IMG, num_classes=K, max_batch=32)
        x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
        s = cg.Smooth(clf, K, 0.25, seed=5)
        xs = torch.stack([x0, x0 * 0.5, x0 + 0.3])
        out = (s.certify(x0, 25, 39, 0.05, 32), s.predict(x0, 30, 0.05, 7), s._sample_noise(x0, 11, 4).tolist(),
               s.certify_many(xs, 9, 11, 0.05, 32),                       # several images per fused pass, sharded
               [s.certify(xs[i], 9, 11, 0.05, 32) for i in range(3)])
        cursor = s._next_sample                                            # image-sharded: rank 0 takes two images, rank 1 one
        by_image = s.certify_images(xs, 9, 11, 0.05, 32)
        s.reset(cursor)
        out += (by_image, [s.certify(xs[i], 9, 11, 0.05, 32) for i in range(3)])
        cursor = s._next_sample
        pred_by_image = [int(v) for v in s.predict_images(xs, 30, 0.05, 7)]
        s.reset(cursor)
        out += (pred_by_image, [int(s.predict(xs[i], 30, 0.05, 7)) for i in range(3)])
        q.put((rank, out))
        clf.close()
    finally:
        if world > 1:
            dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_two_ranks_on_one_gpu_match_single_process():
    single = _launch(1)[0][1]
    two = _launch(2)
    assert two[0][1] == two[1][1] == single, (two, single)
    # certify_many consumes the same sample indices as consecutive certify calls would: re-run from the same cursor
    import certifiedgpt_amd as cg  # noqa: F401
    assert len(single[3]) == 3 and all(isinstance(r[0], int) for r in single[3])
    # Smooth.certify_images (whole images per rank, no vote all-reduce) returns the list of the one-by-one loop from the same cursor
    assert single[5] == single[6] and two[0][1][5] == two[0][1][6] == single[5]
    assert single[7] == single[8] and two[0][1][7] == two[0][1][8] == single[7]


def _nccl_run(rank, world, port, q):
    """One rank per GPU over RCCL (backend "nccl" IS RCCL on ROCm): the all-reduce of the int64 vote histograms on xGMI."""
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import certifiedgpt_amd as cg
    from oracle import model_oracle as mo
    from gpu_util import make_classifier
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        K = 10
        cfg = mo.tiny_config(mode=mo.MODE_ENCODE_IMG, num_classes=K)
        clf = cg.HipClassifier(mode="encode_img", num_classes=K, max_batch=32, device=rank, vit_ln_eps=cfg.vit_ln_eps,
                               ln_vision_eps=cfg.ln_vision_eps, qf_ln_eps=cfg.qf_ln_eps,
                               **{k: getattr(cfg, k) for k in ("img_size", "patch_size", "vit_dim", "vit_depth", "vit_heads", "vit_mlp",
                                                               "qf_layers", "qf_dim", "qf_heads", "qf_ffn", "qf_queries", "qf_xattn_freq", "proj_dim")})
        clf.load_state_dict(mo.init_params(cfg, 20251121))
        x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(f"cuda:{rank}")
        s = cg.Smooth(clf, K, 0.25, seed=5)
        out = (s.certify(x0, 25, 39, 0.05, 32), int(s.predict(x0, 30, 0.05, 7)), s._sample_noise(x0, 11, 4).tolist())
        q.put((rank, out, dist.get_backend(), dist.get_world_size()))
        clf.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (the driver's multi-GPU node); a one-GPU box skips")
def test_two_ranks_two_gpus_rccl_match_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_run, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and got[0][2] == "nccl" and got[0][3] == 2
    single = _launch(1)[0][1]
    assert got[0][1][0] == single[0] and got[0][1][1] == int(single[1]) and got[0][1][2] == single[2]


def _nccl_one_rank(port, q):
    """The product's collective path through backend "nccl" (= RCCL) on the hardware a one-GPU box has: a process group of ONE
    rank.  A world of 1 normally skips the collective (smoothing.py `_reduces`), so `force_collective=True` makes every
    `_sample_noise` end in `dist.all_reduce` of its CUDA int64 histogram -- the call the driver's 8-GPU run makes (launch.py:110-120
    starts one process per device; SURVEY.md 8(e): one all-reduce of the vote counts)."""
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import certifiedgpt_amd as cg
    from oracle import model_oracle as mo
    from gpu_util import tiny_pair, DEV
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        K, G = 10, 3
        clf, p16, params, cfg = tiny_pair(mo.MODE_ENCODE_IMG, num_classes=K, max_batch=32)
        x0 = torch.from_numpy(mo.synthetic_image(cfg)).to(DEV)
        xs = torch.stack([x0, x0 * 0.5, x0 + 0.3])
        plain = cg.Smooth(clf, K, 0.25, seed=5)                           # world of 1: no collective
        forced = cg.Smooth(clf, K, 0.25, seed=5, force_collective=True)   # same draws, every histogram through RCCL
        assert not plain._reduces(1) and forced._reduces(1)
        calls = {"n": 0}
        real = dist.all_reduce

        def counting(t, *a, **kw):
            assert t.is_cuda and t.dtype in (torch.int64, torch.float64)
            calls["n"] += 1
            return real(t, *a, **kw)
        dist.all_reduce = counting
        try:
            def everything(s):
                return (s.certify(x0, 25, 39, 0.05, 32), int(s.predict(x0, 30, 0.05, 7)), s._sample_noise(x0, 11, 4).tolist(),
                        s.certify_many(xs, 9, 11, 0.05, 32), s.certify_images(xs, 9, 11, 0.05, 32),
                        [int(v) for v in s.predict_images(xs, 30, 0.05, 7)])
            a = everything(plain)
            assert calls["n"] == 0
            b = everything(forced)
            n_calls = calls["n"]
        finally:
            dist.all_reduce = real
        # Smooth._all_reduce itself on the [G,2,K] table of certify_many, with the timing hooks bench.py uses
        forced.collect_timing(True)
        table = (torch.arange(G * 2 * K, dtype=torch.int64, device=DEV).reshape(G, 2, K) * 7 - 3)
        want = table.clone()
        forced._all_reduce(table)
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize()
        tm = forced.timing()
        ones = torch.ones(1, device=DEV, dtype=torch.int64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        q.put({"equal": a == b, "calls": n_calls, "table_ok": bool(torch.equal(table, want)), "backend": dist.get_backend(),
               "world": dist.get_world_size(), "summed": int(ones.item()), "allreduce_ms": tm["allreduce_ms"],
               "allreduce_host_ms": tm["allreduce_host_ms"]})
        clf.close()
    finally:
        dist.destroy_process_group()


def test_nccl_product_path_runs_on_one_gpu():
    """the first time RCCL sees the product's all-reduce must not be the 8-GPU run.  One spawned rank (started
    before this process touches the GPU path of the child), backend "nccl", world_size 1: certify / predict / _sample_noise /
    certify_many / certify_images / predict_images with the collective forced give exactly the results of the same calls without
    it, and each of them really went through dist.all_reduce on a CUDA tensor."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert got["backend"] == "nccl" and got["world"] == 1 and got["summed"] == 1
    assert got["equal"] is True and got["table_ok"] is True
    assert got["calls"] == 6, got                       # one collective per call: certify, predict, _sample_noise, certify_many, 2 x images
    assert got["allreduce_ms"] >= 0.0 and got["allreduce_host_ms"] > 0.0


def test_bench_ranks_block_over_nccl_on_one_gpu():
    """`CGPT_BENCH_FORCE_NCCL=1 python bench.py --gpus 1`: bench.py's whole multi-rank code path (process group "nccl", barrier with
    device_ids, MAX over ranks, per-rank timing gather, SUM of ones, image-sharded pass) on a world of one rank."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, CGPT_BENCH_FORCE_NCCL="1", CGPT_BENCH_NO_SUSTAINED="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0",
                        "--cpu-budget-s", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["collective_backend"] == "nccl" and line["rccl_ranks"] == 1
    rk = line["ranks"]
    assert rk["summed_ranks"] == 1 and len(rk["per_rank_ms"]) == 1
    assert rk["per_rank_ms"][0]["sample_noise_calls"] >= 1 and rk["per_rank_ms"][0]["all_reduce_host"] > 0
    assert line["image_sharded"]["equals_sample_sharded"] is True
    assert line["value"] > 0
    # a line of the collective code path carries the CPU leg and the parity block too (configs[0] on the oracle and
    # on the GPU; --cpu-budget-s 0 keeps the 105-s headline leg out of the test suite and the line says so)
    _assert_cpu_leg(line, headline_leg=False)
    # ... and shows the classifier batches it ran, equal to the plan (2 images x 200 draws on a 255-sample engine)
    assert rk["batches_as_planned"] is True and rk["per_rank_ms"][0]["batch_samples"] == [[255, 1], [145, 1]]
    assert rk["per_rank_ms"][0]["host_statistics"] > 0 and rk["scaling_inputs"]["classifier_ms_per_image"] > 0


def _assert_cpu_leg(line, headline_leg):
    cb, par = line["cpu_baseline"], line["parity"]
    assert "error" not in cb, cb
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["config0_certify_s"] > 0
    assert cb["headline_leg"].startswith("ran" if headline_leg else "skipped")
    assert par["label_equal"] is True and par["abs_dR"] <= 1e-3 and par["argmax_agreement"] == "20/20"   # the north star's tolerance on R
    assert line["gpu_over_cpu"] > 1


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts its two ranks itself (a child `torch.distributed.run`) and reports
    n_gpus = 2; on this one-GPU box as a rehearsal (both ranks on cuda:0, gloo), so rccl_ranks is 0 and says so.  A request for
    more GPUs than the node has, outside the rehearsal, is refused with exit code 2 instead of measuring one GPU."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, CGPT_BENCH_ONE_GPU_REHEARSAL="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["collective_backend"] == "gloo" and line["rccl_ranks"] == 0
    # rank 0 ran the configs[0] CPU + parity leg while rank 1 waited at the final barrier (never the 105-s headline leg at N > 1)
    _assert_cpu_leg(line, headline_leg=False)
    assert "multi-rank" in line["cpu_baseline"]["headline_leg"]
    assert line["value"] > 0 and line["single_image_certify_ms"] > 0
    # a multi-rank line explains itself: per-rank classifier / all-reduce / total times and the number of ranks the collective summed over
    rk = line["ranks"]
    assert rk["summed_ranks"] == 2 and len(rk["per_rank_ms"]) == 2
    for r in rk["per_rank_ms"]:
        assert r["classifier_passes"] > 0 and r["all_reduce_device"] >= 0 and r["sample_noise_calls"] >= 1
        assert r["total"] >= r["classifier_passes"] * 0.5
    assert rk["rank_total_ms_max"] >= rk["rank_total_ms_min"] > 0
    # 3 images x 100 draws per rank on a 255-sample engine: the library's own log says [255, 45] on both ranks, as planned
    assert rk["batches_as_planned"] is True
    assert all(r["batch_samples"] == [[255, 1], [45, 1]] == r["planned_batches"] for r in rk["per_rank_ms"])
    # both partitions of SURVEY.md 8(e) in one line: the image-sharded pass certifies the same images to the same (label, radius) list
    im = line["image_sharded"]
    assert im["equals_sample_sharded"] is True and im["value"] > 0 and im["images_per_rank_max"] == 2
    if torch.cuda.device_count() < 8:
        env.pop("CGPT_BENCH_ONE_GPU_REHEARSAL")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and "refusing" in r.stderr
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_bench_four_rank_rehearsal_runs_the_planned_batches():
    """First-contact checklist for the 8-GPU node.  The GEMM shapes of an N-rank run are decided by how every
    rank's rows are cut into classifier batches; `bench.planned_batches` is that rule as host arithmetic (tests/test_distributed_cpu.py
    evaluates it at world 8: --gpus 8 --steps 20 --warmup 5 -> [255, 245] on every rank), and here a FOUR-rank
    rehearsal on this one GPU (gloo, all ranks on cuda:0) checks that what the library really ran -- its own batch log, per rank -- is
    the plan: 8 images x 50 draws per rank = 400 rows -> [255, 145].  Four ranks, not eight: the GPU boxes of this pool kill a run
    with more than six processes on the card (pytest + 4 ranks = 5)."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    sys.path.insert(0, root)
    import bench
    env = dict(os.environ, CGPT_BENCH_ONE_GPU_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "8", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    rk = line["ranks"]
    assert line["n_gpus"] == 4 and rk["summed_ranks"] == 4 and rk["batches_as_planned"] is True
    plan = bench.planned_batches(4, 0, 8, 100, 100, 255, 51)
    assert plan == [[255, 145]] * 4
    for r in rk["per_rank_ms"]:
        assert r["batch_samples"] == [[255, 1], [145, 1]] and r["classifier_batches"] == 2
    si = rk["scaling_inputs"]
    assert si["classifier_ms_per_image"] > 0 and si["host_statistics_ms_per_image"] > 0 and si["ms_per_step"] > 0
    assert line["image_sharded"]["equals_sample_sharded"] is True


def test_c_level_allreduce_counts_single_rank_communicator():
    """cgpt_allreduce_counts (the C-ABI form of the vote all-reduce, for callers without torch.distributed) on a real RCCL
    communicator.  One GPU gives a one-rank communicator (sum over one rank = identity), which still exercises symbol
    resolution, the ncclInt64 / ncclSum constants, the stream argument and in-place operation; the multi-rank sum is RCCL's."""
    import ctypes as C
    import glob
    import certifiedgpt_amd as cg
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so"]
    rccl = None
    for path in cands:
        try:
            rccl = C.CDLL(path)                              # RTLD_LOCAL, as torch maps its bundled copy: invisible to dlsym(RTLD_DEFAULT)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("no librccl.so found")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda:0")                          # HIP context
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        counts = torch.arange(2000, dtype=torch.int64, device="cuda:0") * 3 - 7
        want = counts.clone()
        L = cg.lib()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # explicit form: the ncclAllReduce of the instance that created the communicator
        fn = C.cast(rccl.ncclAllReduce, C.c_void_p)
        rc = L.cgpt_allreduce_counts_fn(fn, comm, C.c_void_p(counts.data_ptr()), counts.numel(), st)
        assert rc == 0, L.cgpt_last_error()
        torch.cuda.synchronize()
        assert torch.equal(counts, want)
        # implicit form: binds to the ONE mapped librccl (same instance) or refuses when several are mapped -- never loads another
        mapped = {os.path.realpath(l.split()[-1]) for l in open("/proc/self/maps") if "librccl.so" in l}
        rc = L.cgpt_allreduce_counts(comm, C.c_void_p(counts.data_ptr()), counts.numel(), st)
        if len(mapped) == 1:
            assert rc == 0, L.cgpt_last_error()
            torch.cuda.synchronize()
            assert torch.equal(counts, want)
        else:
            assert rc == 5 and b"instances are mapped" in L.cgpt_last_error()
        assert {os.path.realpath(l.split()[-1]) for l in open("/proc/self/maps") if "librccl.so" in l} == mapped
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
