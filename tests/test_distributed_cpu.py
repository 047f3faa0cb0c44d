"""world_size-2 gloo test of the sharded `_sample_noise` path: each rank votes on its shard of the sample range,
one all-reduce sums the histograms, every rank then computes identical certify/predict results -- and the result
equals the single-process one.  The engine here is a CPU stand-in with the `sample_counts` protocol whose votes
depend only on the GLOBAL sample index (as the HIP engine's counter-based noise does); it exercises the host logic
of certifiedgpt_amd.Smooth (shard_range, cursor, all-reduce, C-ABI statistics), not the kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import certifiedgpt_amd as cg
from oracle import philox

K = 7


class IndexedEngine:
    """Votes class (philox word of the global sample index) % K; records the ranges it was asked for."""

    def __init__(self):
        self.calls = []

    def eval(self):
        return self

    def sample_counts(self, x, first_sample, num, batch_size, sigma, seed):
        self.calls.append((first_sample, num, batch_size))
        idx = np.arange(first_sample, first_sample + num, dtype=np.uint64)
        r = philox.philox4x32_10(idx, 0, 0, 0, seed, 0)[0]
        labels = np.where(r % 10 < 7, 3, r % K)           # class 3 wins ~70 %
        return torch.from_numpy(np.bincount(labels.astype(np.int64), minlength=K).astype(np.int64))


class PairEngine(IndexedEngine):
    """Adds the fused selection+estimation protocol (HipClassifier.sample_counts_pair)."""

    def sample_counts_pair(self, x, first_a, num_a, first_b, num_b, batch_size, sigma, seed):
        a = IndexedEngine.sample_counts(self, x, first_a, num_a, batch_size, sigma, seed)
        b = IndexedEngine.sample_counts(self, x, first_b, num_b, batch_size, sigma, seed)
        return torch.stack([a, b])


class ImagesEngine(PairEngine):
    """Adds the several-images protocol (HipClassifier.sample_counts_images); votes also depend on the image."""

    def sample_counts_images(self, xs, first_a, num_a, first_b, num_b, image_stride, sigma, seed):
        out = []
        for i in range(xs.shape[0]):
            out.append(self.sample_counts_pair(xs[i], first_a + i * image_stride, num_a, first_b + i * image_stride, num_b, 1 << 30, sigma, seed))
        return torch.stack(out)


def _many_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = cg.Smooth(ImagesEngine(), K, 0.5, seed=11)
        xs = torch.zeros(3, 3, 8, 8)
        q.put((rank, s.certify_many(xs, 51, 77, 0.01, 16), s._next_sample))
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_certify_many_ranks_equal_consecutive_single_process_certify(world):
    ref = cg.Smooth(IndexedEngine(), K, 0.5, seed=11)
    xs = torch.zeros(3, 3, 8, 8)
    expect = [ref.certify(xs[i], 51, 77, 0.01, 16) for i in range(3)]
    one = cg.Smooth(ImagesEngine(), K, 0.5, seed=11)
    assert one.certify_many(xs, 51, 77, 0.01, 16) == expect and one._next_sample == ref._next_sample == 3 * 128
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_many_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, cursor in got:
        assert out == expect and cursor == 3 * 128, (rank, out, expect)


def _images_worker(rank, world, port, q, engine_name):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = {"images": ImagesEngine, "pair": PairEngine, "plain": IndexedEngine}[engine_name]()
        s = cg.Smooth(eng, K, 0.5, seed=11)
        xs = torch.zeros(5, 3, 8, 8)
        out = s.certify_images(xs, 51, 77, 0.01, 16)
        calls = list(eng.calls)
        q.put((rank, out, s._next_sample, calls, s.predict_images(xs, 125, 0.001, 32)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,engine_name", [(2, "images"), (3, "images"), (2, "pair"), (3, "plain")])
def test_image_sharded_certify_equals_consecutive_single_process_certify(world, engine_name):
    """SURVEY.md 8(e), the zero-communication partition: every rank certifies WHOLE images (all 51 + 77 draws of each, at the
    sample indices consecutive certify calls use); the only collective carries (label, radius) pairs.  5 images on 2 / 3 ranks:
    ragged image shards, one rank of 3 with a single image (the pair path instead of the several-images protocol)."""
    ref = cg.Smooth(IndexedEngine(), K, 0.5, seed=11)
    xs = torch.zeros(5, 3, 8, 8)
    expect = [ref.certify(xs[i], 51, 77, 0.01, 16) for i in range(5)]
    expect_pred = [ref.predict(xs[i], 125, 0.001, 32) for i in range(5)]          # continues at the cursor certify left
    one = cg.Smooth(ImagesEngine(), K, 0.5, seed=11)
    assert one.certify_images(xs, 51, 77, 0.01, 16) == expect and one._next_sample == 5 * 128
    got_pred = one.predict_images(xs, 125, 0.001, 32)
    assert got_pred == expect_pred and [type(v) for v in got_pred] == [type(v) for v in expect_pred]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_images_worker, args=(r, world, port, q, engine_name)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    drawn = []
    for rank, out, cursor, calls, pred in got:
        assert out == expect and cursor == 5 * 128, (rank, out, expect)
        assert pred == expect_pred and [type(v) for v in pred] == [type(v) for v in expect_pred]
        drawn += [(f, n) for f, n, _ in calls]
    # every draw of every image was made exactly once, by one rank, as full 51- and 77-draw ranges (no sample sharding)
    assert sorted(drawn) == sorted([(i * 128, 51) for i in range(5)] + [(i * 128 + 51, 77) for i in range(5)])


def _single():
    s = cg.Smooth(IndexedEngine(), K, 0.5, seed=11)
    x = torch.zeros(3, 8, 8)
    return (s.certify(x, 100, 100, 0.001, 32), s.predict(x, 125, 0.001, 32), s._sample_noise(x, 10, 4).tolist(),
            s.certify(x, 51, 77, 0.01, 16))               # odd sizes: remainders go to opposite ends of the rank list


def _worker(rank, world, port, q, fused=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = PairEngine() if fused else IndexedEngine()
        s = cg.Smooth(eng, K, 0.5, seed=11)
        x = torch.zeros(3, 8, 8)
        out = (s.certify(x, 100, 100, 0.001, 32), s.predict(x, 125, 0.001, 32), s._sample_noise(x, 10, 4).tolist(),
               s.certify(x, 51, 77, 0.01, 16))
        q.put((rank, out, eng.calls))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_matches_single_process():
    expect = _single()
    assert expect[0][0] == 3 and expect[0][1] > 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, calls in got:
        assert out == expect, (rank, out, expect)          # identical label / radius / counts on every rank
    # rank 0 took [0,50) of the selection pass, rank 1 [50,100); estimation pass starts at global index 100
    assert got[0][2][:2] == [(0, 50, 32), (100, 50, 32)]
    assert got[1][2][:2] == [(50, 50, 32), (150, 50, 32)]
    # predict: 125 samples -> 63 + 62, starting at cursor 200
    assert got[0][2][2] == (200, 63, 32) and got[1][2][2] == (263, 62, 32)


def test_fused_certify_pass_matches_two_pass_single_and_two_ranks():
    """certify through sample_counts_pair (one pass, one all-reduce for both histograms) == the two-call path."""
    expect = _single()
    s = cg.Smooth(PairEngine(), K, 0.5, seed=11)
    x = torch.zeros(3, 8, 8)
    assert (s.certify(x, 100, 100, 0.001, 32), s.predict(x, 125, 0.001, 32), s._sample_noise(x, 10, 4).tolist(),
            s.certify(x, 51, 77, 0.01, 16)) == expect
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, calls in got:
        assert out == expect, (rank, out, expect)
    # rank 1: selection shard [50,100), estimation shard [150,200) -- same indices as the unfused path
    assert got[1][2][:2] == [(50, 50, 32), (150, 50, 32)]
    # odd sizes in the fused pass (cursor at 335): n0 = 51 -> 26 + 25 (remainder to rank 0), n = 77 -> 38 + 39 (mirrored:
    # remainder to rank 1), i.e. 64 samples on both ranks
    assert got[0][2][-2:] == [(335, 26, 16), (386, 38, 16)]
    assert got[1][2][-2:] == [(361, 25, 16), (424, 39, 16)]


# ------------------------------------------------------------------ text-generating classifier + answer vocabulary, 2 ranks
class AnswerEngine:
    """A generic callable classifier (Smooth's non-engine path is GPU-only, so this stand-in implements sample_counts itself):
    the 'generated answer' of global sample i is one of a few strings whose ORDER OF FIRST APPEARANCE depends on where a
    rank's shard starts -- exactly the situation in which a growing label map gives different ids on different ranks."""
    ANSWERS = ["a red bus", "two dogs", "The Bus.", "dont know", "stop sign", "2 dogs"]

    def __init__(self, label_map):
        self.label_map = label_map

    def eval(self):
        return self

    def sample_counts(self, x, first_sample, num, batch_size, sigma, seed):
        idx = np.arange(first_sample, first_sample + num)
        texts = [self.ANSWERS[(i * 7 + i // 3) % len(self.ANSWERS)] for i in idx]
        ids = [self.label_map(t) for t in texts]
        return torch.from_numpy(np.bincount(ids, minlength=self.label_map.num_classes).astype(np.int64))


def _label_worker(rank, world, port, q):
    from certifiedgpt_amd.agents.label_adapter import AnswerLabelMap
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.zeros(3, 8, 8)
        frozen = AnswerLabelMap(5, ["red bus", "2 dogs", "bus", "stop sign"])      # vocabulary up front -> frozen
        counts = cg.Smooth(AnswerEngine(frozen), 5, 0.5)._sample_noise(x, 60, 16).tolist()
        try:
            cg.Smooth(AnswerEngine(AnswerLabelMap(5)), 5, 0.5)._sample_noise(x, 60, 16)
            grew = "no error"
        except RuntimeError as e:
            grew = str(e)
        q.put((rank, counts, grew))
    finally:
        dist.destroy_process_group()


def test_answer_vocabulary_is_rank_independent_and_growth_is_refused():
    from certifiedgpt_amd.agents.label_adapter import AnswerLabelMap
    frozen = AnswerLabelMap(5, ["red bus", "2 dogs", "bus", "stop sign"])
    single = cg.Smooth(AnswerEngine(frozen), 5, 0.5)._sample_noise(torch.zeros(3, 8, 8), 60, 16).tolist()
    assert sum(single) == 60 and single[4] > 0                      # "dont know" -> "don't know" -> other
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_label_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] == single                          # same ids on every rank: the all-reduce sums like with like
    assert all("would grow under torch.distributed" in g[2] for g in got)


def _forced_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        calls = {"n": 0}
        real = dist.all_reduce

        def counting(t, *a, **kw):
            calls["n"] += 1
            return real(t, *a, **kw)
        dist.all_reduce = counting
        xs = torch.zeros(3, 3, 8, 8)

        def everything(s):
            return (s.certify(xs[0], 51, 77, 0.01, 16), int(s.predict(xs[0], 125, 0.001, 32)), s._sample_noise(xs[0], 40, 8).tolist(),
                    s.certify_many(xs, 51, 77, 0.01, 16), s.certify_images(xs, 51, 77, 0.01, 16))
        plain = everything(cg.Smooth(ImagesEngine(), K, 0.5, seed=11))
        n_plain = calls["n"]
        forced = everything(cg.Smooth(ImagesEngine(), K, 0.5, seed=11, force_collective=True))
        q.put((plain == forced, n_plain, calls["n"]))
    finally:
        dist.destroy_process_group()


def test_force_collective_runs_the_all_reduce_in_a_world_of_one():
    """`force_collective=True` (or CGPT_FORCE_COLLECTIVE=1): a process group of ONE rank still ends every `_sample_noise` in the
    all-reduce -- the switch tests/test_gpu_distributed.py uses to run the RCCL path on a one-GPU box.  Results are unchanged,
    and without torch.distributed initialised the switch does nothing."""
    s = cg.Smooth(ImagesEngine(), K, 0.5, seed=11, force_collective=True)
    assert s._reduces(1) is False and s._reduces(2) is True           # no process group in this process
    os.environ["CGPT_FORCE_COLLECTIVE"] = "1"
    try:
        assert cg.Smooth(ImagesEngine(), K, 0.5).force_collective is True
    finally:
        del os.environ["CGPT_FORCE_COLLECTIVE"]
    assert cg.Smooth(ImagesEngine(), K, 0.5).force_collective is False
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    same, n_plain, n_total = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same is True and n_plain == 0 and n_total == 5


def _list_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = cg.Smooth(ImagesEngine(), K, 0.5, seed=11)
        G = 5
        lo, hi = cg.shard_range(G, rank, world)
        # a rank that walks a dataset only needs ITS images: the others' entries are never touched (None here)
        xs = [torch.zeros(3, 8, 8) if lo <= i < hi else None for i in range(G)]
        out = s.certify_images(xs, 51, 77, 0.01, 16)
        pred = [int(v) for v in s.predict_images(xs, 125, 0.001, 32)]
        q.put((rank, out, pred, s._next_sample))
    finally:
        dist.destroy_process_group()


def test_image_sharded_calls_accept_a_list_and_touch_only_the_ranks_own_images():
    """in shard = images mode a rank need not materialise the other ranks' images.  `certify_images` / `predict_images` take a
    SEQUENCE of image tensors and stack only the slice shard_range gives the rank -- entries outside it may even be None -- and still
    return the list a stacked tensor gives on one rank."""
    ref = cg.Smooth(ImagesEngine(), K, 0.5, seed=11)
    xs = torch.zeros(5, 3, 8, 8)
    want = ref.certify_images(xs, 51, 77, 0.01, 16)
    want_pred = [int(v) for v in ref.predict_images(xs, 125, 0.001, 32)]
    one = cg.Smooth(ImagesEngine(), K, 0.5, seed=11)
    assert one.certify_images([xs[i] for i in range(5)], 51, 77, 0.01, 16) == want          # list form on one rank
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_list_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, pred, cursor in got:
        assert out == want and pred == want_pred and cursor == ref._next_sample, (rank, out, want)


def test_eight_gpu_batch_plan_is_the_shapes_of_design_section_6():
    """VERDICT r5 item 5a, the part that needs no GPU: the per-rank classifier batches of the driver's 8-GPU command
    (`bench.py --gpus 8 --steps 20 --warmup 5`, n0 = n = 100, 255-sample engines, up to 51 images per certify_many call) are
    20 images x 25 draws = 500 rows -> [255, 245] on EVERY rank (13 + 12 and 12 + 13 draws: the mirrored remainders), the warm-up's
    5 images -> [125]; N = 1 runs 15 full 255-sample batches + 175.  bench.py prints the same plan beside the batches the library
    logged (`ranks.per_rank_ms[*].planned_batches` / `batch_samples`); tests/test_gpu_distributed.py checks on the GPU, at four
    ranks, that the log equals the plan.  (smoothing.py:91-98 is the loop whose batch sizes these are.)"""
    import bench
    import certifiedgpt_amd as cg
    assert [bench.rank_share(100, 100, r, 8) for r in range(8)] == [25] * 8
    assert bench.planned_batches(8, 5, 25, 100, 100, 255, 51) == [[255, 245]] * 8
    assert bench.planned_batches(8, 0, 5, 100, 100, 255, 51) == [[125]] * 8
    one = bench.planned_batches(1, 5, 25, 100, 100, 255, 51)[0]
    assert one == [255] * 15 + [175] and sum(one) == 20 * 200
    # 2 and 4 GPUs: 100 / 50 draws per image and rank
    assert bench.planned_batches(2, 5, 25, 100, 100, 255, 51) == [[255] * 7 + [215]] * 2
    assert bench.planned_batches(4, 5, 25, 100, 100, 255, 51) == [[255] * 3 + [235]] * 4
    # groups of 51 images are separate calls: 60 images on one GPU = 51 + 9
    assert bench.planned_batches(1, 0, 60, 100, 100, 255, 51)[0] == [255] * 40 + [255] * 7 + [15]
    # ragged shares (N = 1000 over 8 GPUs is 125 + 13/12 of n0 = 100) and the one-image call
    assert cg.batch_plan(1, 138, 200) == [138] and cg.batch_plan(3, 138, 200) == [200, 200, 14] and cg.batch_plan(0, 5, 8) == []


def _local_only_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = None
        if rank == 0:                                     # rank 1 does NOT take part: a collective here would hang until the timeout
            eng = ImagesEngine()
            s = cg.Smooth(eng, K, 0.5, seed=11, process_group=cg.LOCAL_ONLY, force_collective=True)
            xs = torch.zeros(3, 3, 8, 8)
            out = (s.certify(xs[0], 51, 77, 0.01, 16), int(s.predict(xs[0], 125, 0.001, 32)), s.certify_many(xs, 51, 77, 0.01, 16),
                   s.certify_images(xs, 51, 77, 0.01, 16), eng.calls[:2])
        dist.barrier()                                    # both ranks meet again only here
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_local_only_smooth_never_enters_a_collective_inside_a_multi_rank_job():
    """bench.py at N > 1: rank 0 runs its CPU / parity leg through `Smooth(..., process_group=LOCAL_ONLY)` while the other ranks wait at
    the final barrier.  In a two-rank gloo job rank 0 alone certifies / predicts / certifies groups of images with such an object -- all
    draws on rank 0 (the engine is asked for the WHOLE ranges), no all-reduce even with force_collective -- and gets what a
    single-process Smooth gets; rank 1 goes straight to the barrier."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_local_only_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    eng = ImagesEngine()
    s = cg.Smooth(eng, K, 0.5, seed=11)
    xs = torch.zeros(3, 3, 8, 8)
    want = (s.certify(xs[0], 51, 77, 0.01, 16), int(s.predict(xs[0], 125, 0.001, 32)), s.certify_many(xs, 51, 77, 0.01, 16),
            s.certify_images(xs, 51, 77, 0.01, 16), eng.calls[:2])
    assert got[1] is None and got[0] == want
    assert want[4][0][:2] == (0, 51) and want[4][1][:2] == (51, 77)      # whole ranges: nothing was sharded away
    assert repr(cg.LOCAL_ONLY) == "LOCAL_ONLY"
