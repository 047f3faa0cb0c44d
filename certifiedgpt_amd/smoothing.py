"""Smooth -- drop-in for the reference's randomized_smoothing/smoothing.py:13-117 on MI355X.

Same constructor and method signatures (`Smooth(base_classifier, num_classes, sigma)`, `certify(x, n0, n, alpha,
batch_size) -> (int, float)`, `predict(x, n, alpha, batch_size) -> int`, `ABSTAIN = -1`), same return
conventions.  Differences, all build-side and documented in DESIGN.md:
  * the Monte-Carlo loop (`_sample_noise`, smoothing.py:81-99) runs in libcgpt.so: counter-based Gaussian noise,
    the HIP classifier, argmax and the vote histogram stay on the GPU; one device->host copy per `_sample_noise`
    instead of one per batch (smoothing.py:98);
  * with torch.distributed initialised, the sample range of every `_sample_noise` is sharded over the ranks and the
    int64 histograms are summed with ONE all-reduce (RCCL over xGMI; gloo in CPU tests); `certify_images` is the other
    partition SURVEY.md 8(e) names: whole images per rank, no vote collective, results exchanged at 16 bytes per image;
  * the statistics (smoothing.py:46-56,73-79,107-117) are float64 host functions of the C-ABI -- no scipy /
    statsmodels needed at run time.
There is no CPU fallback: without libcgpt.so import fails, without a GPU the engine cannot be created.
"""
import ctypes as C
from math import ceil

import numpy as np
import torch

from . import _lib


class _LocalOnly:
    """`Smooth(..., process_group=LOCAL_ONLY)`: this object draws ALL of its samples on this rank and never enters a collective, whatever
    process group the process is in -- one rank of a multi-rank job can certify on its own while the others do something else (bench.py:
    rank 0's CPU / parity leg).  No torch.distributed call is made for it (in particular no `new_group`, which every rank would have to enter)."""

    def __repr__(self):
        return "LOCAL_ONLY"


LOCAL_ONLY = _LocalOnly()


def _world(group):
    if group is LOCAL_ONLY:
        return 0, 1
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(num: int, rank: int, world: int, mirrored: bool = False):
    """Samples [lo, hi) of `num` handled by `rank`: contiguous, remainder to the low ranks.
    mirrored=True gives the remainder to the HIGH ranks instead (still an exact partition of [0, num))."""
    base, rem = divmod(num, world)
    if mirrored:
        lo = rank * base + max(0, rank - (world - rem))
        return lo, lo + base + (1 if rank >= world - rem else 0)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def batch_plan(num_images: int, rows_per_image: int, max_batch: int):
    """Samples per classifier batch, in order, when `cgpt_sample_counts_images` runs `rows_per_image` draws of each of `num_images`
    images on an engine of capacity `max_batch`: the rows of all images form one image-major sequence that is cut into windows of
    `max_batch` rows, not aligned to image boundaries (csrc/model.hip).  With one image this is also `_sample_noise`'s own loop
    (smoothing.py:91-98: `this_batch_size = min(batch_size, num)`).  Pure host arithmetic: what `HipClassifier.profile_batches()`
    must report, and what a multi-GPU run's GEMM shapes will be before any GPU is touched (20 images x 25 draws on a 255-sample
    engine -> [255, 245])."""
    total = int(num_images) * int(rows_per_image)
    return [min(int(max_batch), total - r0) for r0 in range(0, total, int(max_batch))]


class Smooth(object):
    """A smoothed classifier g (smoothing.py:13)."""

    ABSTAIN = -1  # smoothing.py:17

    def __init__(self, base_classifier, num_classes: int, sigma: float, seed: int = 0, process_group=None,
                 device_stats: bool = False, non_certifiable=(), force_collective=None):
        """
        :param base_classifier: an engine exposing `sample_counts(x, first_sample, num, batch_size, sigma, seed)`
               (certifiedgpt_amd.HipClassifier), or any callable mapping a [B,C,H,W] CUDA tensor to [B,num_classes]
               logits (e.g. full MiniGPT-4 + label adapter on PyTorch-ROCm): then only noise and vote run in HIP.
        :param num_classes, sigma: as smoothing.py:19-27
        :param seed: key of the counter-based noise stream; sample indices never repeat within one Smooth object
        :param process_group: the torch.distributed group whose ranks share every `_sample_noise` (default: the world); `LOCAL_ONLY`: none --
               all samples on this rank, no collective, even inside an initialised multi-rank job
        :param device_stats: finish certify / predict on the GPU (cgpt_certify_device / cgpt_predict_device: wavefront
               arg-max, Clopper-Pearson bound, binomial test, Phi^-1 in float64) and copy back 16 bytes instead of the
               histograms; same float64 code as the host path.  Applies to `certify` and `predict`; `certify_many` finishes its
               G images' statistics on the host (one copy of the [G,2,K] table)
        :param non_certifiable: class ids that are not classes of the certificate (the "other" bucket of an answer
               vocabulary, agents/label_adapter.py): certify / predict return ABSTAIN when such a class comes out on top
        :param force_collective: run the vote all-reduce also in a process group of ONE rank (a SUM over one rank is the identity,
               so results do not change): lets a one-GPU box execute the RCCL path -- `dist.all_reduce` of the CUDA int64 histograms
               through backend "nccl" -- that a world of 1 otherwise skips.  Default: the environment variable
               CGPT_FORCE_COLLECTIVE=1, else off.  Ignored when torch.distributed is not initialised.
        """
        self.base_classifier = base_classifier
        self.num_classes = num_classes
        self.sigma = sigma
        self.seed = int(seed)
        self.process_group = process_group
        self.device_stats = bool(device_stats)
        self.non_certifiable = frozenset(int(c) for c in non_certifiable)
        if force_collective is None:
            import os
            force_collective = os.environ.get("CGPT_FORCE_COLLECTIVE", "") not in ("", "0")
        self.force_collective = bool(force_collective)
        self._next_sample = 0
        self._lib = _lib.lib()
        self._timing = None                                # see collect_timing()

    # ------------------------------------------------------------------ per-rank timing (measurement only)
    def collect_timing(self, on: bool = True):
        """Start (or stop) recording, for every `_sample_noise`-type call, HIP events around this rank's classifier pass and around
        the all-reduce, plus the host time spent inside `dist.all_reduce`.  Nothing is synchronised while recording; `timing()`
        resolves the events.  Used by bench.py so that a multi-GPU line explains itself (compute vs collective per rank)."""
        self._timing = {"compute": [], "allreduce": [], "allreduce_host_s": 0.0, "stats_host_s": 0.0, "calls": 0} if on else None

    def timing(self):
        """-> dict(compute_ms, allreduce_ms, allreduce_host_ms, calls) of this rank since collect_timing(True) (synchronises)."""
        t = self._timing
        if t is None:
            return None
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        ms = lambda pairs: float(sum(a.elapsed_time(b) for a, b in pairs))
        return {"compute_ms": ms(t["compute"]), "allreduce_ms": ms(t["allreduce"]), "allreduce_host_ms": 1e3 * t["allreduce_host_s"],
                "stats_host_ms": 1e3 * t["stats_host_s"], "calls": t["calls"]}

    def _timed_compute(self, fn):
        t = self._timing
        if t is None or not torch.cuda.is_available():
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        t["compute"].append((e0, e1))
        t["calls"] += 1
        return out

    def _reduces(self, world: int) -> bool:
        """Does this `_sample_noise` end in the collective?  Always with more than one rank; with one rank only on request."""
        if world > 1:
            return True
        if not self.force_collective or self.process_group is LOCAL_ONLY:
            return False
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()

    def _all_reduce(self, counts):
        """The ONE collective of a `_sample_noise` (C1): SUM of the int64 vote histograms over the ranks."""
        import time
        import torch.distributed as dist
        t = self._timing
        if t is None or not counts.is_cuda:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.process_group)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h0 = time.perf_counter()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.process_group)
        t["allreduce_host_s"] += time.perf_counter() - h0
        e1.record()
        t["allreduce"].append((e0, e1))

    # ------------------------------------------------------------------ reference API
    def certify(self, x: torch.tensor, n0: int, n: int, alpha: float, batch_size: int) -> (int, float):
        """smoothing.py:29-56.  Returns (predicted class, certified L2 radius) or (ABSTAIN, 0.0)."""
        self.base_classifier.eval()
        if hasattr(self.base_classifier, "sample_counts_pair"):
            # the n0 selection draws and the n estimation draws are independent (smoothing.py:44,48): run them in the same
            # classifier batches and sum both histograms with ONE all-reduce.  Same sample indices as the two-call path.
            counts_selection, counts_estimation = self._sample_noise_pair(x, n0, n, batch_size)
            if counts_estimation is None:                  # device_stats: [2, K] int64 on the GPU
                return self._certifiable(self._finalize_device(counts_selection[0], counts_selection[1], n, alpha, predict=False))
        else:
            counts_selection = self._sample_noise(x, n0, batch_size)
            counts_estimation = self._sample_noise(x, n, batch_size)
        return self._certifiable(self.certify_from_counts(counts_selection, counts_estimation, n, alpha))

    def _certifiable(self, result):
        """(label, radius) -> (ABSTAIN, 0.0) when the label is one of the non-certifiable classes."""
        if self.non_certifiable and result[0] in self.non_certifiable:
            return Smooth.ABSTAIN, 0.0
        return result

    def _sample_noise_pair(self, x, n0: int, n: int, batch_size):
        first = self._next_sample
        self._next_sample += n0 + n
        rank, world = _world(self.process_group)
        # the remainders of the two ranges go to opposite ends of the rank list: at n0 = n = 100 on 8 GPUs every rank gets
        # 13 + 12 = 25 samples instead of four ranks with 26 and four with 24 (the slowest rank sets the time)
        lo_a, hi_a = shard_range(n0, rank, world)
        lo_b, hi_b = shard_range(n, rank, world, mirrored=True)
        with torch.no_grad():
            counts = self._timed_compute(lambda: self.base_classifier.sample_counts_pair(
                x, first + lo_a, hi_a - lo_a, first + n0 + lo_b, hi_b - lo_b, batch_size, float(self.sigma), self.seed))
        if self._reduces(world):
            self._all_reduce(counts)
        if self.device_stats and counts.is_cuda:
            return counts, None                            # histograms stay on the device (certify finishes there)
        c = counts.cpu().numpy().astype(int)
        return c[0], c[1]

    def certify_many(self, xs, n0: int, n: int, alpha: float, batch_size: int):
        """`certify` for a stack of images xs[G,3,H,W]: the list of (label, radius) that G consecutive `certify` calls
        return (same sample indices, bit-identical counts), computed as ONE fused pass.  Each rank still owns its
        shard_range slice of every image's n0 and n draws, but the slices of several images share a classifier batch
        (floor(batch capacity / slice) images at a time) and all G x 2 histograms travel in one all-reduce: at 8 GPUs a rank
        runs 8 x 25-sample slices per batch instead of one, i.e. the same GEMM shapes as a single GPU."""
        bc = self.base_classifier
        G = int(xs.shape[0])
        if not hasattr(bc, "sample_counts_images") or G <= 1:
            return [self.certify(xs[i], n0, n, alpha, batch_size) for i in range(G)]
        bc.eval()
        first = self._next_sample
        self._next_sample += G * (n0 + n)
        rank, world = _world(self.process_group)
        lo_a, hi_a = shard_range(n0, rank, world)
        lo_b, hi_b = shard_range(n, rank, world, mirrored=True)
        with torch.no_grad():
            counts = self._timed_compute(lambda: bc.sample_counts_images(xs, first + lo_a, hi_a - lo_a, first + n0 + lo_b, hi_b - lo_b,
                                                                          n0 + n, float(self.sigma), self.seed))
        if self._reduces(world):
            self._all_reduce(counts)
        if self._timing is None:
            return [self._certifiable(r) for r in self.certify_many_from_counts(counts.cpu().numpy(), n, alpha)]
        # measurement: the host's share of a group of images -- the one device->host copy of the [G,2,K] table (which waits for the
        # device) is NOT in it, the float64 statistics (Clopper-Pearson bound, Phi^-1) of the G images are
        import time
        table = counts.cpu().numpy()
        h0 = time.perf_counter()
        out = [self._certifiable(r) for r in self.certify_many_from_counts(table, n, alpha)]
        self._timing["stats_host_s"] += time.perf_counter() - h0
        return out

    def certify_images(self, xs, n0: int, n: int, alpha: float, batch_size: int):
        """`certify` for a stack of images xs[G,3,H,W] (or a sequence of G image tensors), IMAGE-sharded (the zero-communication throughput mode;
        launch.py:110-120 starts one process per device): rank r certifies images shard_range(G, r, world) with ALL n0 + n draws
        of each -- no vote histogram leaves the rank -- and the G (label, radius) pairs, 16 bytes per image, are then summed
        into every rank's copy of the result table (the only collective; on one rank there is none).  Image i draws the sample
        indices the i-th of G consecutive `certify` calls would use, so the list equals that of `certify_many` and of the
        one-by-one loop on any number of ranks (counts bit-identical, statistics the same float64 code)."""
        G = len(xs)
        first = self._next_sample
        self._next_sample += G * (n0 + n)
        rank, world = _world(self.process_group)
        lo, hi = shard_range(G, rank, world)
        mine_xs, dev = self._own_images(xs, lo, hi)
        table = torch.zeros(G, 2, dtype=torch.float64, device=dev)
        if hi > lo:
            with torch.no_grad():
                c = self._timed_compute(lambda: self._counts_of_images(mine_xs, first + lo * (n0 + n), n0, n, batch_size))
            mine = [self._certifiable(r) for r in self.certify_many_from_counts(c.cpu().numpy(), n, alpha)]
            table[lo:hi] = torch.tensor(mine, dtype=torch.float64).to(table.device)
        if self._reduces(world):
            self._all_reduce(table)                        # rows of the other ranks are zero here: the SUM is a gather
        t = table.cpu().numpy()
        return [(int(t[i, 0]), float(t[i, 1])) for i in range(G)]

    def predict_images(self, xs, n: int, alpha: float, batch_size: int):
        """`predict` (smoothing.py:58-79) for a stack of images, image-sharded like `certify_images`: rank r takes whole images,
        image i draws the n indices the i-th of G consecutive `predict` calls would use; the labels (ABSTAIN = -1 included) are
        exchanged at 8 bytes per image.  Returns a list of numpy.int64 / ABSTAIN exactly as G `predict` calls return them."""
        G = len(xs)
        first = self._next_sample
        self._next_sample += G * n
        rank, world = _world(self.process_group)
        lo, hi = shard_range(G, rank, world)
        mine_xs, dev = self._own_images(xs, lo, hi)
        table = torch.zeros(G, dtype=torch.float64, device=dev)
        if hi > lo:
            with torch.no_grad():
                c = self._timed_compute(lambda: self._counts_of_images(mine_xs, first + lo * n, n, 0, batch_size))
            c = c.cpu().numpy().astype(int)
            table[lo:hi] = torch.tensor([float(self.predict_from_counts(c[i, 0], alpha)) for i in range(hi - lo)],
                                        dtype=torch.float64).to(table.device)
        if self._reduces(world):
            self._all_reduce(table)
        return [self._predicted(int(v)) for v in table.cpu().numpy()]

    @staticmethod
    def _own_images(xs, lo: int, hi: int):
        """(this rank's images [hi-lo,3,H,W] or None, device of the result table).  `xs` is a stacked tensor [G,3,H,W] or a SEQUENCE of G
        image tensors: with a sequence only the rank's own slice is stacked (an agent that walks a dataset need not materialise the
        other ranks' images on its GPU -- entries outside [lo, hi) are never touched and may be None)."""
        if isinstance(xs, torch.Tensor):
            return (xs[lo:hi] if hi > lo else None), xs.device
        own = [xs[i] for i in range(lo, hi)]
        if own:
            return torch.stack(own), own[0].device
        ref = next((t for t in xs if isinstance(t, torch.Tensor)), None)
        return None, (ref.device if ref is not None else torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                      else torch.device("cpu"))

    def _counts_of_images(self, xs, first: int, n0: int, n: int, batch_size) -> torch.Tensor:
        """int64 [g, 2, K]: selection and estimation histograms of g images, every draw on THIS rank; image i uses the indices
        first + i (n0 + n) ... (the cursor positions of consecutive `certify` calls)."""
        bc = self.base_classifier
        bc.eval()
        g = int(xs.shape[0])
        if hasattr(bc, "sample_counts_images") and g > 1:
            return bc.sample_counts_images(xs, first, n0, first + n0, n, n0 + n, float(self.sigma), self.seed)
        out = []
        for i in range(g):
            f = first + i * (n0 + n)
            if hasattr(bc, "sample_counts_pair"):
                out.append(bc.sample_counts_pair(xs[i], f, n0, f + n0, n, batch_size, float(self.sigma), self.seed))
            else:
                out.append(torch.stack([self._local_counts(xs[i], f, n0, batch_size), self._local_counts(xs[i], f + n0, n, batch_size)]))
        return torch.stack(out)

    def sample_noise_many(self, xs, num: int, batch_size, common_noise: bool = True) -> np.ndarray:
        """`_sample_noise` for a stack of images -> int array [G, num_classes].  common_noise=True: every image sees the
        SAME `num` noise draws (sample indices cursor .. cursor+num-1, consumed once): the finite differences of a query
        attack then measure the images, not the Monte-Carlo draw.  False: image i uses the next `num` indices after image
        i-1, as G consecutive `_sample_noise` calls would."""
        bc = self.base_classifier
        G = int(xs.shape[0])
        first = self._next_sample
        stride = 0 if common_noise else num
        if not hasattr(bc, "sample_counts_images"):
            out = []
            for i in range(G):
                self._next_sample = first + i * stride
                out.append(self._sample_noise(xs[i], num, batch_size))
            self._next_sample = first + (num if common_noise else G * num)
            return np.stack(out)
        self._next_sample = first + (num if common_noise else G * num)
        rank, world = _world(self.process_group)
        lo, hi = shard_range(num, rank, world)
        with torch.no_grad():
            counts = self._timed_compute(lambda: bc.sample_counts_images(xs, first + lo, hi - lo, 0, 0, stride, float(self.sigma),
                                                                          self.seed))[:, 0].contiguous()
        if self._reduces(world):
            self._all_reduce(counts)
        return counts.cpu().numpy().astype(int)

    def predict(self, x: torch.tensor, n: int, alpha: float, batch_size: int) -> int:
        """smoothing.py:58-79.  Returns the predicted class (a numpy.int64, as the reference's `top2[0]`) or ABSTAIN (int)."""
        self.base_classifier.eval()
        if self.device_stats:
            counts = self._sample_noise_device(x, n, batch_size)
            if counts.is_cuda:                             # binomial test on the GPU: 16 bytes come back instead of the histogram
                label = self._finalize_device(counts, counts, n, alpha, predict=True)
                return self._predicted(label)
            return self._predicted(self.predict_from_counts(counts.cpu().numpy().astype(int), alpha))
        counts = self._sample_noise(x, n, batch_size)
        return self._predicted(self.predict_from_counts(counts, alpha))

    def _predicted(self, label: int):
        if label == Smooth.ABSTAIN or label in self.non_certifiable:
            return Smooth.ABSTAIN                          # smoothing.py:77 returns the plain int constant
        return np.int64(label)                             # smoothing.py:79 returns an element of an int64 ndarray

    def _sample_noise_device(self, x: torch.tensor, num: int, batch_size) -> torch.Tensor:
        """`_sample_noise` up to and including the all-reduce; the histogram stays where the classifier left it (the GPU)."""
        first = self._next_sample
        self._next_sample += num
        rank, world = _world(self.process_group)
        lo, hi = shard_range(num, rank, world)
        with torch.no_grad():
            counts = self._timed_compute(lambda: self._local_counts(x, first + lo, hi - lo, batch_size))
        if self._reduces(world):
            self._all_reduce(counts)                       # the one collective (C1)
        return counts

    def _sample_noise(self, x: torch.tensor, num: int, batch_size) -> np.ndarray:
        """smoothing.py:81-99 -> ndarray[int] of length num_classes with the per-class vote counts."""
        return self._sample_noise_device(x, num, batch_size).cpu().numpy().astype(int)

    def _count_arr(self, arr: np.ndarray, length: int) -> np.ndarray:
        """smoothing.py:101-105 (host histogram; the GPU path votes in the HIP kernel instead)."""
        return np.bincount(np.asarray(arr, dtype=np.int64), minlength=length).astype(int)

    def _lower_confidence_bound(self, NA: int, N: int, alpha: float) -> float:
        """smoothing.py:107-117: Clopper-Pearson (1 - alpha) lower bound."""
        return float(self._lib.cgpt_lower_confidence_bound(int(NA), int(N), float(alpha)))

    # ------------------------------------------------------------------ pieces
    def reset(self, next_sample: int = 0):
        """Rewind the sample-index cursor (makes a certify call reproducible bit-for-bit)."""
        self._next_sample = int(next_sample)

    def _local_counts(self, x, first_sample, num, batch_size):
        bc = self.base_classifier
        if hasattr(bc, "sample_counts"):
            return bc.sample_counts(x, first_sample, num, batch_size, float(self.sigma), self.seed)
        # generic classifier on PyTorch-ROCm: HIP noise -> classifier -> HIP argmax/vote
        from .classifier import noise_batch, vote
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("Smooth needs x on a HIP device: certifiedgpt_amd has no CPU path")
        counts = torch.zeros(self.num_classes, dtype=torch.int64, device=x.device)
        done = 0
        for _ in range(ceil(num / batch_size)):
            this_batch_size = min(batch_size, num - done)
            batch = noise_batch(x.float(), first_sample + done, this_batch_size, float(self.sigma), self.seed)
            vote(bc(batch), counts)
            done += this_batch_size
        return counts

    def certify_from_counts(self, counts_selection, counts_estimation, n: int, alpha: float):
        cs = np.ascontiguousarray(counts_selection, dtype=np.int64)
        ce = np.ascontiguousarray(counts_estimation, dtype=np.int64)
        label, radius = C.c_int32(), C.c_double()
        _lib.check(self._lib.cgpt_certify_from_counts(cs.ctypes.data_as(C.c_void_p), ce.ctypes.data_as(C.c_void_p),
                                                      len(cs), int(n), float(alpha), float(self.sigma),
                                                      C.byref(label), C.byref(radius)))
        return int(label.value), float(radius.value)

    def certify_many_from_counts(self, table, n: int, alpha: float):
        """[(label, radius)] for a table of histograms [G, 2, K] (selection, estimation per image): one C call for the group."""
        t = np.ascontiguousarray(table, dtype=np.int64)
        if t.ndim != 3 or t.shape[1] != 2:
            raise ValueError("certify_many_from_counts: table must be [G, 2, num_classes]")
        G, K = int(t.shape[0]), int(t.shape[2])
        labels, radii = np.empty(G, dtype=np.int32), np.empty(G, dtype=np.float64)
        _lib.check(self._lib.cgpt_certify_many_from_counts(t.ctypes.data_as(C.c_void_p), G, K, int(n), float(alpha), float(self.sigma),
                                                           labels.ctypes.data_as(C.c_void_p), radii.ctypes.data_as(C.c_void_p)))
        return [(int(labels[i]), float(radii[i])) for i in range(G)]

    def _finalize_device(self, csel, cest, n, alpha, predict):
        out = torch.empty(2, dtype=torch.float64, device=cest.device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if predict:
            _lib.check(self._lib.cgpt_predict_device(C.c_void_p(cest.data_ptr()), cest.numel(), float(alpha),
                                                     C.c_void_p(out.data_ptr()), st))
            return int(out[0].item())
        _lib.check(self._lib.cgpt_certify_device(C.c_void_p(csel.data_ptr()), C.c_void_p(cest.data_ptr()), cest.numel(), int(n),
                                                 float(alpha), float(self.sigma), C.c_void_p(out.data_ptr()), st))
        o = out.cpu()
        return int(o[0].item()), float(o[1].item())

    def predict_from_counts(self, counts, alpha: float) -> int:
        c = np.ascontiguousarray(counts, dtype=np.int64)
        label = C.c_int32()
        _lib.check(self._lib.cgpt_predict_from_counts(c.ctypes.data_as(C.c_void_p), len(c), float(alpha), C.byref(label)))
        return int(label.value)
