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
