"""CPU, world_size 2 over gloo: the N > 1 path -- contiguous player sharding and the single all-gather of the
packed [cam | R | betas] records, including the ragged (22-player frame) case."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.sharding import all_gather_records, shard_range, shard_sizes


def test_shard_sizes_cover_everything():
    assert shard_sizes(22, 8) == [3, 3, 3, 3, 3, 3, 2, 2]
    assert shard_sizes(256, 8) == [32] * 8
    for n in (0, 1, 7, 22, 255):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _records(n):
    R, b, cam = S.pose_params(n, seed=5)
    return np.concatenate([cam, R.reshape(n, 216), b], axis=1).astype(np.float32)


def _worker(rank, world, port, n_total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        full = torch.from_numpy(_records(n_total))
        lo, hi = shard_range(n_total, world, rank)
        out = all_gather_records(full[lo:hi].clone(), n_total)
        out2, work = all_gather_records(full[lo:hi].clone(), n_total, async_op=True)
        work.wait()
        q.put((rank, bool(torch.equal(out, full)) and bool(torch.equal(out2, full)), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n_total, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_all_gather_even_shards_world2():
    for rank, ok, shape in _run(64):
        assert ok and shape == (64, 229)


def test_all_gather_ragged_shards_world2():
    for rank, ok, shape in _run(23):
        assert ok and shape == (23, 229)
