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


def _exchange_worker(rank, world, port, fail_at, q):
    """RecordExchange set-up with the two library calls stubbed (no GPU here): `fail_at` = 'id' (rank 0 cannot draw the
    unique id), 'init' (rank 1's communicator init fails) or None."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from soccerplayershapepose_amd.sharding import RecordExchange
        calls = []

        class Stubbed(RecordExchange):
            def _draw_unique_id(self):
                calls.append('id')
                return (None, 'stub: no RCCL') if fail_at == 'id' else (bytes(range(128)), '')

            def _init_rank(self, id_bytes):
                calls.append('init')
                assert id_bytes == bytes(range(128))
                return (None, 'stub: init failed') if (fail_at == 'init' and self.rank == 1) else (object(), '')

            def _destroy(self, comm):
                calls.append('destroy')

        ex = Stubbed('cpu', mode='abi')
        full = torch.from_numpy(_records(10))
        lo, hi = shard_range(10, world, rank)
        gathered = ex.all_gather(full[lo:hi].clone(), 10) if ex.mode == 'torch' else None
        q.put((rank, ex.mode, list(calls), ex.describe(), gathered is None or bool(torch.equal(gathered, full))))
    finally:
        dist.destroy_process_group()


def _run_exchange(fail_at):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, fail_at, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)      # a rank stuck in a one-sided collective would time out here
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_record_exchange_setup_failures_fall_back_on_every_rank_together_world2():
    """rank 0 failing to draw the RCCL unique id used to be known to rank 0 only -- the other ranks went
    into the blocking communicator init alone.  Now the id is broadcast with an OK flag: nobody calls init, everybody falls
    back to torch.distributed, and the fallback exchange works.  Same agreement when the init fails on one rank."""
    (r0, m0, c0, d0, ok0), (r1, m1, c1, d1, ok1) = _run_exchange('id')
    assert m0 == m1 == 'torch' and c0 == ['id'] and c1 == [] and ok0 and ok1
    assert 'spsp_comm_unique_id failed on rank 0' in d0 and 'fell back' in d1
    (r0, m0, c0, d0, ok0), (r1, m1, c1, d1, ok1) = _run_exchange('init')
    assert m0 == m1 == 'torch' and c0 == ['id', 'init', 'destroy'] and c1 == ['init'] and ok0 and ok1
    (r0, m0, c0, d0, _), (r1, m1, c1, d1, _) = _run_exchange(None)
    assert m0 == m1 == 'abi' and c0 == ['id', 'init'] and c1 == ['init']


def test_bench_launcher_starts_the_ranks_itself_world2():
    """`python bench.py --gpus 2` with no launcher around it must itself start 2 ranks (fresh children, before any GPU
    call) and relay ONE JSON line saying n_gpus 2 / dp2.  --cpu-rehearsal swaps the GPU work for the exchange step on
    gloo, so the launcher, the rendezvous, the padded all-gather and the max-over-ranks timing run here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--cpu-rehearsal'], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 64
    assert d['exchange_ok'] is True and d['steps'] == 3
    # a rank count the launcher did not create is refused
    env2 = dict(env, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    bad = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--cpu-rehearsal'], capture_output=True,
                         text=True, timeout=120, env=env2)
    assert bad.returncode != 0 and 'WORLD_SIZE=3' in bad.stderr
