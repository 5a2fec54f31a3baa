"""Data-parallel sharding of player crops over the ranks of one node + the single exchange step.

The reference has no distributed path at all; this is new design: every player is an
independent unit, so crops shard in contiguous blocks with NO data-path collective, and the only exchange is one
all-gather of the packed per-player record [cam(3) | R(216) | betas(10)] (916 B/player) after the fit --
`torch.distributed` backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  The message is tens of
KB per rank: latency-bound, so it is ONE collective on equal-sized padded blocks, not a ring of small sends."""
import os

import torch
import torch.distributed as dist


def shard_sizes(n, world):
    """Contiguous split of n players over `world` ranks, sizes differing by at most one (22 players on 8 ranks ->
    3,3,3,3,3,3,2,2)."""
    q, r = divmod(n, world)
    return [q + 1 if i < r else q for i in range(world)]


def shard_range(n, world, rank):
    sizes = shard_sizes(n, world)
    start = sum(sizes[:rank])
    return start, start + sizes[rank]


class _Done:
    """Stand-in for a completed torch.distributed Work."""
    def wait(self):
        return True


def all_gather_records(local, n_total, group=None, force_collective=False, async_op=False):
    """local: [n_local, 229] records of this rank's shard (shard_range order) -> [n_total, 229] on every rank.
    Ragged shards are padded to the largest shard so a single all_gather_into_tensor suffices.  With one rank the
    input is returned as is unless `force_collective` (rehearsal of the N>1 path on a single GPU).

    async_op=True (equal shards only) returns (tensor, work): the collective is enqueued behind the work already on
    the current stream and the caller's stream does NOT wait for it -- `work.wait()` before the tensor is read.
    (Measured with RCCL at world size 1 inside the 4-slot pipeline: 57 k players/s against 87 k with the plain
    synchronous call -- the extra bookkeeping of asynchronous work costs more than the stream wait it saves; bench.py
    uses the synchronous form.)"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        return (local, _Done()) if async_op else local
    sizes = shard_sizes(n_total, world)
    rank = dist.get_rank(group)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f'rank {rank}: got {local.shape[0]} records, shard_sizes says {sizes[rank]}')
    width = local.shape[1]
    m = max(sizes)
    padded = local
    if local.shape[0] < m:
        padded = torch.zeros((m, width), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    out = torch.empty((world * m, width), dtype=local.dtype, device=local.device)
    even = all(s == m for s in sizes)
    if async_op and even:
        work = dist.all_gather_into_tensor(out, padded.contiguous(), group=group, async_op=True)
        return out, work
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if even:
        return (out, _Done()) if async_op else out
    if async_op:
        return torch.cat([out[i * m:i * m + s] for i, s in enumerate(sizes)], dim=0), _Done()
    return torch.cat([out[i * m:i * m + s] for i, s in enumerate(sizes)], dim=0)


class RecordExchange:
    """The exchange step of a multi-GPU run: ONE all-gather of the packed records per batch, on the CURRENT stream.

    mode 'abi' (default): the library's own `spsp_allgather_records` (csrc/comm.hip: one ncclAllGather through the RCCL
    copy this process already holds) on a communicator created here -- rank 0 draws the 128-byte unique id and
    torch.distributed broadcasts it.  All ranks agree on the outcome of the communicator set-up: if it failed on any
    rank, every rank falls back to `torch.distributed.all_gather_into_tensor` and `describe()` says so.
    mode 'torch': torch.distributed only (also what the gloo CPU tests use)."""

    def __init__(self, device, mode='abi', group=None, lanes=1):
        """lanes: communicators to create (abi mode).  A pipeline that issues its collectives from several streams gives
        every stream its OWN communicator (`all_gather(..., lane=i)`): RCCL orders the collectives of one communicator
        across the streams they are issued on, which chains the streams together."""
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.mode = mode
        self.note = ''
        self._comm = None
        self._comms = []
        self._rt = None
        if mode == 'abi':
            for _ in range(max(1, int(lanes))):
                self._init_abi()
                if self.mode != 'abi':
                    break
                self._comms.append(self._comm)
            if self.mode == 'abi':
                self._comm = self._comms[0]

    # The two library calls of the set-up, separately overridable (the world-size-2 gloo test stubs them on the CPU).
    def _draw_unique_id(self):
        """rank 0: (128 id bytes, '') or (None, error text)"""
        import ctypes as C
        from . import runtime as rt
        r = rt.get(self.device)
        buf = (C.c_ubyte * 128)()
        if r.lib.spsp_comm_unique_id(buf) != 0:
            return None, r.lib.spsp_last_error().decode('utf-8', 'replace')
        return bytes(bytearray(buf)), ''

    def _init_rank(self, id_bytes):
        """every rank (collective: blocks until all ranks have called it): (communicator handle, '') or (None, error text)"""
        import ctypes as C
        from . import runtime as rt
        r = rt.get(self.device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = r.lib.spsp_comm_init_rank(r.ctx, id_bytes, self.world, self.rank, C.byref(h))
        if rc != 0:
            return None, r.lib.spsp_last_error().decode('utf-8', 'replace')
        return h, ''

    def _init_abi(self):
        """Collective.  The id travels with an OK flag (byte 128 of the broadcast): if rank 0 could not draw one, EVERY rank
        skips the blocking `spsp_comm_init_rank` and falls back together -- with the failure recorded on rank 0 alone the
        others would enter ncclCommInitRank and wait for a rank that never comes.  A second agreement (MIN all-reduce)
        follows the init itself."""
        ident = torch.zeros(129, dtype=torch.uint8, device=self.device)
        err = ''
        if self.rank == 0:
            raw, err = self._draw_unique_id()
            if raw is not None:
                ident[:128] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
                ident[128] = 1
        if dist.is_initialized():
            dist.broadcast(ident, src=0, group=self.group)
        have_id = int(ident[128].item()) == 1
        ok = torch.tensor([1 if have_id else 0], dtype=torch.int32, device=self.device)
        if have_id:
            h, err = self._init_rank(bytes(ident[:128].cpu().numpy().tobytes()))
            if h is None:
                ok.zero_()
            else:
                self._comm = h
            if dist.is_initialized():
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:          # collective decision: nobody uses a half-built communicator
            self.close()
            self.mode = 'torch'
            what = 'spsp_comm_unique_id failed on rank 0' if not have_id else 'spsp_comm_init_rank failed on a rank'
            self.note = f' ({what}{": " + err if err else ""}; every rank fell back to torch.distributed)'


    _xs = None          # the exchange's own stream (abi mode), created at the first all_gather

    def describe(self):
        how = f'spsp_allgather_records (libspsp RCCL call, {len(self._comms)} communicator(s): one per pipeline stream)' if len(self._comms) > 1 else 'spsp_allgather_records (libspsp RCCL call on the exchange stream)' if self.mode == 'abi' else 'torch.distributed.all_gather_into_tensor'
        return how + self.note

    def all_gather(self, local, n_total, wait=True, lane=None):
        """local [n_local, 229] (this rank's shard, shard_range order) -> [n_total, 229] on every rank.
        wait=True: the caller's current stream waits for the collective (the result can be used on it at once).
        lane=i (abi mode, `lanes` > 1 at construction): run on communicator i ON THE CURRENT STREAM -- for a caller that
        always uses lane i from the same stream.
        wait=False: returns (gathered, event) and leaves the caller's stream alone -- a consumer does
        `stream.wait_event(event)`; whoever owns `local` must not rewrite it before `event` (PlayerPipeline results:
        `res.hold_until(event)`).  In the four-slot pipeline the blocking form costs 14 % of throughput: the slot's next
        batch waits for a 30 KB copy that sits on a fifth hardware queue behind four saturated ones."""
        if self.mode != 'abi':
            out = all_gather_records(local, n_total, group=self.group, force_collective=True)
            if wait:
                return out
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            return out, ev
        from . import runtime as rt
        sizes = shard_sizes(n_total, self.world)
        if local.shape[0] != sizes[self.rank]:
            raise ValueError(f'rank {self.rank}: got {local.shape[0]} records, shard_sizes says {sizes[self.rank]}')
        m = max(sizes)
        padded = local.contiguous()
        if local.shape[0] < m:
            padded = torch.zeros((m, local.shape[1]), dtype=local.dtype, device=local.device)
            padded[:local.shape[0]] = local
        # The collective runs on this exchange's OWN stream, tied to the caller's stream by two events (what
        # torch.distributed's process group does as well).  Issued directly on the pipeline's four slot streams, the ONE
        # communicator orders its collectives across those streams and the slots stop overlapping: measured 72 k instead of
        # 102 k players/s at world size 1.
        if lane is not None:
            # this stream's own communicator: the collective sits in the stream's order, nothing to wait for or hold
            out = torch.empty((self.world * m, local.shape[1]), dtype=local.dtype, device=local.device)
            r = rt.get(self.device)
            r.check(r.lib.spsp_allgather_records(r.ctx, self._comms[lane], rt.ptr(padded), m, rt.ptr(out), r.stream()), 'spsp_allgather_records')
            if not all(s == m for s in sizes):
                out = torch.cat([out[i * m:i * m + s] for i, s in enumerate(sizes)], dim=0)
            if wait:
                return out
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            return out, ev
        cur = torch.cuda.current_stream(self.device)
        if self._xs is None:
            self._xs = torch.cuda.Stream(device=self.device)
        xs = self._xs
        out = torch.empty((self.world * m, local.shape[1]), dtype=local.dtype, device=local.device)   # on the caller's stream
        xs.wait_stream(cur)
        r = rt.get(self.device)
        with torch.cuda.stream(xs):
            r.check(r.lib.spsp_allgather_records(r.ctx, self._comm, rt.ptr(padded), m, rt.ptr(out), r.stream()), 'spsp_allgather_records')
        padded.record_stream(xs)
        out.record_stream(xs)
        ev = None
        if wait:
            cur.wait_stream(xs)
        else:
            ev = torch.cuda.Event()
            ev.record(xs)
        if not all(s == m for s in sizes):
            if wait:
                out = torch.cat([out[i * m:i * m + s] for i, s in enumerate(sizes)], dim=0)
            else:
                with torch.cuda.stream(xs):
                    out = torch.cat([out[i * m:i * m + s] for i, s in enumerate(sizes)], dim=0)
                    out.record_stream(cur)
                    ev.record(xs)
        return out if wait else (out, ev)

    def _destroy(self, comm):
        from . import runtime as rt
        rt.get(self.device).lib.spsp_comm_destroy(comm)

    def close(self):
        seen = []
        for c in list(self._comms) + [self._comm]:
            if c is not None and not any(c is d for d in seen):
                seen.append(c)
                self._destroy(c)
        self._comms, self._comm = [], None
