#!/usr/bin/env python3
"""bench.py -- players/sec end-to-end (backbone -> rot6d -> 50-iter pose+shape fit -> SMPL forward -> vertices)
on synthetic 256x256 proxy crops, one process per GPU.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself (torch.distributed.run, as
fresh children, BEFORE anything here touches the GPU) and relays rank 0's line; under an existing launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) it is one of the ranks.

A "step" is one pass of the hot path over one batch of 32 crops per GPU (BASELINE.json configs[1]), inputs
already resident in HBM in the backbone-native layout.  With N > 1 every rank processes its own 32 crops (weak
scaling, crops shard data-parallel with no data-path collective) and the step ends with ONE RCCL all-gather of the
packed per-player record [cam | R | betas] = 229 fp32, issued through the library's own
spsp_allgather_records on the slot's stream.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import re
import socket
import subprocess
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One hardware queue per pipeline slot: HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues
# round-robin, and two slots sharing a queue serialise their whole chains.  Must be set before HIP initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BATCH_PER_GPU = 32
FRAME_PLAYERS = 22                      # BASELINE.json configs[3]: one match frame
FIT_ITERS = 50
# algorithmic work per player (SURVEY 8d / BASELINE.md section 4)
FLOP_CONV1 = 2 * 924_844_032            # conv1 7x7/2, 18 -> 64, real (unpadded) MACs
FLOP_L1CONV = 2 * 150_994_944           # one 3x3 stride-1 conv of ResNet-18 (every one has the same MACs per crop)
FLOP_BACKBONE = 2 * 3_139_436_544
MFMA_PEAK_TFLOPS = 2500.0               # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='crops per GPU per step')
    ap.add_argument('--fit', default='pose_shape', choices=['pose_shape', 'camera'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--depth', type=int, default=4, help='pipeline slots (HIP streams); 1 = strictly serial steps')
    ap.add_argument('--precision', default='f16', choices=['bf16', 'f16'],
                    help='backbone storage / MFMA element type.  f16 (default): the mode whose vertices stay within the '
                         'north-star 1e-3 m of the all-fp32 CPU path; bf16: same bytes and MFMA rate, ~1.5e-3 m')
    ap.add_argument('--no-alt-precision', action='store_true', help='skip the second (other element type) throughput figure')
    ap.add_argument('--tail-group', type=int, default=2,
                    help='also report the throughput with the latency-bound tail (IEF -> fit -> SMPL) run once per this many consecutive '
                         'batches of a slot (the convolutions keep their batch; PlayerPipeline.submit_group); 1 = skip')
    ap.add_argument('--from-detections', action='store_true',
                    help='each step also builds the native proxy input on the GPU from silhouettes [B,512,512] u8 + joints2D (SURVEY 8f-1)')
    ap.add_argument('--rehearse-dist', action='store_true',
                    help='run the N>1 code path (RCCL communicators, per-step all-gather, barriers) even with one rank')
    ap.add_argument('--plan-hooks', default='', help=argparse.SUPPRESS)         # experiment: 'what=value,...' (SPSP_DEBUG_PLAN_*) on every slot's regressor
    ap.add_argument('--probe', action='store_true', help=argparse.SUPPRESS)       # diagnostic: re-time the pipeline after the timed region (stderr)
    ap.add_argument('--profile-pass', default='', choices=['', 'latency', 'throughput'],
                    help='run ONLY the untimed per-kernel pass of this regressor plan (one batch in flight, a synchronize after every '
                         'step, no event pairs) and exit without a JSON line: the command tools/gpu_profile.sh puts under rocprofv3 '
                         '--kernel-trace --stats, so that the tracked trace (profiles/kernel_stats_isolated_<plan>_plan.csv) and '
                         'roofline.avg_us measure the same launches')
    ap.add_argument('--inputs', type=int, default=0,
                    help='distinct resident input batches rotated through the steps (0 = one per pipeline slot, at least 4): every step '
                         'reads another 80 MB buffer, so the input is streamed from HBM and not re-read from the Infinity Cache')
    ap.add_argument('--exchange-lanes', type=int, default=1,
                    help='RCCL communicators of the exchange step.  1 (default): ONE communicator on the exchange\'s own stream, the '
                         'slot\'s next batch held until the collective has read the record (no cross-stream ordering between ranks to '
                         'get wrong).  N > 1: one communicator per pipeline stream, collectives issued on the slot streams -- measured '
                         'at world size 1 only (DESIGN 6); concurrent communicators at N > 1 are opt-in until an N-GPU run has exercised them')
    ap.add_argument('--exchange', default='abi', choices=['abi', 'torch'],
                    help='N>1 exchange step: abi = spsp_allgather_records (the library\'s RCCL call on the slot stream), '
                         'torch = torch.distributed.all_gather_into_tensor')
    ap.add_argument('--no-graphs', dest='graphs', action='store_false',
                    help='eager launches in the timed region (then the dominant kernel keeps a HIP-event pair there).  Default: every '
                         'slot replays its step from a captured HIP graph -- ~20 us of host time per batch instead of ~180 us, so the '
                         'figure does not depend on how busy the host is (measured on a loaded box: eager 65 k vs graphs 93 k players/s)')
    ap.add_argument('--cpu-fit-sample', type=int, default=16, help='players fitted by the CPU oracle (0.15-0.2 s each)')
    ap.add_argument('--cpu-threads', type=int, default=0, help='host threads for the CPU oracle; 0 = all physical cores this process may use')
    ap.add_argument('--cpu-rehearsal', action='store_true',
                    help='launcher / exchange rehearsal on gloo without a GPU (CPU test of the --gpus N path): no kernels run, value is null')
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """--gpus N without a launcher: start N ranks as FRESH children (torch.distributed.run) and relay rank 0's JSON
    line.  This parent never initialises the GPU (device_count() only counts devices on this image), never execs."""
    if not args.cpu_rehearsal:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in p.stdout.decode('utf-8', 'replace').splitlines() if ln.startswith('{') and '"metric"' in ln]
    if p.returncode != 0 or not lines:
        sys.stderr.write(p.stdout.decode('utf-8', 'replace'))
        raise SystemExit(p.returncode or 1)
    sys.stdout.write(lines[-1] + '\n')
    sys.stdout.flush()
    raise SystemExit(0)


def rocprof_isolated_path(plan):
    return os.path.join('profiles', f'kernel_stats_isolated_{plan}_plan.csv')


def rocprof_average_us(table_name, plan):
    """Average duration of the library kernel `table_name` (e.g. 'conv3x3_patch_kernel<false>') in the tracked rocprofv3
    kernel trace of `bench.py --profile-pass <plan>` (this file's own per-kernel pass, one batch in flight; profiles/README.md), or
    None.  The trace names carry every template argument; the table name's arguments are a prefix of them."""
    import csv
    path = os.path.join(ROOT, rocprof_isolated_path(plan))
    if not os.path.exists(path):
        return None
    base, _, targs = table_name.partition('<')
    want = base + ('<' + targs.rstrip('>') if targs else '')
    calls = tot = 0.0
    for row in csv.DictReader(open(path)):
        name = row['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        name = re.sub(r'^conv1_pool32r?_kernel', 'conv1_pool_kernel', name)        # the stem's forms are one table row
        name = re.sub(r'^ief_poll_kernel<\d+>', 'ief_fused_kernel', name)          # ... and the IEF's
        if name.startswith(want) and (not targs or name[len(want)] in ',>'):
            calls += float(row['Calls'])
            tot += float(row['TotalDurationUs'])
    return tot / calls if calls else None


def host_threads(requested):
    """All physical cores this process may run on (the GPU box gives one GPU's CPU share), or the explicit request."""
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    if requested > 0:
        return max(1, min(requested, allowed))
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or allowed
    except Exception:
        phys = allowed
    return max(1, min(phys, allowed))


def cpu_rehearsal(args, rank, world):
    """The launcher + exchange path on gloo with no GPU: shards, padded all-gather, barriers, max-over-ranks time."""
    from soccerplayershapepose_amd.sharding import all_gather_records, shard_range
    dist.init_process_group('gloo')
    B = args.batch
    g = torch.Generator().manual_seed(7)
    full = torch.rand((world * B, 229), generator=g)
    lo, hi = shard_range(world * B, world, rank)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = all_gather_records(full[lo:hi].clone(), world * B)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ok = bool(torch.equal(out, full))
    # ragged frame (22 players -> 3,3,3,... per rank), padded to equal blocks
    flo, fhi = shard_range(FRAME_PLAYERS, world, rank)
    frame = all_gather_records(full[flo:fhi].clone(), FRAME_PLAYERS)
    ok = ok and bool(torch.equal(frame, full[:FRAME_PLAYERS]))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    line = {'metric': 'players/sec end-to-end (backbone+SMPL+fit) on 256x256 crops', 'value': None, 'unit': 'players/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * el.item() / max(1, args.steps),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision,
            'data': 'launcher rehearsal on gloo: no GPU work, exchange step only',
            'config': {'workload': 'launcher + exchange rehearsal', 'crops_per_gpu': B, 'global_batch': world * B,
                       'parallelism': f'dp{world}'}, 'exchange_ok': bool(flag.item())}
    dist.destroy_process_group()
    return line if rank == 0 else None


def cpu_baseline(sd, model, x_nchw, tgt, n_fit, all_cores, gpu_pre, gpu_post):
    """The oracle (restated reference path, PyTorch-CPU fp32: full SMPL.forward + autograd + torch.optim.Adam per fit
    iteration) timed on this box's host cores, stage by stage: backbone+IEF at B=1 and B=32, the 50-iteration
    fit per player, the SMPL forward -- on ALL cores this process may use (printed), on 16 (one GPU's nominal CPU share on
    this pool) and on ONE.  The small ops of the fit do not scale with threads, so the three differ a lot; `value` is the
    fastest of them, `cores` the threads it used, `stages` holds all three.  It is the CHECKER timed as a reported
    baseline, and the same outputs give the bench line its own parity figures against the GPU result (gpu_pre: vertices of
    regressor -> SMPL, gpu_post: vertices after the fit, both [B,6890,3] for the players of the batch)."""
    from soccerplayershapepose_amd import synthetic as S
    from oracle import regressor as OR, geometry as G, fit as OF, smpl as OS
    osmpl = OS.OracleSMPL(model)
    mp = S.mean_params()
    B = x_nchw.shape[0]
    n_fit = max(1, min(n_fit, B))

    def timeit(fn, warm, reps):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    cam, pose, shape = OR.regressor_forward(sd, x_nchw, mp)
    R = G.rot6d_to_rotmat(pose).view(B, 24, 3, 3)
    with torch.no_grad():
        pre = osmpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=shape).vertices
    post = {}

    def measure(threads, fits):
        torch.set_num_threads(threads)
        threads = torch.get_num_threads()
        t_b32 = timeit(lambda: OR.regressor_forward(sd, x_nchw, mp), 1, 2)
        t_b1 = timeit(lambda: OR.regressor_forward(sd, x_nchw[:1], mp), 2, 4)
        with torch.no_grad():
            t_s32 = timeit(lambda: osmpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=shape), 1, 3)
            t_s1 = timeit(lambda: osmpl(body_pose=R[:1, 1:], global_orient=R[:1, :1], betas=shape[:1]), 2, 4)
        t_fit = []
        for i in range(fits):               # one player at a time, as the reference drives it
            t0 = time.perf_counter()
            r = OF.fit_pose_shape(osmpl, R[i:i + 1, 1:], R[i:i + 1, :1], shape[i:i + 1], tgt[i], cam[i:i + 1], iters=FIT_ITERS)
            t_fit.append(time.perf_counter() - t0)
            if i not in post:
                body, Gm, be, ca, _ = r['last']
                with torch.no_grad():
                    post[i] = osmpl(body_pose=body, global_orient=Gm, betas=be).vertices[0]
        t_f = float(np.median(t_fit[1:] if len(t_fit) > 1 else t_fit))
        per_player = t_b32 / B + t_f + t_s32 / B            # best schedule: batched backbone + SMPL, per-player fit
        return {'threads': threads, 'backbone_ief_B32_s': t_b32, 'backbone_ief_B1_s': t_b1, 'fit_50it_per_player_s': t_f,
                'fit_players_timed': fits, 'smpl_forward_B32_s': t_s32, 'smpl_forward_B1_s': t_s1,
                'players_per_s': 1.0 / per_player}

    plan = [(16 if all_cores >= 16 else all_cores, n_fit)]
    if all_cores != plan[0][0]:
        plan.append((all_cores, 2))
    if plan[0][0] != 1:
        plan.append((1, 2))
    stages = [measure(t, f) for t, f in plan]
    torch.set_num_threads(plan[0][0])
    best = max(stages, key=lambda d: d['players_per_s'])
    # parity of THIS run's GPU result against the all-fp32 CPU path, same players
    pre_dev = (gpu_pre.cpu() - pre).abs().amax(dim=(1, 2))
    post_dev = torch.stack([(gpu_post[i].cpu() - post[i]).abs().max() for i in sorted(post)])
    parity = {'reference': 'all-fp32 CPU oracle, same crops and fit targets', 'tolerance_m': 1e-3,
              'pre_fit_players': int(B), 'pre_fit_max_m': float(pre_dev.max()),
              'post_fit_players': len(post), 'post_fit_median_m': float(post_dev.median()), 'post_fit_max_m': float(post_dev.max())}
    base = {'value': best['players_per_s'], 'unit': 'players/s', 'cores': best['threads'], 'kind': 'port',
            'host_cores_available': all_cores,
            'sample': f'torch-CPU fp32 oracle: backbone+IEF on {B} crops (B=32 x2, B=1 x4), the {FIT_ITERS}-iteration fit on '
                      f'{n_fit} players (2 at the other thread counts), SMPL forward B=32 x3 / B=1 x4; timed on '
                      f'{[d["threads"] for d in stages]} threads, value = the fastest', 'stages': stages}
    return base, parity


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.from_detections:
        args.graphs = False          # the proxy builder hands every step a fresh input buffer: nothing fixed to capture
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        launch_ranks(args, argv)
    # stdout carries exactly ONE line, the JSON result: native libraries write banners there (RCCL prints its version
    # block to stdout at communicator creation), so fd 1 is pointed at stderr for the duration of the run and the
    # result line goes to a saved copy of the original stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    if args.cpu_rehearsal:
        line = cpu_rehearsal(args, rank, world)
        if line is not None:
            os.write(result_fd, (json.dumps(line) + '\n').encode())
        return
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f'bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.rehearse_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)      # nccl == RCCL on ROCm

    from soccerplayershapepose_amd import _abi, synthetic as S
    from soccerplayershapepose_amd import runtime as rt
    from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
    from soccerplayershapepose_amd.models.smpl_official import SMPL
    from soccerplayershapepose_amd.player_recon import PlayerPipeline, reconstruct_players
    from soccerplayershapepose_amd.sharding import RecordExchange, shard_range
    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat

    B = args.batch
    sd = S.regressor_state_dict(seed=0)
    model = S.smpl_model(seed=0)

    def make_pipe(precision, streams=None):
        mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=precision).load_state_dict(sd)
        mk_s = lambda: SMPL(model, max_batch=B)
        p_ = PlayerPipeline(mk_r, mk_s, dev, depth=max(1, args.depth), streams=streams)
        for kv in filter(None, args.plan_hooks.split(',')):
            what, val = (int(v) for v in kv.split('='))
            for sl in p_.slots:
                rt.get(dev).check(rt.get(dev).lib.spsp_regressor_debug(sl.regressor.handle, what, val), 'spsp_regressor_debug')
        return p_

    pipe = make_pipe(args.precision)
    reg, smpl = pipe.slots[0].regressor, pipe.slots[0].smpl
    r = rt.get(dev)
    lanes = max(1, min(args.exchange_lanes, len(pipe.slots)))
    exchange = RecordExchange(dev, mode=args.exchange, lanes=lanes) if use_dist else None
    lane_of = {id(sl.stream): i % lanes for i, sl in enumerate(pipe.slots)}      # (lanes > 1) a slot's stream -> its communicator

    # ---- synthetic inputs (seed 1234 + rank + 1000 * buffer), resident in HBM in the backbone-native layout.  SEVERAL
    # distinct batches rotate through the steps (one per pipeline slot, at least 4: 4 x 80 MB > the 256 MB Infinity Cache),
    # each with its own fit targets: regressor-initialised model, perturbed + 1 px noise; set-up, untimed ----
    n_inputs = args.inputs if args.inputs > 0 else max(4, len(pipe.slots))

    def make_input(k):
        x_nchw_, joints_in_ = S.proxy_crops(B, seed=1234 + rank + 1000 * k)
        x_dev_ = torch.from_numpy(x_nchw_).to(dev)
        native_ = reg.pack(x_dev_)
        cam0, pose0, shape0 = reg(native_)
        R0 = rot6d_to_rotmat(pose0).view(B, 24, 3, 3)
        pre_ = smpl(full_pose=R0, betas=shape0).vertices.clone()
        R2, b2, _ = S.pose_params(B, seed=77 + rank + 1000 * k, body_deg=8, global_deg=10)
        Rp = torch.matmul(torch.from_numpy(R2).to(dev), R0)
        jp = smpl(body_pose=Rp[:, 1:], global_orient=Rp[:, :1], betas=shape0 + torch.from_numpy(b2).to(dev) / 3,
                  return_verts=False).joints
        tgt_np_ = S.fit_targets_from_joints(jp.cpu().numpy(), cam0.cpu().numpy() * np.array([1.1, 1, 1], np.float32),
                                            seed=4321 + rank + 1000 * k)
        tgt_ = torch.from_numpy(tgt_np_[:, :, :2]).float().to(dev).contiguous()
        del x_dev_
        return SimpleNamespace(k=k, x_nchw=x_nchw_, joints_in=joints_in_, native=native_, tgt=tgt_, tgt_np=tgt_np_, pre_fit_vertices=pre_)

    inputs = [make_input(k) for k in range(n_inputs)]
    x_nchw, joints_in, native, tgt = inputs[0].x_nchw, inputs[0].joints_in, inputs[0].native, inputs[0].tgt
    torch.cuda.synchronize()

    if args.from_detections:
        from soccerplayershapepose_amd.predict.predict_3D import create_proxy_representation_batch
        # the detections of EVERY rotating input batch (silhouettes 512 x 512 u8 + joints): step k builds batch k's proxy, so that the
        # crops a step's result belongs to are the ones its fit targets and the parity check's oracle belong to
        dets = []
        for d in inputs:
            sil_k = torch.from_numpy(np.repeat(np.repeat(d.x_nchw[:, 0], 2, axis=1), 2, axis=2).astype(np.uint8)).to(dev)
            joints_k = torch.from_numpy(d.joints_in.astype(np.float32)).to(dev)
            assert torch.equal(create_proxy_representation_batch(sil_k, joints_k, precision=args.precision).view(torch.int16),
                               d.native.view(torch.int16))
            dets.append((sil_k, joints_k))

    def make_step(pipe_, natives_, precision):
        state = SimpleNamespace(i=0)

        def step():
            # one batch through the WHOLE path on the next pipeline slot (its own stream): [proxy builder ->] regressor ->
            # rot6d -> 50-iter fit -> SMPL forward [-> all-gather of the records]; the input batch rotates
            k = state.i % len(natives_)
            state.i += 1
            inp = create_proxy_representation_batch(dets[k][0], dets[k][1], precision=precision) if args.from_detections else natives_[k]
            # (N > 1: the record is read by the exchange on its own stream -- two alternating sets of result buffers per slot, so that
            # a slot's next batch does not wait for the all-gather of its previous one)
            bset = ((state.i - 1) // len(pipe_.slots)) & 1 if use_dist and lanes == 1 else 0
            res = pipe_.submit(inp, inputs[k].tgt, buffer_set=bset, fit=args.fit, iters=FIT_ITERS, return_record=use_dist)
            res.input_index = k
            if use_dist:
                with torch.cuda.stream(res.stream):
                    if lanes > 1:
                        # opt-in: the slot's stream has its own communicator, the collective is the last thing in the slot's order
                        res.gathered = exchange.all_gather(res.record, world * B, lane=lane_of.get(id(res.stream)))
                    else:
                        # ONE communicator on the exchange's stream: every rank issues its collectives in the same (program) order;
                        # the slot's next batch waits until this one's record has been read
                        res.gathered, ev = exchange.all_gather(res.record, world * B, wait=False)
                        res.hold_until(ev)
            return res
        step.state = state
        return step

    step = make_step(pipe, [d.native for d in inputs], args.precision)

    def barrier():
        if use_dist:
            dist.barrier()

    if args.profile_pass:                             # (the trace should hold launches of the one plan only)
        for sl in pipe.slots:
            sl.regressor.set_plan(args.profile_pass)
    for _ in range(max(1, len(pipe.slots))):          # first use of every slot (one-time kernel attribute set-up); the --warmup steps
        step()                                        # proper run right in front of the timed region, inside timed()
    torch.cuda.synchronize()

    def read_profile(nsteps):
        nk = r.lib.spsp_profile_num_kernels()
        ms = (C.c_double * nk)()
        cnt = (C.c_int * nk)()
        wgs = (C.c_double * nk)()
        r.check(r.lib.spsp_profile_read_grids(r.ctx, nk, ms, cnt, wgs), 'spsp_profile_read_grids')
        out = {}
        for i in range(nk):
            if cnt[i]:
                out[r.lib.spsp_profile_kernel_name(i).decode()] = {'id': i, 'launches_per_step': cnt[i] / nsteps,
                                                                   'avg_us': 1e3 * ms[i] / cnt[i],
                                                                   'us_per_step': 1e3 * ms[i] / nsteps,
                                                                   'avg_workgroups': wgs[i] / cnt[i]}
        return out

    # ---- untimed per-kernel pass: every kernel bracketed by a HIP-event pair on its launch stream, ONE batch in
    # flight (slot 0, a sync after every step), so a launch's duration is free of the cross-batch sharing the
    # pipelined timed region has by design (and agrees with the rocprofv3 average of the same command)
    # -- in plan "latency", the plan FOR one batch in flight (every grid fills the chip): a launch's duration against the chip's
    # peak says how good the kernel is.  The pipeline's slots run plan "throughput" (narrow persistent grids: layer1 on a quarter
    # of the CUs ...), whose launches take longer alone by design; their isolated durations are reported beside
    # (kernels_isolated_throughput_plan_*), the pipelined region itself by roofline.end_to_end_backbone_frac.
    iso_steps = 2 * n_inputs

    def isolated_pass(plan):
        reg.set_plan(plan)
        pipe._next = 0
        step()
        torch.cuda.synchronize()
        r.check(r.lib.spsp_profile_select(r.ctx, -1), 'spsp_profile_select')
        r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'spsp_profile_enable')
        for _ in range(iso_steps):
            pipe._next = 0
            step()
            torch.cuda.synchronize()
        r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'spsp_profile_enable')
        return read_profile(iso_steps)
    pipe_plan = 'throughput' if len(pipe.slots) >= 2 else 'latency'      # (what PlayerPipeline selected for its slots)
    if args.profile_pass:
        # the same launches, without event pairs, for rocprofv3's kernel trace (tools/gpu_profile.sh)
        reg.set_plan(args.profile_pass)
        for _ in range(4 * iso_steps):
            pipe._next = 0
            step()
            torch.cuda.synchronize()
        return          # (a normal exit: the profiler writes its trace from an exit handler)
    kernels_iso_tp = isolated_pass('throughput')
    kernels_iso = isolated_pass('latency')
    reg.set_plan(pipe_plan)
    pipe._next = 0
    # dominant kernel = the MFMA conv kernel with the largest time per step among those whose launches all do the same
    # algorithmic work (the library names it: spsp_profile_kernel_flop_per_crop > 0)
    cands = {}
    for k, v in kernels_iso.items():
        fl = r.lib.spsp_profile_kernel_flop_per_crop(v['id'])
        if fl > 0:
            cands[k] = fl * B
    cands_tp = {k: r.lib.spsp_profile_kernel_flop_per_crop(v['id']) * B for k, v in kernels_iso_tp.items()
                if r.lib.spsp_profile_kernel_flop_per_crop(v['id']) > 0 and v['avg_workgroups'] > 0}
    num_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    # ... IN THE PLAN THE TIMED REGION RUNS (round 5; rounds 1-4 took it from the latency plan whatever was timed)
    iso_sel, cands_sel = (kernels_iso_tp, cands_tp) if pipe_plan == 'throughput' else (kernels_iso, cands)
    dom = max(cands_sel, key=lambda k: iso_sel[k]['us_per_step'])

    # ---- timed region: EXACTLY --steps steps, barrier + synchronize on both sides.  Only the dominant kernel keeps
    # its event pair here: a pair is a barrier on its in-order stream, and bracketing all ~60 launches of a step costs
    # ~15 % throughput (measured), which would make `value` a profile-mode number.
    def timed(step_fn, pipe_, select_id):
        if args.graphs:
            # each slot's step is captured into a HIP graph (untimed) and replayed; events are not recorded inside graphs
            pipe_.use_graphs = True
            import math
            for _ in range(math.lcm(len(pipe_.slots), n_inputs) * (2 if use_dist else 1)):      # every (slot, input batch[, buffer set]) that will occur
                step_fn()
            torch.cuda.synchronize()
        # untimed: the W warm-up steps of the contract -- through the very path that is timed, with nothing but the barrier +
        # synchronize between them and the timed region
        for _ in range(args.warmup):
            step_fn()
        if not args.graphs and select_id is not None:
            torch.cuda.synchronize()
            r.check(r.lib.spsp_profile_select(r.ctx, select_id), 'spsp_profile_select')
            r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'spsp_profile_enable')
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out_ = step_fn()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'spsp_profile_enable')
        r.check(r.lib.spsp_profile_select(r.ctx, -1), 'spsp_profile_select')
        el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return el.item(), out_

    elapsed, out = timed(step, pipe, iso_sel[dom]['id'])
    if args.probe and args.graphs:
        for label, fn in (('step()', step), ('submit(natives[i % 4], one target set)', None)):
            rates = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for i_ in range(args.steps):
                    if fn is not None:
                        fn()
                    else:
                        pipe.submit(inputs[i_ % n_inputs].native, inputs[i_ % n_inputs].tgt, fit=args.fit, iters=FIT_ITERS, return_record=use_dist)
                torch.cuda.synchronize()
                rates.append(round(B * args.steps / (time.perf_counter() - t0_)))
            print(f'[probe] {label}: {rates} players/s (timed region: {B * args.steps / elapsed:.0f})', file=sys.stderr)
    if use_dist:        # every rank holds every player's record, in shard order
        assert out.gathered.shape == (world * B, 229) and torch.equal(out.gathered[rank * B:(rank + 1) * B, :3], out.cam)
    assert torch.isfinite(out.vertices).all() and out.vertices.abs().max().item() > 0.1
    kernels = read_profile(args.steps)
    post_fit_vertices = out.vertices.clone()
    parity_input = inputs[out.input_index]
    if args.graphs:
        # Events cannot be read back from inside a replayed graph, so the timed region above carries no event pair.  The same
        # pipeline is therefore run a little longer EAGERLY (same slots, same streams, four batches in flight; not part of
        # `value`) with the dominant kernel's scope enabled: its duration while it shares the chip with the other slots'
        # kernels, which is what `avg_us_timed_region` meant in the eager default of round 1.
        pipe.use_graphs = False
        nprobe = min(args.steps, 48)
        for _ in range(len(pipe.slots)):
            step()
        torch.cuda.synchronize()
        r.check(r.lib.spsp_profile_select(r.ctx, iso_sel[dom]['id']), 'spsp_profile_select')
        r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'spsp_profile_enable')
        for _ in range(nprobe):
            step()
        torch.cuda.synchronize()
        r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'spsp_profile_enable')
        r.check(r.lib.spsp_profile_select(r.ctx, -1), 'spsp_profile_select')
        kernels = read_profile(nprobe)
        pipe.use_graphs = True

    # ---- BASELINE.json configs[3]: latency of ONE 22-player frame, nothing else in flight (players shard over the ranks
    # 3,3,3,...; the frame is done when every rank holds every record)
    flo, fhi = shard_range(FRAME_PLAYERS, world, rank)
    fn = fhi - flo
    reg.set_plan('latency')        # one batch in flight: layer4 on the whole chip (include/spsp.h, SPSP_PLAN_LATENCY)
    pipe.use_graphs = bool(args.graphs)
    lat = []
    for i in range(25):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fn > 0:
            # through slot 0 of the pipeline: with HIP graphs (the default) the frame's ~35 launches are ONE graph replay (captured
            # in the first of the 5 untimed repetitions, with the latency plan) -- the launch gaps of the eager chain are ~0.07 ms
            if args.graphs:
                pipe._next = 0
                fres = pipe.submit(native[:fn], tgt[:fn], fit=args.fit, iters=FIT_ITERS, return_record=use_dist)
            else:
                fres = reconstruct_players(reg, smpl, native[:fn], tgt[:fn], fit=args.fit, iters=FIT_ITERS, return_record=use_dist)
        if use_dist:
            with torch.cuda.stream(pipe.slots[0].stream if args.graphs else torch.cuda.current_stream(dev)):
                rec = fres.record if fn > 0 else torch.zeros((0, 229), dtype=torch.float32, device=dev)
                exchange.all_gather(rec, FRAME_PLAYERS)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    reg.set_plan(pipe_plan)
    pipe._next = 0
    lat_t = torch.tensor(sorted(lat[5:])[len(lat[5:]) // 2], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(lat_t, op=dist.ReduceOp.MAX)
    frame_latency_ms = 1e3 * lat_t.item()

    # ---- the other element type, same steps: a second throughput figure (not `value`)
    alt = None
    if not args.no_alt_precision:
        alt_prec = 'bf16' if args.precision == 'f16' else 'f16'
        del fres
        torch.cuda.synchronize()
        # on the first pipeline's (now idle) streams: four more streams would land on further hardware queues, and past
        # four active queues the GPU time-slices them (the second figure then read 25 % low)
        pipe2 = make_pipe(alt_prec, streams=[sl.stream for sl in pipe.slots])
        reg2 = pipe2.slots[0].regressor
        natives2 = [reg2.pack(torch.from_numpy(d.x_nchw).to(dev)) for d in inputs]
        c2, p2, s2 = reg2(natives2[0])
        pre2 = smpl(full_pose=rot6d_to_rotmat(p2).view(B, 24, 3, 3), betas=s2).vertices.clone()
        step2 = make_step(pipe2, natives2, alt_prec)
        for _ in range(len(pipe2.slots)):
            step2()
        torch.cuda.synchronize()
        el2, out2 = timed(step2, pipe2, None)
        alt = {'dtype': alt_prec, 'value': world * B * args.steps / el2, 'ms_per_step': 1e3 * el2 / args.steps,
               'pre_fit_vertices': pre2, 'post_fit_vertices': out2.vertices.clone()}

    # ---- the same steps with ONE tail per `--tail-group` consecutive batches of a slot (a third figure, not `value`): every batch
    # still runs the backbone's batch-32 launches; IEF -> fit -> SMPL run once over the group's players (latency-bound chains of
    # small launches: as long for 64 players as for 32), so a slot spends half as long per batch with no convolution in flight
    grouped = None
    G = max(1, args.tail_group)
    if G > 1 and not use_dist and not args.from_detections and args.steps % G == 0:
        if 'fres' in dir():
            del fres
        torch.cuda.synchronize()
        mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=G * B, precision=args.precision).load_state_dict(sd)
        mk_s = lambda: SMPL(model, max_batch=G * B)
        pipe3 = PlayerPipeline(mk_r, mk_s, dev, depth=max(1, args.depth), streams=[sl.stream for sl in pipe.slots])
        gstate = SimpleNamespace(i=0)

        def gstep():
            ks = [(gstate.i * G + j) % n_inputs for j in range(G)]
            gstate.i += 1
            return ks, pipe3.submit_group([inputs[k].native for k in ks], [inputs[k].tgt for k in ks], fit=args.fit, iters=FIT_ITERS)
        ks0, r0 = gstep()
        torch.cuda.synchronize()
        # every player of the group bit-identical to the per-batch path
        same = True
        for k, one in zip(ks0, r0):
            ref_ = reconstruct_players(reg, smpl, inputs[k].native, inputs[k].tgt, fit=args.fit, iters=FIT_ITERS)
            torch.cuda.synchronize()
            same = same and all(torch.equal(getattr(one, f), getattr(ref_, f)) for f in ('cam', 'rotmats', 'betas', 'vertices', 'joints', 'translation'))
        if args.graphs:
            import math
            pipe3.use_graphs = True
            for _ in range(math.lcm(len(pipe3.slots), max(1, n_inputs // math.gcd(n_inputs, G)))):
                gstep()
            torch.cuda.synchronize()
        for _ in range(max(1, args.warmup // G)):
            gstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps // G):
            gstep()
        torch.cuda.synchronize()
        el3 = time.perf_counter() - t0
        grouped = {'batches_per_tail': G, 'value': world * B * args.steps / el3, 'unit': 'players/s', 'ms_per_step': 1e3 * el3 / args.steps,
                   'bit_identical_to_per_batch_path': bool(same),
                   'what': f'the same {args.steps} steps of {B} crops; the backbone runs batch by batch, IEF -> fit -> SMPL once per {G} consecutive '
                           f'batches of a slot (PlayerPipeline.submit_group / spsp_pipeline_group)'}
        del pipe3

    if rank == 0:
        achieved = cands_sel[dom] / (kernels[dom]['avg_us'] * 1e-6) / 1e12 if dom in kernels else None
        achieved_iso = cands_sel[dom] / (iso_sel[dom]['avg_us'] * 1e-6) / 1e12
        cu_share = min(1.0, iso_sel[dom]['avg_workgroups'] / num_cus) if iso_sel[dom]['avg_workgroups'] > 0 else 1.0
        conv_us = sum(v['us_per_step'] for k, v in kernels_iso.items() if k.startswith('conv'))
        pmc_path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        traffic = None
        if os.path.exists(pmc_path):
            traffic = json.load(open(pmc_path)).get(dom)
        line = {
            'metric': 'players/sec end-to-end (backbone+SMPL+fit) on 256x256 crops',
            'value': world * B * args.steps / elapsed, 'unit': 'players/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': f'batch-{B} 256x256 synthetic proxy crops per GPU: ResNet-18+IEF regressor ({args.precision} storage and MFMA, '
                                   f'fp32 accumulate) -> rot6d -> {FIT_ITERS}-iter {args.fit} Adam fit -> SMPL forward '
                                   f'(6890 verts, 90 joints)' + (' -> RCCL all-gather of [cam|R|betas]' if use_dist else ''),
                       'crops_per_gpu': B, 'global_batch': world * B, 'fit_iters': FIT_ITERS, 'fit_variant': args.fit,
                       'parallelism': f'dp{world}', 'input_layout': f'NHWC {args.precision} [262,264,18] resident in HBM',
                       'pipeline_slots': max(1, args.depth), 'hip_graphs': bool(args.graphs), 'input_batches_rotated': n_inputs,
                       'regressor_plan': reg.plan + (' + hooks ' + args.plan_hooks if args.plan_hooks else ''),
                       'proxy_built_on_gpu': bool(args.from_detections),
                       'exchange': (exchange.describe() if exchange else None)},
            # The dominant kernel = the MFMA kernel with the most time per step IN THE PLAN THE TIMED REGION RAN (`plan`); achieved =
            # algorithmic FLOPs of one launch / its average duration (the kernel's own begin / end timestamps, ONE batch in flight, that
            # plan's grid) -- against the WHOLE chip's peak, so a narrow launch of plan "throughput" reads low by design: `workgroups`
            # and `frac_of_cus_used` price it against the CUs it occupies; `end_to_end_backbone_frac` is what the timed region as a
            # whole achieves.  In the pipelined timed region the same kernel's event pair also spans the time its workgroups share CUs
            # with other batches' kernels: *_timed_region.  `latency_plan`: the chip-filling launches of the same kernels.
            'roofline': {'bound': 'mfma', 'kernel': dom, 'plan': pipe_plan, 'achieved': achieved_iso, 'peak': MFMA_PEAK_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': achieved_iso / MFMA_PEAK_TFLOPS,
                         'workgroups': iso_sel[dom]['avg_workgroups'], 'frac_of_cus_used': achieved_iso / (MFMA_PEAK_TFLOPS * cu_share),
                         'traffic': traffic,
                         'avg_us': iso_sel[dom]['avg_us'],
                         'flop_per_launch': cands_sel[dom],
                         # (with HIP graphs: from the eager continuation of the timed pipeline, see above)
                         'avg_us_timed_region': kernels.get(dom, {}).get('avg_us'),
                         'achieved_timed_region': achieved,
                         'all_convs_tflops': (FLOP_BACKBONE * B / (conv_us * 1e-6) / 1e12) if conv_us else None,
                         'all_convs_frac': (FLOP_BACKBONE * B / (conv_us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS) if conv_us else None,
                         # every kernel of the table that does the same work in each launch: isolated duration -> fraction of the peak
                         'latency_plan': {'what': 'the same kernels launched chip-filling (a handle\'s default plan; one batch in flight)',
                                          'mfma_kernels': {k: {'launches_per_step': kernels_iso[k]['launches_per_step'], 'avg_us': round(kernels_iso[k]['avg_us'], 2),
                                                               'frac': round(f / (kernels_iso[k]['avg_us'] * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
                                                           for k, f in cands.items() if k in kernels_iso}},
                         'end_to_end_backbone_tflops': FLOP_BACKBONE * world * B * args.steps / elapsed / 1e12,
                         'end_to_end_backbone_frac': FLOP_BACKBONE * B * args.steps / elapsed / 1e12 / MFMA_PEAK_TFLOPS},
            'latency': {'workload': f'one {FRAME_PLAYERS}-player match frame (BASELINE configs[3]) through the whole path, nothing else in '
                                    f'flight (regressor plan "latency"' + (', one HIP-graph replay' if args.graphs else '') + f'), {world} GPU(s)' + (', players sharded 3,3,... + all-gather' if world > 1 else ''),
                        'ms': frame_latency_ms},
            'kernels_isolated_plan': 'latency (one batch in flight: chip-filling grids)',
            'kernels_isolated_us_per_step': {k: round(v['us_per_step'], 2) for k, v in kernels_iso.items()},
            'kernels_isolated_avg_us': {k: round(v['avg_us'], 2) for k, v in kernels_iso.items()},
            'kernels_isolated_throughput_plan_us_per_step': {k: round(v['us_per_step'], 2) for k, v in kernels_iso_tp.items()},
            # the MFMA kernels of plan "throughput" against the peak of the CUs their launches occupy (one workgroup per CU: 150 KB of
            # LDS each; a launch of 64 workgroups cannot use more than a quarter of the chip and is not meant to)
            'mfma_kernels_throughput_plan': {k: {'launches_per_step': kernels_iso_tp[k]['launches_per_step'], 'avg_us': round(kernels_iso_tp[k]['avg_us'], 2),
                                                 'avg_workgroups': round(kernels_iso_tp[k]['avg_workgroups'], 1),
                                                 'frac_of_chip': round(f / (kernels_iso_tp[k]['avg_us'] * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                                 'frac_of_cus_used': round(f / (kernels_iso_tp[k]['avg_us'] * 1e-6) / 1e12 / (MFMA_PEAK_TFLOPS * min(1.0, kernels_iso_tp[k]['avg_workgroups'] / num_cus)), 4)}
                                             for k, f in cands_tp.items()},
        }
        # cross-check against the TRACKED rocprofv3 kernel trace of the isolated command (same build): the live figure must be
        # reproducible from a file under profiles/
        rp = rocprof_average_us(dom, pipe_plan) if args.precision == 'f16' and B == BATCH_PER_GPU else None
        if rp is not None:
            line['roofline']['rocprof'] = {'source': rocprof_isolated_path(pipe_plan) + f' (rocprofv3 --kernel-trace --stats -- python3 bench.py --profile-pass {pipe_plan})',
                                           'avg_us': rp, 'frac': cands_sel[dom] / (rp * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS,
                                           'live_over_rocprof': iso_sel[dom]['avg_us'] / rp}
        if alt is not None:
            line['alt_precision'] = {'dtype': alt['dtype'], 'value': alt['value'], 'unit': 'players/s', 'ms_per_step': alt['ms_per_step']}
        if grouped is not None:
            line['grouped_tail'] = grouped
        if world == 1 and not args.no_cpu_baseline:
            threads = host_threads(args.cpu_threads)
            base, parity = cpu_baseline(sd, model, torch.from_numpy(parity_input.x_nchw), parity_input.tgt_np, args.cpu_fit_sample, threads,
                                        parity_input.pre_fit_vertices, post_fit_vertices)
            line['cpu_baseline'] = base
            # the north-star tolerance (vertices within 1e-3 m of the CPU reference path) is a pass / fail property of the line
            parity['ok'] = bool(parity['pre_fit_max_m'] <= parity['tolerance_m'] and parity['post_fit_max_m'] <= parity['tolerance_m'])
            line['parity'] = dict(parity, dtype=args.precision, input_batch=int(parity_input.k))
            if alt is not None:
                _, parity2 = cpu_baseline_parity_only(sd, model, torch.from_numpy(x_nchw), alt['pre_fit_vertices'])
                line['alt_precision']['parity'] = parity2
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + '\n').encode())
        parity_failed = 'parity' in line and not line['parity']['ok']
    else:
        parity_failed = False
    if use_dist:
        dist.barrier()
        if exchange:
            exchange.close()
        dist.destroy_process_group()
    if parity_failed:
        sys.stderr.write(f'bench.py: PARITY FAILED -- vertices deviate from the all-fp32 CPU reference path by more than '
                         f'{line["parity"]["tolerance_m"]} m ({line["parity"]}); the line above is not a valid result\n')
        raise SystemExit(3)


def cpu_baseline_parity_only(sd, model, x_nchw, gpu_pre):
    """Pre-fit vertex deviation of a second element type against the all-fp32 oracle (no timing)."""
    from soccerplayershapepose_amd import synthetic as S
    from oracle import regressor as OR, geometry as G, smpl as OS
    osmpl = OS.OracleSMPL(model)
    cam, pose, shape = OR.regressor_forward(sd, x_nchw, S.mean_params())
    R = G.rot6d_to_rotmat(pose).view(x_nchw.shape[0], 24, 3, 3)
    with torch.no_grad():
        pre = osmpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=shape).vertices
    return None, {'pre_fit_players': int(x_nchw.shape[0]), 'pre_fit_max_m': float((gpu_pre.cpu() - pre).abs().max())}


if __name__ == '__main__':
    main()
