#!/usr/bin/env python3
"""bench.py -- players/sec end-to-end (backbone -> rot6d -> 50-iter pose+shape fit -> SMPL forward -> vertices)
on synthetic 256x256 proxy crops, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of 32 crops per GPU (BASELINE.json configs[1]), inputs
already resident in HBM in the backbone-native layout.  With N > 1 every rank processes its own 32 crops (weak
scaling, crops shard data-parallel with no data-path collective) and the step ends with ONE RCCL all-gather of the
packed per-player record [cam | R | betas] = 229 fp32.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One hardware queue per pipeline slot: HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues
# round-robin, and two slots sharing a queue serialise their whole chains.  Must be set before HIP initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BATCH_PER_GPU = 32
FIT_ITERS = 50
# algorithmic work per player (SURVEY 8d / BASELINE.md section 4)
FLOP_CONV1 = 2 * 924_844_032            # conv1 7x7/2, 18 -> 64, real (unpadded) MACs
FLOP_L1CONV = 2 * 150_994_944           # one 3x3 64->64 conv at 64x64
FLOP_BACKBONE = 2 * 3_139_436_544
MFMA_BF16_PEAK_TFLOPS = 2500.0          # dense bf16, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='crops per GPU per step')
    ap.add_argument('--fit', default='pose_shape', choices=['pose_shape', 'camera'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--depth', type=int, default=4, help='pipeline slots (HIP streams); 1 = strictly serial steps')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'f16'], help='backbone storage / MFMA element type')
    ap.add_argument('--from-detections', action='store_true',
                    help='each step also builds the native proxy input on the GPU from silhouettes [B,512,512] u8 + joints2D (SURVEY 8f-1)')
    ap.add_argument('--rehearse-dist', action='store_true',
                    help='run the N>1 code path (RCCL process group, per-step all-gather, barriers) even with one rank')
    ap.add_argument('--graphs', action='store_true', help='replay each slot\'s step from a captured HIP graph (no in-region kernel events)')
    ap.add_argument('--cpu-sample', type=int, default=64, help='players timed through the CPU oracle (64: ~12 s on 16 threads)')
    ap.add_argument('--cpu-threads', type=int, default=16, help="host threads for the CPU oracle (the GPU box's CPU share)")
    return ap.parse_args()


def cpu_baseline(sd, model, x_nchw, tgt, n, threads):
    """The oracle (restated reference path, PyTorch-CPU fp32, full SMPL.forward + autograd + Adam per fit
    iteration) timed on this box's host cores over a bounded sample.  Checker timed as a BASELINE only."""
    from soccerplayershapepose_amd import synthetic as S
    from oracle import regressor as OR, geometry as G, fit as OF, smpl as OS
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    threads = torch.get_num_threads()
    osmpl = OS.OracleSMPL(model)
    idx = [i % x_nchw.shape[0] for i in range(n)]           # the batch's players, cycled if the sample is larger
    t0 = time.perf_counter()
    parts = [OR.regressor_forward(sd, x_nchw[idx[i:i + 32]], S.mean_params()) for i in range(0, n, 32)]
    cam, pose, shape = (torch.cat([p[k] for p in parts]) for k in range(3))
    R = G.rot6d_to_rotmat(pose).view(n, 24, 3, 3)
    t1 = time.perf_counter()
    for i in range(n):
        r = OF.fit_pose_shape(osmpl, R[i:i + 1, 1:], R[i:i + 1, :1], shape[i:i + 1], tgt[idx[i]], cam[i:i + 1], iters=FIT_ITERS)
        body, Gm, be, ca, _ = r['last']
        with torch.no_grad():
            osmpl(body_pose=body, global_orient=Gm, betas=be)
    t2 = time.perf_counter()
    return {'value': n / (t2 - t0), 'unit': 'players/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n} players end-to-end (regressor {t1 - t0:.2f}s + {FIT_ITERS}-iter fit & SMPL {t2 - t1:.2f}s), '
                      f'torch-CPU fp32 oracle, {threads} threads'}


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON result: native libraries write banners there (RCCL prints its version
    # block to stdout at communicator creation), so fd 1 is pointed at stderr for the duration of the run and the
    # result line goes to a saved copy of the original stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
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
    from soccerplayershapepose_amd.player_recon import PlayerPipeline, pack_record
    from soccerplayershapepose_amd.sharding import all_gather_records
    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat

    B = args.batch
    sd = S.regressor_state_dict(seed=0)
    model = S.smpl_model(seed=0)
    def make_regressor():
        m = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=args.precision)
        return m.load_state_dict(sd)

    def make_smpl():
        return SMPL(model, max_batch=B)

    pipe = PlayerPipeline(make_regressor, make_smpl, dev, depth=max(1, args.depth))
    reg, smpl = pipe.slots[0].regressor, pipe.slots[0].smpl
    r = rt.get(dev)

    # ---- synthetic inputs (seed 1234 + rank), resident in HBM in the backbone-native layout ----
    x_nchw, joints_in = S.proxy_crops(B, seed=1234 + rank)
    native = reg.pack(torch.from_numpy(x_nchw).to(dev))
    # fit targets: the regressor-initialised model, perturbed + 1 px noise; set-up, untimed
    cam0, pose0, shape0 = reg(native)
    R0 = rot6d_to_rotmat(pose0).view(B, 24, 3, 3)
    R2, b2, _ = S.pose_params(B, seed=77 + rank, body_deg=8, global_deg=10)
    Rp = torch.matmul(torch.from_numpy(R2).to(dev), R0)
    jp = smpl(body_pose=Rp[:, 1:], global_orient=Rp[:, :1], betas=shape0 + torch.from_numpy(b2).to(dev) / 3,
              return_verts=False).joints
    tgt_np = S.fit_targets_from_joints(jp.cpu().numpy(), cam0.cpu().numpy() * np.array([1.1, 1, 1], np.float32),
                                       seed=4321 + rank)
    tgt = torch.from_numpy(tgt_np[:, :, :2]).float().to(dev).contiguous()
    torch.cuda.synchronize()

    if args.from_detections:
        from soccerplayershapepose_amd.predict.predict_3D import create_proxy_representation_batch
        sil_dev = torch.from_numpy(np.repeat(np.repeat(x_nchw[:, 0], 2, axis=1), 2, axis=2).astype(np.uint8)).to(dev)
        joints_dev = torch.from_numpy(joints_in.astype(np.float32)).to(dev)
        assert torch.equal(create_proxy_representation_batch(sil_dev, joints_dev, precision=args.precision).view(torch.int16),
                           native.view(torch.int16))

    def step():
        # one batch through the WHOLE path on the next pipeline slot (its own stream): [proxy builder ->] regressor ->
        # rot6d -> 50-iter fit -> SMPL forward [-> all-gather of the records]
        inp = create_proxy_representation_batch(sil_dev, joints_dev, precision=args.precision) if args.from_detections else native
        res = pipe.submit(inp, tgt, fit=args.fit, iters=FIT_ITERS)
        if use_dist:
            with torch.cuda.stream(res.stream):
                res.gathered = all_gather_records(pack_record(res.cam, res.rotmats, res.betas), world * B,
                                                  force_collective=True)
        return res

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    def read_profile(nsteps):
        nk = r.lib.spsp_profile_num_kernels()
        ms = (C.c_double * nk)()
        cnt = (C.c_int * nk)()
        r.check(r.lib.spsp_profile_read(r.ctx, nk, ms, cnt), 'spsp_profile_read')
        out = {}
        for i in range(nk):
            if cnt[i]:
                out[r.lib.spsp_profile_kernel_name(i).decode()] = {'id': i, 'launches_per_step': cnt[i] / nsteps,
                                                                   'avg_us': 1e3 * ms[i] / cnt[i],
                                                                   'us_per_step': 1e3 * ms[i] / nsteps}
        return out

    # ---- untimed per-kernel pass: every kernel bracketed by a HIP-event pair on its launch stream, ONE batch in
    # flight (slot 0, a sync after every step), so a launch's duration is free of the cross-batch sharing the
    # pipelined timed region has by design (and agrees with the rocprofv3 average of the same command)
    iso_steps = 8
    r.check(r.lib.spsp_profile_select(r.ctx, -1), 'spsp_profile_select')
    r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'spsp_profile_enable')
    for _ in range(iso_steps):
        pipe._next = 0
        step()
        torch.cuda.synchronize()
    r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'spsp_profile_enable')
    kernels_iso = read_profile(iso_steps)
    pipe._next = 0
    # dominant kernel = the MFMA conv kernel with the largest time per step among those whose launches all have the
    # same shape: the stem (1 launch/step) and the 3x3 stride-1 kernels (every 3x3/1 conv of ResNet-18 has
    # 150 994 944 MACs per crop, whatever its layer)
    cands = {'conv1_half_kernel': FLOP_CONV1 * B, 'conv1_patch_kernel': FLOP_CONV1 * B,
             'conv_igemm_kernel<128, 64, 128, 8>': FLOP_CONV1 * B}
    for k in kernels_iso:
        if k.startswith('conv3x3_'):
            cands[k] = FLOP_L1CONV * B
    dom = max(cands, key=lambda k: kernels_iso.get(k, {'us_per_step': 0})['us_per_step'])

    # ---- timed region: EXACTLY --steps steps, barrier + synchronize on both sides.  Only the dominant kernel keeps
    # its event pair here: a pair is a barrier on its in-order stream, and bracketing all ~60 launches of a step costs
    # ~15 % throughput (measured), which would make `value` a profile-mode number.
    if args.graphs:
        # each slot's step is captured into a HIP graph (untimed) and replayed; events are not recorded inside graphs
        pipe.use_graphs = True
        for _ in range(len(pipe.slots)):
            step()
        torch.cuda.synchronize()
    else:
        r.check(r.lib.spsp_profile_select(r.ctx, kernels_iso[dom]['id']), 'spsp_profile_select')
        r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'spsp_profile_enable')
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'spsp_profile_enable')
    r.check(r.lib.spsp_profile_select(r.ctx, -1), 'spsp_profile_select')
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = elapsed.item()
    if use_dist:        # every rank holds every player's record, in shard order
        assert out.gathered.shape == (world * B, 229) and torch.equal(out.gathered[rank * B:(rank + 1) * B, :3], out.cam)
    assert torch.isfinite(out.vertices).all() and out.vertices.abs().max().item() > 0.1
    kernels = read_profile(args.steps)

    if rank == 0:
        achieved = cands[dom] / (kernels[dom]['avg_us'] * 1e-6) / 1e12 if dom in kernels else None
        achieved_iso = cands[dom] / (kernels_iso[dom]['avg_us'] * 1e-6) / 1e12 if dom in kernels_iso else None
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
            'config': {'workload': f'batch-{B} 256x256 synthetic proxy crops per GPU: ResNet-18+IEF regressor ({args.precision} MFMA, '
                                   f'fp32 accumulate) -> rot6d -> {FIT_ITERS}-iter {args.fit} Adam fit -> SMPL forward '
                                   f'(6890 verts, 90 joints)' + (' -> RCCL all-gather of [cam|R|betas]' if use_dist else ''),
                       'crops_per_gpu': B, 'global_batch': world * B, 'fit_iters': FIT_ITERS, 'fit_variant': args.fit,
                       'parallelism': f'dp{world}', 'input_layout': f'NHWC {args.precision} [262,264,18] resident in HBM',
                       'pipeline_slots': max(1, args.depth), 'hip_graphs': bool(args.graphs),
                       'proxy_built_on_gpu': bool(args.from_detections)},
            # achieved = algorithmic FLOPs of one launch / its average duration between HIP events on its stream with ONE
            # batch in flight (agrees with the rocprofv3 average).  In the pipelined timed region the same kernel's
            # event pair also spans the time its workgroups share CUs with other batches' kernels; that figure is
            # kept as *_timed_region.
            'roofline': {'bound': 'mfma', 'kernel': dom, 'achieved': achieved_iso, 'peak': MFMA_BF16_PEAK_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': (achieved_iso / MFMA_BF16_PEAK_TFLOPS) if achieved_iso else None,
                         'traffic': traffic,
                         'avg_us': kernels_iso.get(dom, {}).get('avg_us'),
                         'flop_per_launch': cands[dom],
                         'avg_us_timed_region': kernels.get(dom, {}).get('avg_us'),
                         'achieved_timed_region': achieved,
                         'all_convs_tflops': (FLOP_BACKBONE * B / (conv_us * 1e-6) / 1e12) if conv_us else None,
                         'end_to_end_backbone_tflops': FLOP_BACKBONE * B * args.steps / elapsed / 1e12},
            'kernels_isolated_us_per_step': {k: round(v['us_per_step'], 2) for k, v in kernels_iso.items()},
            'kernels_isolated_avg_us': {k: round(v['avg_us'], 2) for k, v in kernels_iso.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(sd, model, torch.from_numpy(x_nchw), tgt_np, args.cpu_sample, args.cpu_threads)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
