"""Seventh probe: the latency-bound tail (rot6d -> fit -> SMPL forward) of a slot on a stream of its own, so that the slot's
convolution stream starts the next batch's backbone while the tail still runs (two HIP graphs per slot and buffer set, joined by
events; the regressor -> tail hand-off double-buffered).
usage: python tools/pipe_probe7.py <tail streams: 0 = tail on the convolution stream (today's pipeline) | 1 | 2 | 4> [fit plan]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S, config, _abi
from soccerplayershapepose_amd import runtime as rt
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import _fit_config
NT = int(sys.argv[1])
plan = sys.argv[2] if len(sys.argv) > 2 else 'latency'
B, steps, depth = 32, 200, 4
dev = torch.device('cuda', 0)
r = rt.get(dev)
lib = r.lib
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
conv_streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
tail_streams = [torch.cuda.Stream(device=dev) for _ in range(NT)]
x, _ = S.proxy_crops(B, seed=1234)
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev).contiguous()
f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
cfg = _fit_config(r, 1, 50, None, None, _abi.FIT_SELECT_LAST)
slots = []
for i in range(depth):
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd).to(dev)
    smpl = SMPL(model, max_batch=B).to(dev)
    nat = reg.pack(torch.from_numpy(x).to(dev))
    sets = [dict(cam=f32(B, 3), pose6d=f32(B, 144), betas=f32(B, 10), R=f32(B, 24, 3, 3), v=f32(B, 6890, 3), j=f32(B, 90, 3), t=f32(B, 3)) for _ in range(2)]
    slots.append(dict(reg=reg, smpl=smpl, nat=nat, sets=sets, A=conv_streams[i], T=tail_streams[i % NT] if NT else conv_streams[i],
                      g_bb=[None, None], g_tail=[None, None], ev_bb=torch.cuda.Event(), ev_tail=[None, None]))
torch.cuda.synchronize()


def backbone(sl, o):
    r.check(lib.spsp_regressor_forward(r.ctx, sl['reg'].handle, rt.ptr(sl['nat']), _abi.PROXY_NATIVE, B, rt.ptr(o['cam']), rt.ptr(o['pose6d']),
                                       rt.ptr(o['betas']), None, r.stream()), 'regressor')


def tail(sl, o):
    r.check(lib.spsp_rot6d_to_rotmat(r.ctx, rt.ptr(o['pose6d']), B * 24, rt.ptr(o['R']), r.stream()), 'rot6d')
    r.check(lib.spsp_fit(r.ctx, sl['smpl'].handle, C.byref(cfg), rt.ptr(o['R']), rt.ptr(o['betas']), rt.ptr(o['cam']), rt.ptr(tgt), B, None, None,
                         r.stream()), 'fit')
    r.check(lib.spsp_smpl_forward(r.ctx, sl['smpl'].handle, rt.ptr(o['R']), rt.ptr(o['betas']), B, rt.ptr(o['v']), rt.ptr(o['j']), r.stream()), 'smpl')
    r.check(lib.spsp_wp_to_translation(r.ctx, rt.ptr(o['cam']), B, config.FOCAL_LENGTH, config.proxy_rep_input_wh, rt.ptr(o['t']), r.stream()), 'wp')


for sl in slots:                # warm-up (eager) and capture
    for p in range(2):
        o = sl['sets'][p]
        with torch.cuda.stream(sl['A']):
            backbone(sl, o)
            sl['A'].synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=sl['A']):
                backbone(sl, o)
            sl['g_bb'][p] = g
        with torch.cuda.stream(sl['T']):
            tail(sl, o)
            sl['T'].synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=sl['T']):
                tail(sl, o)
            sl['g_tail'][p] = g
torch.cuda.synchronize()
ref = slots[0]['sets'][0]['v'].clone()


def submit(n):
    sl = slots[n % depth]
    p = (n // depth) & 1
    A, T = sl['A'], sl['T']
    with torch.cuda.stream(A):
        if sl['ev_tail'][p] is not None:
            A.wait_event(sl['ev_tail'][p])           # the tail two batches back on this slot read this buffer set
        sl['g_bb'][p].replay()
        sl['ev_bb'].record(A)
    with torch.cuda.stream(T):
        if T is not A:
            T.wait_event(sl['ev_bb'])
        sl['g_tail'][p].replay()
        ev = torch.cuda.Event()
        ev.record(T)
        sl['ev_tail'][p] = ev


for n in range(16):
    submit(n)
torch.cuda.synchronize()
v = []
for rd in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for n in range(steps):
        submit(n)
    torch.cuda.synchronize()
    v.append(round(B * steps / (time.perf_counter() - t0)))
same = all(bool(torch.equal(sl['sets'][p]['v'], ref)) for sl in slots for p in range(2))
print(f'tail streams {NT} (HW queues {os.environ["GPU_MAX_HW_QUEUES"]}), fit plan {plan}: {v} players/s; results identical across slots and sets: {same}')
