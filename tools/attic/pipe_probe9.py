"""Ninth probe (bisection between pipe_probe6, which reads 123 k, and the single-pipeline processes, which read 109 k on the same box).
modes: one | three (three pipelines built and warmed, only A timed) | three_env (as pipe_probe6: the second one captured with the
level sweeps) | three_nofit (the third one without the fit) | six"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd import runtime as rt
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import PlayerPipeline
mode = sys.argv[1]
B, steps = 32, 200
dev = torch.device('cuda', 0)
rt.get(dev)
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd)
mk_s = lambda: SMPL(model, max_batch=B)
smode = os.environ.get('PROBE_STREAMS', 'normal')
if smode == 'normal':
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
elif smode == 'mixed':
    streams = [torch.cuda.Stream(device=dev, priority=p) for p in (0, -1, 0, -1)]
elif smode == 'high':
    streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(4)]
elif smode == 'even_of_8':
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)][0::2]
elif smode == 'last_of_8':
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)][4:]
elif smode == 'own':
    streams = None
elif smode == 'skip_first':
    streams = [torch.cuda.Stream(device=dev) for _ in range(5)][1:]
x, _ = S.proxy_crops(B, seed=1234)
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
FIT = dict(fit='pose_shape', iters=50)
def build(kw):
    pipe = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True, streams=streams)
    nat = [pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev)) for _ in range(4)]
    for i in range(8):
        pipe.submit(nat[i % 4], tgt, **kw)
    torch.cuda.synchronize()
    return pipe, nat, kw
n = {'one': 1, 'three': 3, 'three_env': 3, 'three_nofit': 3, 'six': 6, 'one_torchop': 1, 'one_sidestream': 1, 'one_interleave_torchop': 1}[mode]
pipes = []
for k in range(n):
    if mode == 'three_env' and k == 1:
        os.environ['SPSP_FIT_LEVEL_SWEEPS'] = '1'
    pipes.append(build(dict(fit=None) if (mode == 'three_nofit' and k == 2) else FIT))
    os.environ.pop('SPSP_FIT_LEVEL_SWEEPS', None)
pipe, nat, kw = pipes[0]
v = []
if mode == 'one_torchop':       # a PyTorch kernel on the default stream before the timed loops (what pipe_probe6 / ab_pipeline do)
    out = pipe.submit(nat[0], tgt, **kw)
    torch.cuda.synchronize()
    print('finite:', bool(torch.isfinite(out.vertices).all()), float(out.vertices.abs().max()))
import contextlib
recipe = os.environ.get('PROBE_RECIPE', '')
if recipe == 'R1':
    torch.cuda.synchronize(); keep = torch.zeros(64, device=dev); torch.cuda.synchronize()
elif recipe == 'R2':
    keep = torch.zeros(64, device=dev)
elif recipe == 'R3':
    torch.cuda.synchronize(); keep = torch.zeros(64, device=dev)
elif recipe == 'R4':
    keep = torch.zeros(64, device=dev); torch.cuda.synchronize()
elif recipe == 'R6':        # first launch of a kernel from ANOTHER translation unit of libspsp (csrc/proxy.hip): a code object load, no torch kernel
    from soccerplayershapepose_amd.predict.predict_3D import create_proxy_representation_batch
    keep = create_proxy_representation_batch(torch.from_numpy(np.zeros((2, 512, 512), np.uint8)).to(dev), torch.from_numpy(np.full((2, 17, 2), 200, np.float32)).to(dev))
elif recipe == 'R7':        # a 512 MB allocation, no kernel
    keep = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device=dev)
elif recipe == 'R8':        # an already-loaded libspsp kernel on the default stream
    keep = pipe.slots[0].regressor.pack(torch.from_numpy(x[:1]).to(dev))
elif recipe == 'R5':        # the big clone of pipe_probe6
    keep = pipe.submit(nat[0], tgt, **kw).vertices.clone()
side = torch.cuda.Stream(device=dev) if mode == 'one_sidestream' else None
for rd in range(4):
    if mode == 'one_interleave_torchop' and rd == 2:
        print('finite:', bool(torch.isfinite(pipe.submit(nat[0], tgt, **kw).wait().vertices).all()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        for i in range(steps):
            pipe.submit(nat[i % 4], tgt, **kw)
    th = time.perf_counter()
    torch.cuda.synchronize()
    v.append((round(B * steps / (time.perf_counter() - t0)), round(1e6 * (th - t0) / steps)))
print(f'{mode:12s} A: {v} (players/s, host us per submit)', flush=True)
