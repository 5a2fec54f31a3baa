"""Sixth probe: what the reprojection fit costs the pipelined throughput mode -- the fit with its tree walks, with the level sweeps
they replaced (SPSP_FIT_LEVEL_SWEEPS, read at every launch: a captured graph keeps the kernel it was captured with), and no fit at
all; interleaved rounds on the same four streams."""
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
B, steps = 32, 200
dev = torch.device('cuda', 0)
rt.get(dev)
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
x, _ = S.proxy_crops(B, seed=1234)
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
cases = [('fit, walks', '', dict(fit='pose_shape', iters=50)), ('fit, level sweeps', '1', dict(fit='pose_shape', iters=50)),
         ('no fit', '', dict(fit=None))]
build = [int(c) for c in sys.argv[1]] if len(sys.argv) > 1 else [0, 1, 2]      # which cases to build ...
timed = [int(c) for c in sys.argv[2]] if len(sys.argv) > 2 else build          # ... and which to time (diagnosis)
cases = [cases[i] for i in build]
timed_names = [c[0] for i, c in zip(build, cases) if i in timed]
pipes = {}
for name, sweeps, kw in cases:
    os.environ.pop('SPSP_FIT_LEVEL_SWEEPS', None)
    if sweeps:
        os.environ['SPSP_FIT_LEVEL_SWEEPS'] = sweeps
    pipe = PlayerPipeline(mk_r, lambda: SMPL(model, max_batch=B), dev, depth=4, use_graphs=True, streams=streams)
    nat = [pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev)) for _ in range(4)]
    for i in range(8):
        out = pipe.submit(nat[i % 4], tgt, **kw)
        if os.environ.get('PROBE_DROP_OUT'):
            out = None
    torch.cuda.synchronize()
    trig = os.environ.get('PROBE_TRIGGER', 'clone')
    keep = None
    if out is None or os.environ.get('PROBE_NO_CLONE'):
        pass
    elif trig == 'clone':
        keep = out.vertices.clone()
    elif trig == 'clone_unrelated':
        keep = torch.zeros(32, 6890, 3, device=dev).clone()
    elif trig == 'clone_small':
        keep = tgt.clone()
    elif trig == 'clone_dropped':
        out.vertices.clone()
    elif trig == 'add':                     # a compute kernel with a fresh 2.6 MB output instead of a copy
        keep = out.vertices + 1.0
    elif trig == 'memset':
        keep = torch.zeros(32, 6890, 3, device=dev)
    pipes[name] = (pipe, nat, kw, keep)
os.environ.pop('SPSP_FIT_LEVEL_SWEEPS', None)
if 'fit, walks' in pipes and 'fit, level sweeps' in pipes:
    a, b = pipes['fit, walks'][3], pipes['fit, level sweeps'][3]
    print('walks vs level sweeps: max |dv| =', float((a - b).abs().max()), 'm')
res = {k: [] for k in pipes}
for rd in range(4):
    for name, (pipe, nat, kw, _) in pipes.items():
        if name not in timed_names:
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(nat[i % 4], tgt, **kw)
        th = time.perf_counter()
        torch.cuda.synchronize()
        res[name].append((round(B * steps / (time.perf_counter() - t0)), round(1e6 * (th - t0) / steps)))
for k, v in res.items():
    if v:
        print(f'{k:18s} {v}')
