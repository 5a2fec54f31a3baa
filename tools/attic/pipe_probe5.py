"""Fifth probe: what the latency-bound tail (fit, SMPL) costs the pipelined throughput -- the same pipeline with 50 / 10 / 1 fit
iterations and without the fit."""
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
mk_s = lambda: SMPL(model, max_batch=B)
x, _ = S.proxy_crops(B, seed=1234)
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
pipe = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True)
nat = [pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev)) for _ in range(4)]
cases = [('fit 50', dict(fit='pose_shape', iters=50)), ('fit 10', dict(fit='pose_shape', iters=10)), ('fit 1', dict(fit='pose_shape', iters=1)),
         ('no fit', dict(fit=None))]
res = {k: [] for k, _ in cases}
for rd in range(4):
    for name, kw in cases:
        for i in range(8):
            pipe.submit(nat[i % 4], tgt, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(nat[i % 4], tgt, **kw)
        torch.cuda.synchronize()
        res[name].append(round(B * steps / (time.perf_counter() - t0)))
for k, v in res.items():
    print(f'{k:8s} {v}')
