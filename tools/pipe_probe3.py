"""Third probe: streams created up front (before any handle is loaded) vs created one per slot between the handle loads."""
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
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd)
mk_s = lambda: SMPL(model, max_batch=B)
streams = None
if mode == 'upfront':
    rt.get(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
elif mode == 'upfront_noinit':
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
elif mode == 'upfront8':
    rt.get(dev)
    allst = [torch.cuda.Stream(device=dev) for _ in range(8)]
    streams = allst[4:]
pipe = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True, streams=streams)
x, _ = S.proxy_crops(B, seed=1234)
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
nat = [pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev)) for _ in range(4)]
for i in range(8):
    pipe.submit(nat[i % 4], tgt, fit='pose_shape', iters=50)
torch.cuda.synchronize()
v = []
for rd in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipe.submit(nat[i % 4], tgt, fit='pose_shape', iters=50)
    torch.cuda.synchronize()
    v.append(B * steps / (time.perf_counter() - t0))
print(f'{mode:16s} {sorted(v)[len(v) // 2]:9.0f} players/s (rounds {[round(a) for a in v]})')
