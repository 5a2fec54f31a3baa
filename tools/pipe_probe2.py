"""Second probe of the harness difference: stream ownership and allocation layout."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import PlayerPipeline
B, steps = 32, 200
dev = torch.device('cuda', 0)
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd)
mk_s = lambda: SMPL(model, max_batch=B)
x, _ = S.proxy_crops(B, seed=1234)
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
pipes = {}
pipes['A: first pipeline, own streams'] = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True)
shared = [torch.cuda.Stream(device=dev) for _ in range(4)]
pipes['B: second pipeline, explicit streams'] = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True, streams=shared)
pipes['C: third pipeline, the same explicit streams'] = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True, streams=shared)
pipes['D: fourth pipeline, own streams'] = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True)
nat = {}
for name, p in pipes.items():
    nat[name] = [p.slots[0].regressor.pack(torch.from_numpy(x).to(dev)) for _ in range(4)]
    for i in range(8):
        p.submit(nat[name][i % 4], tgt, fit='pose_shape', iters=50)
torch.cuda.synchronize()
res = {k: [] for k in pipes}
for rd in range(4):
    for name, p in pipes.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            p.submit(nat[name][i % 4], tgt, fit='pose_shape', iters=50)
        torch.cuda.synchronize()
        res[name].append(B * steps / (time.perf_counter() - t0))
for name, v in res.items():
    print(f'{sorted(v)[len(v) // 2]:9.0f} players/s (rounds {[round(a) for a in v]})  {name}')
print('streams A', [hex(s.stream.cuda_stream) for s in pipes['A: first pipeline, own streams'].slots], 'shared', [hex(s.cuda_stream) for s in shared])
