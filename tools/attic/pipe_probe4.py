"""Fourth probe: K identical pipelines in one process, measured in interleaved rounds (what tools/ab_pipeline.py does)."""
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
K, shared, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B, steps = 32, 200
dev = torch.device('cuda', 0)
rt.get(dev)
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd)
mk_s = lambda: SMPL(model, max_batch=B)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)] if shared else None
x, _ = S.proxy_crops(B, seed=1234)
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
pipes = []
for k in range(K):
    pipe = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True, streams=streams)
    nat = [pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev)) for _ in range(4)]
    for i in range(8):
        pipe.submit(nat[i % 4], tgt, fit='pose_shape', iters=50)
    torch.cuda.synchronize()
    pipes.append((pipe, nat))
res = [[] for _ in pipes]
for rd in range(rounds):
    for k, (pipe, nat) in enumerate(pipes):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(nat[i % 4], tgt, fit='pose_shape', iters=50)
        torch.cuda.synchronize()
        res[k].append(round(B * steps / (time.perf_counter() - t0)))
print(f'K={K} shared={shared}:', res)
