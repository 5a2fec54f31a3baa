"""Eighth probe: what makes a pipeline faster once ANOTHER pipeline of the process has run (ab_pipeline's first round reads 110 k,
every later one 124 k; a single-pipeline process stays at 109 k).  Sequence of timed 200-step loops: A A | B | A A | A with no fit | A A."""
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
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
x, _ = S.proxy_crops(B, seed=1234)
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
def build():
    pipe = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True, streams=streams)
    nat = [pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev)) for _ in range(4)]
    return pipe, nat
def run(pipe, nat, label, **kw):
    kw = kw or dict(fit='pose_shape', iters=50)
    for i in range(8):
        pipe.submit(nat[i % 4], tgt, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipe.submit(nat[i % 4], tgt, **kw)
    torch.cuda.synchronize()
    print(f'{label:34s} {B * steps / (time.perf_counter() - t0):9.0f} players/s', flush=True)
A = build()
run(*A, 'A'); run(*A, 'A')
if 'sleep' in sys.argv:
    time.sleep(1.0); run(*A, 'A after 1 s idle'); run(*A, 'A')
run(*A, 'A without the fit', fit=None); run(*A, 'A'); run(*A, 'A')
Bp = build()
run(*Bp, 'B (second pipeline, same streams)'); run(*A, 'A'); run(*A, 'A'); run(*Bp, 'B'); run(*A, 'A')
