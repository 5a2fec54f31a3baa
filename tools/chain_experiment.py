"""How much of the pipelined step time is the serial tail of a stream's chain?  Throughput of the 4-slot graph-replay
pipeline with the IEF shortened (ief_iters 3 / 1 / 0) and the fit shortened (50 / 10 / 0 iterations): results are NOT
the product's numbers, the deltas say what a faster IEF / fit would be worth."""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.getcwd())
import torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import PlayerPipeline
dev = torch.device('cuda', 0)
B = 32
sd = S.regressor_state_dict(seed=0); model = S.smpl_model(seed=0)
x, _ = S.proxy_crops(B, seed=1)
tgt = torch.rand(B, 17, 2, device=dev) * 512
def run(ief, fit_iters, steps=200):
    pipe = PlayerPipeline(lambda: SingleInputRegressor(18, 18, ief, mean_params=S.mean_params(), max_batch=B).load_state_dict(sd),
                          lambda: SMPL(model, max_batch=B), dev, depth=4, use_graphs=True)
    native = pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev))
    kw = dict(fit='pose_shape' if fit_iters else None, iters=fit_iters or 50)
    for _ in range(12): pipe.submit(native, tgt, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): pipe.submit(native, tgt, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'ief_iters {ief}  fit_iters {fit_iters:3d}: {1e6 * dt / steps:7.1f} us/step  {B * steps / dt:9.0f} players/s', flush=True)
for ief, fi in ((3, 50), (1, 50), (0, 50), (3, 10), (3, 0), (0, 0)):
    run(ief, fi)
