"""The pipelined throughput mode alone (PlayerPipeline: 4 slots, HIP graphs, plan "throughput"; regressor -> fit -> SMPL at batch 32):
the command tools/pipeline_occupancy.py's kernel trace is taken from.
usage: python tools/pipeline_run.py [steps] [what=value,...] [plan]     (SPSP_DEBUG_PLAN_* hooks on every slot's regressor; plan: auto | latency | throughput)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402

from soccerplayershapepose_amd import synthetic as S  # noqa: E402
from soccerplayershapepose_amd import runtime as rt  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402
from soccerplayershapepose_amd.models.smpl_official import SMPL  # noqa: E402
from soccerplayershapepose_amd.player_recon import PlayerPipeline  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hooks = [tuple(int(v) for v in h.split('=')) for h in sys.argv[2].split(',')] if len(sys.argv) > 2 and sys.argv[2] else []
B = 32
dev = torch.device('cuda', 0)
r = rt.get(dev)
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)


def mk_r():
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd).to(dev)
    for what, val in hooks:
        r.check(r.lib.spsp_regressor_debug(reg.handle, what, val), 'spsp_regressor_debug')
    return reg


plan = sys.argv[3] if len(sys.argv) > 3 else 'auto'
pipe = PlayerPipeline(mk_r, lambda: SMPL(model, max_batch=B), dev, depth=4, use_graphs=True, plan=plan)
x, _ = S.proxy_crops(B, seed=1234)
natives = [pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev)) for _ in range(4)]
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
for i in range(16):
    pipe.submit(natives[i % 4], tgt, fit='pose_shape', iters=50)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    out = pipe.submit(natives[i % 4], tgt, fit='pose_shape', iters=50)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert torch.isfinite(out.vertices).all()
print(f'{B * steps / dt:9.0f} players/s, {1e6 * dt / steps:6.1f} us per batch of {B} ({steps} steps, hooks {hooks}, plan {plan})')
