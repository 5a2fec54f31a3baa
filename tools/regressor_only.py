"""A few regressor forwards at batch 32, nothing else: the command profiled by the rocprofv3 --pmc passes
(profiles/*_pmc_sq_*): one batch in flight, so the counters are per kernel, not per mix.
usage: python tools/regressor_only.py [B] [forwards] [f16|bf16] [latency|throughput]      (plan "latency" = a handle's default)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from soccerplayershapepose_amd import synthetic as S  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prec = sys.argv[3] if len(sys.argv) > 3 else 'f16'
plan = sys.argv[4] if len(sys.argv) > 4 else 'latency'
dev = torch.device('cuda', 0)
reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec, plan=plan).load_state_dict(S.regressor_state_dict(seed=0)).to(dev)
x, _ = S.proxy_crops(B, seed=1)
native = reg.pack(torch.from_numpy(x).to(dev))
for _ in range(N):
    reg(native)
torch.cuda.synchronize()
