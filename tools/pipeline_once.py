"""The whole hot path (regressor -> rot6d -> 50-iteration fit -> SMPL forward) at batch 32, a few times, one batch in
flight: the command profiled by the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes behind profiles/hbm_traffic.json.
usage: python tools/pipeline_once.py [B] [repeats] [players per skin workgroup: 8 | 16] [regressor plan: latency | throughput]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from soccerplayershapepose_amd import synthetic as S  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402
from soccerplayershapepose_amd.models.smpl_official import SMPL  # noqa: E402
from soccerplayershapepose_amd.player_recon import reconstruct_players  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda', 0)
reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(S.regressor_state_dict(seed=0)).to(dev)
smpl = SMPL(S.smpl_model(seed=0), max_batch=B).to(dev)
if len(sys.argv) > 3:
    from soccerplayershapepose_amd import _abi
    smpl.debug(_abi.DEBUG_SKIN_PLAYERS, int(sys.argv[3]))
if len(sys.argv) > 4:
    reg.set_plan(sys.argv[4])
x, _ = S.proxy_crops(B, seed=1)
native = reg.pack(torch.from_numpy(x).to(dev))
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
for _ in range(N):
    reconstruct_players(reg, smpl, native, tgt, fit='pose_shape', iters=50)
torch.cuda.synchronize()
