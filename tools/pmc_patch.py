"""Single-stream regressor forwards for PMC collection (rocprofv3 --pmc ... -- python3 tools/pmc_patch.py)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
dev = torch.device('cuda', 0)
B = 32
reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(S.regressor_state_dict(seed=0)).to(dev)
x, _ = S.proxy_crops(B, seed=1)
native = reg.pack(torch.from_numpy(x).to(dev))
for _ in range(3):
    reg(native)
torch.cuda.synchronize()
