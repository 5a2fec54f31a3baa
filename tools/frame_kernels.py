"""Per-kernel durations of ONE frame of N players through the whole path (plan "latency", one batch in flight, eager with the
library's profiling scopes): where the 22-player latency of bench.py goes.   usage: python tools/frame_kernels.py [N] [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from soccerplayershapepose_amd import synthetic as S  # noqa: E402
from soccerplayershapepose_amd import runtime as rt  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402
from soccerplayershapepose_amd.models.smpl_official import SMPL  # noqa: E402
from soccerplayershapepose_amd.player_recon import reconstruct_players  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
r = rt.get(dev)
reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=32, precision='f16', plan='latency').load_state_dict(S.regressor_state_dict(seed=0)).to(dev)
smpl = SMPL(S.smpl_model(seed=0), max_batch=32).to(dev)
x, _ = S.proxy_crops(N, seed=1)
native = reg.pack(torch.from_numpy(x).to(dev))
tgt = (torch.rand(N, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
for _ in range(3):
    reconstruct_players(reg, smpl, native, tgt, fit='pose_shape', iters=50)
torch.cuda.synchronize()
nk = r.lib.spsp_profile_num_kernels()
names = [r.lib.spsp_profile_kernel_name(i).decode() for i in range(nk)]
ms, cnt, wgs = (C.c_double * nk)(), (C.c_int * nk)(), (C.c_double * nk)()
r.check(r.lib.spsp_profile_select(r.ctx, -1), 'select')
r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'enable')
for _ in range(reps):
    reconstruct_players(reg, smpl, native, tgt, fit='pose_shape', iters=50)
    torch.cuda.synchronize()
r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'enable')
r.check(r.lib.spsp_profile_read_grids(r.ctx, nk, ms, cnt, wgs), 'read')
tot = 0.0
for i in sorted(range(nk), key=lambda i: -ms[i]):
    if cnt[i]:
        print(f'{names[i]:34s} {cnt[i] / reps:4.1f} launches  {1e3 * ms[i] / cnt[i]:7.2f} us each  {1e3 * ms[i] / reps:7.2f} us / frame   {wgs[i] / cnt[i]:6.0f} workgroups')
        tot += 1e3 * ms[i] / reps
print(f'sum {tot:.1f} us per frame of {N} players')
