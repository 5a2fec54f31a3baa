"""SMPL forward alone: per-kernel durations (the library's profiling scopes: each kernel's own begin / end timestamps) at
B = 32 / 22 / 8 / 1, and the largest vertex / joint deviation from the float64 oracle twin -- with 8 (default) and 16 players per
skin workgroup (`spsp_smpl_debug(SPSP_DEBUG_SKIN_PLAYERS)`: the blend basis read four times / twice per 32 players).
usage: python tools/smpl_time.py [reps [batch ...]]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from soccerplayershapepose_amd import _abi, synthetic as S  # noqa: E402
from soccerplayershapepose_amd import runtime as rt  # noqa: E402
from soccerplayershapepose_amd.models.smpl_official import SMPL  # noqa: E402
from oracle import smpl as OS  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
model = S.smpl_model(seed=0)
r = rt.get(dev)
nk = r.lib.spsp_profile_num_kernels()
names = [r.lib.spsp_profile_kernel_name(i).decode() for i in range(nk)]
for B, PB in [(b_, pb_) for b_ in ([int(v) for v in sys.argv[2:]] or (32, 22, 8, 1)) for pb_ in (8, 16)]:
    smpl = SMPL(model, max_batch=B).to(dev)
    smpl.debug(_abi.DEBUG_SKIN_PLAYERS, PB)
    R, b, cam = S.pose_params(B, seed=5 + B, body_deg=30, global_deg=60)
    Rt, bt = torch.from_numpy(R).to(dev), torch.from_numpy(b).to(dev)
    out = smpl(body_pose=Rt[:, 1:], global_orient=Rt[:, :1], betas=bt)
    torch.cuda.synchronize()
    v64, j64 = OS.smpl_forward_np(model, R, b)
    ev = np.abs(out.vertices.cpu().numpy() - v64).max()
    ej = np.abs(out.joints.cpu().numpy() - j64).max()
    ms, cnt = (C.c_double * nk)(), (C.c_int * nk)()
    r.check(r.lib.spsp_profile_select(r.ctx, -1), 'select')
    r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'enable')
    for _ in range(reps):
        smpl(body_pose=Rt[:, 1:], global_orient=Rt[:, :1], betas=bt)
        torch.cuda.synchronize()
    r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'enable')
    r.check(r.lib.spsp_profile_read(r.ctx, nk, ms, cnt), 'read')
    k = ', '.join(f'{names[i]} {1e3 * ms[i] / cnt[i]:.2f} us' for i in range(nk) if cnt[i])
    print(f'B={B}, {PB} players per skin workgroup: {k}; max |vertex - float64 oracle| {ev:.2e} m, joints {ej:.2e} m', flush=True)
