"""The 30-frame warm-start clip (BASELINE configs[5]) free-running on both sides: the GPU chains its own fits, the CPU oracle chains its
own from the same regressor outputs; vertices of the followed players after every chained fit.  How far two CORRECT implementations of
the same 50-iteration Adam fit drift apart when each frame starts from the previous result -- with the product's fit kernel (schedule 2:
1-ulp reciprocal / square root in the Adam step) and with the ten-barrier kernel (schedule 1: IEEE division and square root).
usage: python tools/clip_chain.py [frames] [regressor precision: f16 | bf16]   (the precision only selects the trajectory: both chains start from the GPU regressor's outputs)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from soccerplayershapepose_amd import _abi, synthetic as S  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402
from soccerplayershapepose_amd.models.smpl_official import SMPL  # noqa: E402
from soccerplayershapepose_amd.player_recon import reconstruct_players, single_view_optimization  # noqa: E402
from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat  # noqa: E402
from oracle import fit as OF, smpl as OS  # noqa: E402  (test infrastructure: this tool is a checker, not the product)

T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
N, follow = 22, [0, 21]
model = S.smpl_model(seed=0)
reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=N, precision=sys.argv[2] if len(sys.argv) > 2 else 'f16').load_state_dict(S.regressor_state_dict(seed=0)).to(dev)
smpl = SMPL(model, max_batch=N).to(dev)
oracle_smpl = OS.OracleSMPL(model)
x, _ = S.proxy_crops(N, seed=31)
xt = torch.from_numpy(x).to(dev)
cam0, pose0, shape0 = reg(xt)
R0 = rot6d_to_rotmat(pose0).view(N, 24, 3, 3)
Rt_true = R0.cpu().numpy().astype(np.float64)
targets = []
for t in range(T):
    step, _, _ = S.pose_params(N, seed=40 + t, body_deg=2, global_deg=2)
    Rt_true = step.astype(np.float64) @ Rt_true
    _, jp = OS.smpl_forward_np(model, Rt_true.astype(np.float32), shape0.cpu().numpy())
    targets.append(S.fit_targets_from_joints(jp, cam0.cpu().numpy() * np.array([1.05, 1, 1], np.float32), seed=50 + t))
# the oracle's own chain, once
free = (R0[follow, 1:].cpu(), R0[follow, :1].cpu(), shape0[follow].cpu(), cam0[follow].cpu())
oracle_v = []
for t in range(T):
    tg = np.stack([targets[t][i] for i in follow])
    f = OF.fit_pose_shape_batch(oracle_smpl, free[0], free[1], free[2], tg, free[3], iters=50)['last']
    free = tuple(f[k].detach() for k in range(4))
    with torch.no_grad():
        oracle_v.append(oracle_smpl(body_pose=free[0], global_orient=free[1], betas=free[2]).vertices)
    print(f'oracle frame {t}', flush=True)
for sched, name in ((2, 'schedule v2, 1-ulp rcp / sqrt (the product)'), (1, 'ten-barrier schedule, IEEE division / sqrt')):
    smpl.debug(_abi.DEBUG_FIT_SCHEDULE, sched)
    res = reconstruct_players(reg, smpl, xt, targets[0], fit='pose_shape', iters=50)
    body, orient, betas, cam = res.rotmats[:, 1:], res.rotmats[:, :1], res.betas, res.cam
    dev_t = []
    for t in range(T):
        if t > 0:
            rec = single_view_optimization(smpl, body, orient, betas, cam, targets[t], interation=50)
            body, orient, betas, cam = rec.body_pose, rec.global_orient, rec.betas, rec.cam_wp
        out = smpl(body_pose=body, global_orient=orient, betas=betas)
        dev_t.append((out.vertices[follow].cpu() - oracle_v[t]).abs().max().item())
    print(f'{name}: max |vertex - free-running oracle chain| per frame (m): ' + ' '.join(f'{d:.1e}' for d in dev_t), flush=True)
