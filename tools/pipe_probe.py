"""Why does tools/ab_pipeline.py read 18 % more players/s than bench.py on the same box?  One pipeline (default plan), measured
in interleaved rounds under variations of what differs between the two harnesses."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import PlayerPipeline
from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
B, steps = 32, 200
dev = torch.device('cuda', 0)
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd)
mk_s = lambda: SMPL(model, max_batch=B)
pipe = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs=True)
reg, smpl = pipe.slots[0].regressor, pipe.slots[0].smpl
def realistic(k):
    x, _ = S.proxy_crops(B, seed=1234 + 1000 * k)
    native = reg.pack(torch.from_numpy(x).to(dev))
    cam0, pose0, shape0 = reg(native)
    R0 = rot6d_to_rotmat(pose0).view(B, 24, 3, 3)
    R2, b2, _ = S.pose_params(B, seed=77 + 1000 * k, body_deg=8, global_deg=10)
    Rp = torch.matmul(torch.from_numpy(R2).to(dev), R0)
    jp = smpl(body_pose=Rp[:, 1:], global_orient=Rp[:, :1], betas=shape0 + torch.from_numpy(b2).to(dev) / 3, return_verts=False).joints
    t = S.fit_targets_from_joints(jp.cpu().numpy(), cam0.cpu().numpy() * np.array([1.1, 1, 1], np.float32), seed=4321 + 1000 * k)
    return native, torch.from_numpy(t[:, :, :2]).float().to(dev).contiguous()
real = [realistic(k) for k in range(4)]
same_native = [real[0][0].clone() for _ in range(4)]
rand_tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
CASES = {
    'distinct crops, realistic targets (bench.py)': lambda i: (real[i % 4][0], real[i % 4][1]),
    'distinct crops, ONE random target set': lambda i: (real[i % 4][0], rand_tgt),
    'identical crops x4, ONE random target set (ab_pipeline.py)': lambda i: (same_native[i % 4], rand_tgt),
    'identical crops x4, realistic targets of batch 0': lambda i: (same_native[i % 4], real[0][1]),
}
for name, pick in CASES.items():
    for i in range(8):
        pipe.submit(*pick(i), fit='pose_shape', iters=50)
torch.cuda.synchronize()
res = {k: [] for k in CASES}
for rd in range(4):
    for name, pick in CASES.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(*pick(i), fit='pose_shape', iters=50)
        torch.cuda.synchronize()
        res[name].append(B * steps / (time.perf_counter() - t0))
for name, v in res.items():
    print(f'{sorted(v)[len(v) // 2]:9.0f} players/s (rounds {[round(a) for a in v]})  {name}')
