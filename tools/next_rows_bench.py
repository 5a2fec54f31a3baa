"""Timings of the SURVEY 8f rows that are not part of bench.py's step (one GPU, synthetic inputs), next to their CPU oracle:
  8f-1 front  frame [1080,1920,3] + 22 boxes -> 22 crops [512,512,3]           (HBM bytes / time)
  8f-3        multi-view joint fit, B players x 4 views, 3 rounds x 2 blocks x 50 epochs
  8f-4        512 x 512 silhouettes + metric of B meshes; the reference's best-iterate rule over a 50-iterate trajectory
Prints one JSON line; `--cpu` also times the oracle restatements on the host (test infrastructure, here only as the clock
beside the kernels, like bench.py's cpu_baseline)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from soccerplayershapepose_amd import synthetic as S  # noqa: E402
from soccerplayershapepose_amd.global_utils import crop_players  # noqa: E402
from soccerplayershapepose_amd.models.smpl_official import SMPL  # noqa: E402
from soccerplayershapepose_amd.player_recon import (broad_view_optimization, multi_view_optimization, single_view_optimization,  # noqa: E402
                                                    single_view_optimization_test)
from soccerplayershapepose_amd.renderer import SilhouetteRenderer  # noqa: E402
from soccerplayershapepose_amd.utils.cam_utils import convert_weak_perspective_to_camera_translation_torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cpu', action='store_true')
ap.add_argument('--reps', type=int, default=20)
args = ap.parse_args()
dev = torch.device('cuda', 0)


def gpu_time(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
rng = np.random.default_rng(0)
# ---- 8f-1 front: one 1080p frame, 22 detections
frame = torch.from_numpy(rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)).to(dev)
cx, cy = rng.integers(100, 1820, 22), rng.integers(150, 930, 22)
w, h = rng.integers(30, 90, 22), rng.integers(80, 220, 22)
boxes = np.stack([cx - w // 2, cy - h // 2, cx + w // 2, cy + h // 2], 1).astype(np.int32)
t = gpu_time(lambda: crop_players(frame, boxes), args.reps)
out['crop_players'] = {'boxes': 22, 'ms': 1e3 * t, 'written_MB': 22 * 512 * 512 * 3 / 1e6, 'GB_per_s': 22 * 512 * 512 * 3 / t / 1e9}
# ---- SMPL + poses
B = 32
model = S.smpl_model(seed=0)
smpl = SMPL(model, max_batch=B * 4).to(dev)
R, betas, _ = S.pose_params(B, seed=3, body_deg=20, global_deg=30)
R = torch.from_numpy(R).to(dev).contiguous(); betas = torch.from_numpy(betas).to(dev).contiguous()
cam = torch.tensor([[0.9, 0.02, -0.03]], device=dev).repeat(B, 1).contiguous()
verts = smpl(full_pose=R, betas=betas).vertices
trans = convert_weak_perspective_to_camera_translation_torch(cam, 5000.0, 512)
# ---- 8f-4: silhouettes + metric
rend = SilhouetteRenderer(model['faces'], max_batch=B).to(dev)
target = (rend(verts, trans).silhouettes > 0.5).float()
t = gpu_time(lambda: rend(verts, trans, target=target), args.reps)
out['render_silhouettes'] = {'meshes': B, 'ms': 1e3 * t, 'ms_per_mesh': 1e3 * t / B}
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
t = gpu_time(lambda: single_view_optimization(smpl, R[:, 1:].clone(), R[:, :1].clone(), betas.clone(), cam.clone(), tgt, interation=50,
                                              select_mode='reference', silhouette=target, renderer=rend), 3)
out['fit_with_reference_rule'] = {'players': B, 'iters': 50, 'ms': 1e3 * t, 'note': 'trajectory -> SMPL forward + render per iterate -> select'}
t = gpu_time(lambda: single_view_optimization(smpl, R[:, 1:].clone(), R[:, :1].clone(), betas.clone(), cam.clone(), tgt, interation=50), 5)
out['fit_l2e_rule'] = {'players': B, 'iters': 50, 'ms': 1e3 * t}
# ---- 8f-3: multi-view
V = 4
bp = R[:, None, 1:].repeat(1, V, 1, 1, 1).contiguous()
go = R[:, None, :1].repeat(1, V, 1, 1, 1).contiguous()
bt = betas[:, None].repeat(1, V, 1).contiguous()
tr = trans[:, None].repeat(1, V, 1).contiguous()
j2 = (torch.rand(B, V, 17, 2, generator=torch.Generator().manual_seed(5)) * 300 + 100).to(dev)
t = gpu_time(lambda: multi_view_optimization(smpl, bp, go, bt, tr, j2, interation=50), 3)
out['multi_view_optimization'] = {'players': B, 'views': V, 'rounds': 3, 'epochs_per_block': 50, 'ms': 1e3 * t, 'ms_per_player': 1e3 * t / B}
# ---- 8f-4, differentiable half: loss + gradient of 32 meshes, SMPL vertices VJP, the silhouette-in-loss fit (50 epochs), broad view
w = torch.full((B,), 1e6, device=dev)
t = gpu_time(lambda: rend.loss_grad(verts, trans, target, weight=w), args.reps)
out['silhouette_loss_grad'] = {'meshes': B, 'ms': 1e3 * t, 'note': 'ownership raster + loss sum + neural_renderer backward pass -> dL/dvertices, dL/dt'}
gv = torch.randn(B, 6890, 3, device=dev)
t = gpu_time(lambda: smpl.vertices_vjp(R, betas, gv), args.reps)
out['smpl_vertices_vjp'] = {'meshes': B, 'ms': 1e3 * t}
t = gpu_time(lambda: single_view_optimization_test(smpl, rend, R[:, 1:].clone(), R[:, :1].clone(), betas.clone(), cam.clone(), tgt, target, interation=50), 2)
out['fit_silhouette_in_loss'] = {'players': B, 'epochs': 50, 'ms': 1e3 * t, 'ms_per_epoch': 1e3 * t / 50}
t = gpu_time(lambda: broad_view_optimization(smpl, R[:, 1:].clone(), R[:, :1].clone(), betas.clone(), cam.clone(), tgt), 5)
out['broad_view_optimization'] = {'players': B, 'epochs': 100, 'ms': 1e3 * t}
if args.cpu:
    from oracle import geometry as OG, silhouette as OS
    t0 = time.perf_counter(); OG.crop_players(frame.cpu().numpy(), boxes[:4], 40); out['crop_players']['cpu_oracle_ms_per_box'] = 1e3 * (time.perf_counter() - t0) / 4
    v0, t0_ = verts[0].cpu().numpy(), trans[0].cpu().numpy()
    t0 = time.perf_counter(); OS.render_silhouette(v0, model["faces"], t0_); out['render_silhouettes']['cpu_oracle_ms_per_mesh'] = 1e3 * (time.perf_counter() - t0)
print(json.dumps(out))
