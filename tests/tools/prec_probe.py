"""Deviation of the GPU regressor (bf16 / f16 backbone) from the all-fp32 oracle, before and after the fit."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import reconstruct_players
from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
from oracle import regressor as OR, geometry as G, smpl as OS, fit as OF
dev = torch.device('cuda', 0)
B = 3
sd = S.regressor_state_dict(seed=0); model = S.smpl_model(seed=0)
x, _ = S.proxy_crops(B, seed=12)
xt = torch.from_numpy(x).to(dev)
osm = OS.OracleSMPL(model)
smpl = SMPL(model, max_batch=8).to(dev)
cam_o, pose_o, shape_o = OR.regressor_forward(sd, x, S.mean_params())
f_o = OR.encoder_forward(sd, x)
R_o = G.rot6d_to_rotmat(pose_o).view(B, 24, 3, 3)
R2, b2, _ = S.pose_params(B, seed=13, body_deg=8, global_deg=10)
Rp = (R2.astype(np.float64) @ R_o.numpy().astype(np.float64)).astype(np.float32)
_, jp = OS.smpl_forward_np(model, Rp, shape_o.numpy() + b2 / 3)
tgt = S.fit_targets_from_joints(jp, cam_o.numpy() * np.array([1.1, 1, 1], np.float32), seed=14)
vo = []
for i in range(B):
    r = OF.fit_pose_shape(osm, R_o[i:i+1, 1:], R_o[i:i+1, :1], shape_o[i:i+1], tgt[i], cam_o[i:i+1], iters=50)
    body, Gm, be, ca, _ = r['last']
    with torch.no_grad(): vo.append(osm(body_pose=body, global_orient=Gm, betas=be).vertices[0])
vo = torch.stack(vo)
with torch.no_grad(): pre_o = osm(body_pose=R_o[:, 1:], global_orient=R_o[:, :1], betas=shape_o).vertices
for prec in ('bf16', 'f16'):
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=8, precision=prec).load_state_dict(sd).to(dev)
    cam, pose, shape, feats = reg(xt, return_feats=True)
    f_e = OR.encoder_forward(sd, x, emulate_bf16=prec)
    print(prec, 'feats vs fp32 oracle: max %.2e (scale %.2f); vs emulation %.2e; emulation vs fp32 %.2e' % (
        (feats.cpu() - f_o).abs().max().item(), f_o.abs().max().item(), (feats.cpu() - f_e).abs().max().item(), (f_e - f_o).abs().max().item()))
    print(prec, 'cam %.2e pose6d %.2e shape %.2e' % ((cam.cpu() - cam_o).abs().max().item(), (pose.cpu() - pose_o).abs().max().item(), (shape.cpu() - shape_o).abs().max().item()))
    R = rot6d_to_rotmat(pose).view(B, 24, 3, 3)
    pre = smpl(full_pose=R, betas=shape).vertices.cpu()
    res = reconstruct_players(reg, smpl, xt, tgt, fit='pose_shape', iters=50)
    print(prec, 'vertices before fit %.2e, after fit %.2e' % ((pre - pre_o).abs().max().item(), (res.vertices.cpu() - vo).abs().max().item()),
          [round((res.vertices[i].cpu() - vo[i]).abs().max().item(), 6) for i in range(B)])

print('--- targets derived from the GPU regressor outputs, as in tests/test_gpu_parity.py ---')
for prec in ('bf16', 'f16'):
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=32, precision=prec).load_state_dict(sd).to(dev)
    cam0, pose0, shape0 = reg(xt)
    R0 = rot6d_to_rotmat(pose0).view(B, 24, 3, 3)
    Rp = (R2.astype(np.float64) @ R0.cpu().numpy().astype(np.float64)).astype(np.float32)
    _, jp = OS.smpl_forward_np(model, Rp, shape0.cpu().numpy() + b2 / 3)
    tgt2 = S.fit_targets_from_joints(jp, cam0.cpu().numpy() * np.array([1.1, 1, 1], np.float32), seed=14)
    res = reconstruct_players(reg, smpl, xt, tgt2, fit='pose_shape', iters=50)
    w = []
    for i in range(B):
        r = OF.fit_pose_shape(osm, R_o[i:i+1, 1:], R_o[i:i+1, :1], shape_o[i:i+1], tgt2[i], cam_o[i:i+1], iters=50)
        body, Gm, be, ca, _ = r['last']
        with torch.no_grad(): v = osm(body_pose=body, global_orient=Gm, betas=be).vertices[0]
        w.append((res.vertices[i].cpu() - v).abs().max().item())
        dR = (res.rotmats[i].cpu() - torch.cat([Gm, body], 1)[0]).abs().reshape(24, 9).max(1).values
        print(prec, 'player', i, 'vertex dev %.2e' % w[-1], 'cam dev %.2e betas dev %.2e' % ((res.cam[i].cpu() - ca[0]).abs().max().item(), (res.betas[i].cpu() - be[0]).abs().max().item()),
              'worst joints (R dev):', [(int(k), round(dR[k].item(), 5)) for k in dR.argsort(descending=True)[:4]])
