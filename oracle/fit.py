"""Oracle: the per-player reprojection fit loops (Appendix D/E).
Test infrastructure only.  Inherits "parity unpinned" from the SMPL restatement (oracle/smpl.py).

Both loops follow the reference line-for-line -- full `SMPL.forward` every iteration, the joints2D
branch of the multi-task loss, real `torch.autograd` and a fresh `torch.optim.Adam` per player --
minus the NMR silhouette render, whose output never enters the loss (`losses_on=['joints2D']`,
`PR/PyTorch3DTest.py:3023-3027`, `PR/player_recon.py:1172`):

  fit_camera       `optimize_camera`              PR/PyTorch3DTest.py:3055-3124  (params: cam_wp, global_orient)
  fit_pose_shape   `single_view_optimization`     PR/player_recon.py:1172-1294   (params: global_orient,
                                                  body joints idx 0-5 & 8-20, cam_wp, betas)
  fit_pose_shape_batch  the same for B independent players in one autograd graph (test-suite accelerator; checked against the
                                                  per-player loop in tests/test_oracle_golden.py)
"""
import numpy as np
import torch
from torch import optim

from .geometry import (SMPL_TO_KPRCNN_MAP, PROXY_REP_INPUT_WH, joints2d_loss, joints2d_l2e,
                       orthographic_project, undo_keypoint_normalisation, wp_to_translation)


def _project17(smpl_out, cam_wp):
    j2d = orthographic_project(smpl_out.joints, cam_wp)[:, SMPL_TO_KPRCNN_MAP, :]
    return undo_keypoint_normalisation(j2d, PROXY_REP_INPUT_WH)


def fit_camera(smpl, body_pose, global_orient, betas, joints2D, cam_wp, lr=0.01, iters=50,
               record=None):
    """Variant A for ONE player (batch dim 1), mutating `global_orient` / `cam_wp` in place like
    `optimize_camera` (`PR/PyTorch3DTest.py:3055`); the LAST iterate is what is left behind.

    body_pose [1,23,3,3], global_orient [1,1,3,3], betas [1,10], cam_wp [1,3] fp32 tensors;
    joints2D [17,>=2] numpy (512-px space).  `record`: optional dict filled with per-iteration
    loss / l2e / projected joints / parameter snapshots."""
    cam_wp.requires_grad = True
    global_orient.requires_grad = True
    optimiser = optim.Adam([cam_wp, global_orient], lr=lr)
    target = torch.from_numpy(np.asarray(joints2D)[:, :2]).float()
    for epoch in range(1, iters + 1):
        out = smpl(body_pose=body_pose, global_orient=global_orient, betas=betas, pose2rot=False)
        pred = _project17(out, cam_wp)
        optimiser.zero_grad()
        loss = joints2d_loss(pred[0], target)
        if record is not None:
            record.setdefault('loss', []).append(float(loss.item()))
            record.setdefault('l2e', []).append(joints2d_l2e(pred[0].detach().numpy(), target.numpy()))
            record.setdefault('pred', []).append(pred[0].detach().numpy().copy())
        loss.backward()
        if record is not None and epoch == 1:
            record['grad_cam'] = cam_wp.grad.detach().numpy().copy()
            record['grad_G'] = global_orient.grad.detach().numpy().copy()
        optimiser.step()
        if record is not None:
            record.setdefault('cam', []).append(cam_wp.detach().numpy().copy())
            record.setdefault('G', []).append(global_orient.detach().numpy().copy())
    cam_wp.requires_grad = False
    global_orient.requires_grad = False


def fit_pose_shape(smpl, body_pose, global_orient, betas, joints2D, cam_wp, lr=1e-3, iters=50,
                   record=None):
    """Variant B for ONE player (`PR/player_recon.py:1175-1282`).

    Returns dict(last=..., best=...) of (body_pose[1,23,3,3], global_orient, betas, cam_wp,
    translation) tuples.  `best` is the iterate with the lowest joints2D_l2es evaluated BEFORE the
    step, ties resolved to the LATER iterate (`determine_save_model_weights_this_epoch` uses `>`,
    `PR/metrics/train_loss_and_metrics_tracker.py:291-298`).  The reference additionally gates on a
    rasterised-silhouette metric (out of scope) and saves the un-optimised body_pose
    (reference bug); the optimised one is returned here."""
    body_pose = body_pose.detach().clone()
    global_orient = global_orient.detach().clone()
    betas = betas.detach().clone()
    cam_wp = cam_wp.detach().clone()
    opt_body = torch.cat([body_pose[:, :6], body_pose[:, 8:21]], dim=1).detach().clone()
    for p in (global_orient, betas, opt_body, cam_wp):
        p.requires_grad = True
    optimiser = optim.Adam([global_orient, opt_body, cam_wp, betas], lr=lr)
    target = torch.from_numpy(np.asarray(joints2D)[:, :2]).float()

    def assemble():
        return torch.cat([opt_body[:, :6], body_pose[:, 6:8], opt_body[:, 6:19], body_pose[:, 21:]], dim=1)

    def snapshot():
        with torch.no_grad():
            return (assemble().clone(), global_orient.detach().clone(), betas.detach().clone(),
                    cam_wp.detach().clone(), wp_to_translation(cam_wp.detach()))

    best, best_l2e = snapshot(), np.inf
    for epoch in range(1, iters + 1):
        full_body = assemble()
        out = smpl(body_pose=full_body, global_orient=global_orient, betas=betas, pose2rot=False)
        pred = _project17(out, cam_wp)
        optimiser.zero_grad()
        loss = joints2d_loss(pred[0], target)
        l2e = joints2d_l2e(pred[0].detach().numpy(), target.numpy())
        if not (l2e > best_l2e):
            best, best_l2e = snapshot(), l2e
        if record is not None:
            record.setdefault('loss', []).append(float(loss.item()))
            record.setdefault('l2e', []).append(l2e)
            record.setdefault('pred', []).append(pred[0].detach().numpy().copy())
        loss.backward()
        if record is not None and epoch == 1:
            record['grad_G'] = global_orient.grad.detach().numpy().copy()
            record['grad_body'] = opt_body.grad.detach().numpy().copy()
            record['grad_cam'] = cam_wp.grad.detach().numpy().copy()
            record['grad_betas'] = betas.grad.detach().numpy().copy()
        optimiser.step()
        if record is not None:
            s = snapshot()
            record.setdefault('body', []).append(s[0].numpy())
            record.setdefault('G', []).append(s[1].numpy())
            record.setdefault('betas', []).append(s[2].numpy())
            record.setdefault('cam', []).append(s[3].numpy())
    return {'last': snapshot(), 'best': best, 'best_l2e': best_l2e}


def fit_pose_shape_batch(smpl, body_pose, global_orient, betas, joints2D, cam_wp, lr=1e-3, iters=50):
    """`fit_pose_shape` for B INDEPENDENT players in one autograd graph (a test-suite accelerator: a per-player fit is 2.4 s of
    framework overhead on tensors of one player).  Exactly the per-player algorithm: the loss is the SUM of the players'
    joints2D losses, so a player's gradient is the gradient of its own loss, and Adam is elementwise with one step count, so
    one optimiser over the stacked parameters takes the steps B fresh optimisers would.  What differs is the floating-point
    order inside batched matrix products (checked against the per-player loop: tests/test_oracle_golden.py).
    body_pose [B,23,3,3], global_orient [B,1,3,3], betas [B,10], cam_wp [B,3] tensors; joints2D [B,17,>=2].
    Returns dict(last=(body_pose, global_orient, betas, cam_wp, translation), best=(... per player), best_l2e [B])."""
    body_pose = body_pose.detach().clone()
    global_orient = global_orient.detach().clone()
    betas = betas.detach().clone()
    cam_wp = cam_wp.detach().clone()
    B = body_pose.shape[0]
    opt_body = torch.cat([body_pose[:, :6], body_pose[:, 8:21]], dim=1).detach().clone()
    for p in (global_orient, betas, opt_body, cam_wp):
        p.requires_grad = True
    optimiser = optim.Adam([global_orient, opt_body, cam_wp, betas], lr=lr)
    target = torch.from_numpy(np.asarray(joints2D, dtype=np.float32)[:, :, :2]).float()

    def assemble():
        return torch.cat([opt_body[:, :6], body_pose[:, 6:8], opt_body[:, 6:19], body_pose[:, 21:]], dim=1)

    def snapshot():
        with torch.no_grad():
            return [assemble().clone(), global_orient.detach().clone(), betas.detach().clone(), cam_wp.detach().clone(),
                    wp_to_translation(cam_wp.detach())]

    best, best_l2e = snapshot(), np.full(B, np.inf)
    for epoch in range(1, iters + 1):
        out = smpl(body_pose=assemble(), global_orient=global_orient, betas=betas, pose2rot=False)
        pred = _project17(out, cam_wp)
        optimiser.zero_grad()
        loss = sum(joints2d_loss(pred[b], target[b]) for b in range(B))
        cur = snapshot()
        for b in range(B):
            l2e = joints2d_l2e(pred[b].detach().numpy(), target[b].numpy())
            if not (l2e > best_l2e[b]):
                best_l2e[b] = l2e
                for dst, src in zip(best, cur):
                    dst[b] = src[b]
        loss.backward()
        optimiser.step()
    return {'last': tuple(snapshot()), 'best': tuple(best), 'best_l2e': best_l2e}


def fit_broad_view(smpl, body_pose, global_orient, betas, joints2D, cam_wp, lr=1e-2, iters=100, silhouette_metric=None):
    """`broad_view_optimization` (`PR/player_recon.py:2128-2229`) for ONE player: Adam([cam_wp, global_orient]); every epoch a
    train pass (step) then a validation pass on the stepped parameters; an epoch is saved iff its validation joints2D_l2es --
    and, when `silhouette_metric(vertices [6890,3], cam_wp [3]) -> float` is given, that metric too -- is <= its value at the
    previous save (`determine_save_model_weights_this_epoch`).  Returns dict(best=(G, cam), best_epoch (1-based), val_l2e, val_sil)."""
    global_orient = global_orient.detach().clone().requires_grad_(True)
    cam_wp = cam_wp.detach().clone().requires_grad_(True)
    optimiser = optim.Adam([cam_wp, global_orient], lr=lr)
    target = torch.from_numpy(np.asarray(joints2D)[:, :2]).float()
    best, best_epoch, bl, bs = (global_orient.detach().clone(), cam_wp.detach().clone()), 0, np.inf, np.inf
    val_l2e, val_sil = [], []
    for epoch in range(1, iters + 1):
        for is_train in (True, False):
            out = smpl(body_pose=body_pose, global_orient=global_orient, betas=betas, pose2rot=False)
            pred = _project17(out, cam_wp)
            if is_train:
                optimiser.zero_grad()
                joints2d_loss(pred[0], target).backward()
                optimiser.step()
            else:
                l2e = joints2d_l2e(pred[0].detach().numpy(), target.numpy())
                sm = silhouette_metric(out.vertices[0].detach().numpy(), cam_wp[0].detach().numpy()) if silhouette_metric else 0.0
                val_l2e.append(l2e)
                val_sil.append(sm)
                if not (l2e > bl) and not (silhouette_metric is not None and sm > bs):
                    bl, bs, best_epoch = l2e, sm, epoch
                    best = (global_orient.detach().clone(), cam_wp.detach().clone())
    return {'best': best, 'best_epoch': best_epoch, 'val_l2e': np.asarray(val_l2e), 'val_sil': np.asarray(val_sil)}
