"""Oracle: the multi-view joint fit.  Test infrastructure only; inherits "parity unpinned" from the
SMPL restatement (oracle/smpl.py).

Restates the optimisation body of `multi_view_optimization` (`PR/player_recon.py:1691-1979`) statement by statement on
real `torch.autograd` + `torch.optim.Adam`:

  :1691-1694  body_pose_mean / betas_mean = mean over the views of the single-view results
  :1706-1708  best_pose = body_pose_mean[:, {0-5, 8-20}]
  :1720       for i in range(3):
  :1724-1731    body_poses_with_hand_feet = cat(...)           <- built ONCE per round, before requires_grad is set
  :1734-1742    block 1: Adam([cam_wp_mult, global_orient_mult]) -- the STACKED per-view tensors
  :1744-1838      per epoch: train pass (one loss.backward() + optimiser.step() per view), validation pass,
                  tracker.update_per_epoch(), determine_save_model_weights_this_epoch -> best snapshot
  :1843-1845    cam_wp_mult = best_cam_wp_mult; global_orient_mult = best_global_orient_mult
  :1862-1871    block 2: Adam([body_poses_without_hands_feet, betas_mean])
  :1873-1951      same epoch structure; the forward still uses the tensor concatenated at :1724-1731, so
                  `body_poses_without_hands_feet.grad` stays None and Adam skips it: only betas move (reference
                  behaviour, `stale_pose_cat=True`); `stale_pose_cat=False` rebuilds the posed tensor from the leaf
                  every step, as the single-view loop does (:1202-1206)
  :1954-1957    betas_mean = best_betas_mean; body_poses_without_hands_feet = best_pose
  :1976-1979    saved: body_pose / betas of the record, translation / global_orient of every view

By default the views are visited in the order 0..V-1 and the record is decided by joints2D_l2es alone; `view_orders` takes the
orders the reference draws with `random.shuffle` before every pass, `silhouette_metric` adds the second tracked metric of its
record rule.  If a pose block never reaches the record the reference reads
`best_betas_mean` before assignment in round 0 (NameError) and a stale copy later; here the block's parameters return
to their values at the block's start, which is what the stale copy holds from round 1 on."""
import numpy as np
import torch
from torch import optim

from .fit import _project17
from .geometry import joints2d_l2e, joints2d_loss, translation_to_wp, wp_to_translation


def fit_multi_view(smpl, body_pose_views, global_orient_views, betas_views, translation_views, joints2D_views, lr=1e-3,
                   iters=50, rounds=3, stale_pose_cat=True, record=None, view_orders=None, silhouette_metric=None):
    """ONE player.  body_pose_views [V,1,23,3,3], global_orient_views [V,1,1,3,3], betas_views [V,1,10],
    translation_views [V,1,3] (torch fp32), joints2D_views [V,17,>=2] numpy.  Returns dict(body_pose [1,23,3,3],
    betas [1,10], translation [V,1,3], global_orient [V,1,1,3,3], cam_wp [V,1,3], best_val, val_history)."""
    V = body_pose_views.shape[0]
    targets = [torch.from_numpy(np.asarray(joints2D_views[j])[:, :2]).float() for j in range(V)]
    translation_mult = [translation_views[j].detach().clone() for j in range(V)]
    cam_wp_mult = torch.stack([translation_to_wp(translation_views[j]) for j in range(V)]).detach().clone()      # [V,1,3]
    global_orient_mult = torch.stack([global_orient_views[j] for j in range(V)]).detach().clone()                # [V,1,1,3,3]
    body_pose_mean = torch.stack([body_pose_views[j] for j in range(V)]).mean(axis=0).detach().clone()           # [1,23,3,3]
    betas_mean = torch.stack([betas_views[j] for j in range(V)]).mean(axis=0).detach().clone()                   # [1,10]

    best_pose = torch.cat([body_pose_mean[:, :6], body_pose_mean[:, 8:21]], dim=1)
    best_val, best_sil = np.inf, np.inf
    sil_history = []
    best_model_wts = [body_pose_mean.detach().numpy().copy(), betas_mean.detach().numpy().copy()]
    best_model_view = [[translation_mult[j].detach().numpy().copy(), global_orient_mult[j].detach().numpy().copy()] for j in range(V)]
    best_cam_view = [cam_wp_mult[j].detach().numpy().copy() for j in range(V)]
    history = []
    order = list(range(V))

    def assemble(without):
        hf = torch.cat([body_pose_mean[:, 6:8], body_pose_mean[:, 21:]], dim=1)
        return torch.cat([without[:, :6], hf[:, 0:2], without[:, 6:19], hf[:, 2:]], dim=1)

    for i in range(rounds):
        body_poses_without_hands_feet = best_pose
        body_poses_with_hand_feet = assemble(body_poses_without_hands_feet)      # :1724-1731 (constant within the round)

        # ---------------------------------------------------------------- block 1: cameras + orientations
        betas_mean.requires_grad = False
        body_poses_without_hands_feet.requires_grad = False
        global_orient_mult.requires_grad = True
        cam_wp_mult.requires_grad = True
        optimiser = optim.Adam([cam_wp_mult, global_orient_mult], lr=lr)
        best_cam_wp_mult, best_global_orient_mult = cam_wp_mult.clone().detach(), global_orient_mult.clone().detach()
        for epoch in range(iters):
            val, silv = 0.0, 0.0
            for is_train in (True, False):
                for index in (view_orders[(i * 2 + 0) * iters + epoch] if (view_orders is not None and is_train) else order):
                    index = int(index)
                    out = smpl(body_pose=body_poses_with_hand_feet, global_orient=global_orient_mult[index], betas=betas_mean,
                               pose2rot=False)
                    pred = _project17(out, cam_wp_mult[index])
                    if is_train:
                        optimiser.zero_grad()
                        loss = joints2d_loss(pred[0], targets[index])
                        loss.backward()
                        optimiser.step()
                        translation_mult[index] = wp_to_translation(cam_wp_mult[index]).detach()
                    else:
                        val += joints2d_l2e(pred[0].detach().numpy(), targets[index].numpy())
                        if silhouette_metric is not None:
                            silv += silhouette_metric(index, out.vertices[0].detach().numpy(), cam_wp_mult[index][0].detach().numpy())
            val /= V
            silv /= V
            history.append(val)
            sil_history.append(silv)
            if not (val > best_val) and not (silhouette_metric is not None and silv > best_sil):
                best_val, best_sil = val, silv
                best_model_wts = [body_poses_with_hand_feet.detach().numpy().copy(), betas_mean.detach().numpy().copy()]
                for j in range(V):
                    best_model_view[j] = [translation_mult[j].detach().numpy().copy(), global_orient_mult[j].detach().numpy().copy()]
                    best_cam_view[j] = cam_wp_mult[j].detach().numpy().copy()
                best_global_orient_mult = global_orient_mult.clone().detach()
                best_cam_wp_mult = cam_wp_mult.clone().detach()
            if record is not None:
                record.setdefault('cam', []).append(cam_wp_mult.detach().numpy().copy())
                record.setdefault('G', []).append(global_orient_mult.detach().numpy().copy())
        cam_wp_mult = best_cam_wp_mult
        global_orient_mult = best_global_orient_mult
        for j in range(V):
            translation_mult[j] = wp_to_translation(cam_wp_mult[j]).detach()

        # ---------------------------------------------------------------- block 2: shared pose (see header) + betas
        betas_mean.requires_grad = True
        body_poses_without_hands_feet = body_poses_without_hands_feet.detach().clone()
        body_poses_without_hands_feet.requires_grad = True
        global_orient_mult.requires_grad = False
        cam_wp_mult.requires_grad = False
        optimiser_pose = optim.Adam([body_poses_without_hands_feet, betas_mean], lr=lr)
        best_betas_mean = betas_mean.clone().detach()
        best_pose = body_poses_without_hands_feet.clone().detach()
        for epoch in range(iters):
            val, silv = 0.0, 0.0
            for is_train in (True, False):
                for index in (view_orders[(i * 2 + 1) * iters + epoch] if (view_orders is not None and is_train) else order):
                    index = int(index)
                    posed = body_poses_with_hand_feet if stale_pose_cat else assemble(body_poses_without_hands_feet)
                    out = smpl(body_pose=posed, global_orient=global_orient_mult[index], betas=betas_mean, pose2rot=False)
                    pred = _project17(out, cam_wp_mult[index])
                    if is_train:
                        optimiser_pose.zero_grad()
                        loss = joints2d_loss(pred[0], targets[index])
                        loss.backward()
                        optimiser_pose.step()
                    else:
                        val += joints2d_l2e(pred[0].detach().numpy(), targets[index].numpy())
                        if silhouette_metric is not None:
                            silv += silhouette_metric(index, out.vertices[0].detach().numpy(), cam_wp_mult[index][0].detach().numpy())
            val /= V
            silv /= V
            history.append(val)
            sil_history.append(silv)
            if not (val > best_val) and not (silhouette_metric is not None and silv > best_sil):
                best_val, best_sil = val, silv
                posed = body_poses_with_hand_feet if stale_pose_cat else assemble(body_poses_without_hands_feet)
                best_model_wts = [posed.detach().numpy().copy(), betas_mean.detach().numpy().copy()]
                for j in range(V):
                    best_model_view[j] = [translation_mult[j].detach().numpy().copy(), global_orient_mult[j].detach().numpy().copy()]
                    best_cam_view[j] = cam_wp_mult[j].detach().numpy().copy()
                best_pose = body_poses_without_hands_feet.clone().detach()
                best_betas_mean = betas_mean.clone().detach()
            if record is not None:
                record.setdefault('betas', []).append(betas_mean.detach().numpy().copy())
        betas_mean = best_betas_mean
    return {'body_pose': best_model_wts[0], 'betas': best_model_wts[1],
            'translation': np.stack([b[0] for b in best_model_view]), 'global_orient': np.stack([b[1] for b in best_model_view]),
            'cam_wp': np.stack(best_cam_view), 'best_val': best_val, 'val_history': np.asarray(history, np.float32),
            'best_sil': best_sil, 'sil_history': np.asarray(sil_history, np.float64)}
