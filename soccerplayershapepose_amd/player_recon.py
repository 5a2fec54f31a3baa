"""The per-player fit loops and the batched inference pipeline, re-designed for gfx950.

Mirrors (same names, argument meaning, in-place contract):
  optimize_camera           PR/PyTorch3DTest.py:3055-3124   cam + global-orientation fit, 50 Adam steps, lr 0.01
  single_view_optimization  PR/player_recon.py:1172-1294    pose + shape + camera fit ("theta fit"), lr 1e-3
  evaluate_model_2d core    PR/PyTorch3DTest.py:3738-3776   regressor -> rot6d -> per-player fit -> SMPL -> 2-D verts
The reference runs these one player at a time with a full SMPL.forward + autograd per iteration; here a
whole batch of players is ONE kernel launch (csrc/fit.hip), one wavefront per player."""
import ctypes as C
import os
from types import SimpleNamespace

import torch

from . import _abi, config
from . import runtime as rt
from .utils.cam_utils import (convert_weak_perspective_to_camera_translation_torch,
                              orthographic_project_to_pixels)
from .utils.rigid_transform_utils import rot6d_to_rotmat


def _fit_config(r, variant, iters, lr, param_mask, select_mode):
    cfg = _abi.FitConfig()
    r.lib.spsp_fit_config_default(C.byref(cfg), variant)
    if iters is not None:
        cfg.iters = int(iters)
    if lr is not None:
        cfg.lr = float(lr)
    if param_mask is not None:
        cfg.param_mask = int(param_mask)
    if select_mode is not None:
        cfg.select_mode = int(select_mode)
    return cfg


def fit_players(smpl, full_pose, betas, cam_wp, joints2D, *, variant, iters=None, lr=None, param_mask=None,
                select_mode=None, return_history=False, return_trajectory=False):
    """Batched fit, IN PLACE on contiguous CUDA fp32 tensors full_pose [B,24,3,3], betas [B,10], cam_wp [B,3].
    joints2D [B,17,>=2] detected joints in 512-px space.  return_history -> (loss [B,iters], l2e [B,iters]);
    return_trajectory -> additionally params [B,iters,229], the record [cam | R | betas] every iteration evaluated."""
    for name, t, shp in (('full_pose', full_pose, (24, 3, 3)), ('betas', betas, (10,)), ('cam_wp', cam_wp, (3,))):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                and tuple(t.shape[1:]) == shp):
            raise ValueError(f'fit_players: {name} must be a contiguous CUDA fp32 tensor [B,{",".join(map(str, shp))}]')
    B = full_pose.shape[0]
    r = rt.get(full_pose.device)
    j2d = rt.dev_f32(torch.as_tensor(joints2D)[..., :2], r.device).reshape(B, 17, 2).contiguous()
    cfg = _fit_config(r, variant, iters, lr, param_mask, select_mode)
    loss = l2e = traj = None
    if return_history or return_trajectory:
        loss = torch.empty((B, cfg.iters), dtype=torch.float32, device=r.device)
        l2e = torch.empty((B, cfg.iters), dtype=torch.float32, device=r.device)
    if return_trajectory:
        traj = torch.empty((B, cfg.iters, _abi.RECORD_FLOATS), dtype=torch.float32, device=r.device)
    smpl.to(r.device)
    r.check(r.lib.spsp_fit_trajectory(r.ctx, smpl.handle, C.byref(cfg), rt.ptr(full_pose), rt.ptr(betas), rt.ptr(cam_wp),
                                      rt.ptr(j2d), B, rt.ptr(loss), rt.ptr(l2e), rt.ptr(traj), r.stream()), 'spsp_fit')
    if return_trajectory:
        return loss, l2e, traj
    return (loss, l2e) if return_history else None


def optimize_camera(body_pose, global_orient, betas, joints2D, silhouette, cam_wp, smpl=None, lr=None, iters=None):
    """`optimize_camera(body_pose, global_orient, betas, joints2D, silhouette, cam_wp) -> None`
    (`PR/PyTorch3DTest.py:3055`): fits cam_wp [1,3] and global_orient [1,1,3,3] IN PLACE (the arguments may be
    views into batch tensors, as at :3766-3767); `silhouette` is accepted and ignored, as its render never enters
    the reference loss.  `smpl` replaces the per-call `SMPL(config.SMPL_MODEL_DIR)` construction (:3057)."""
    if smpl is None:
        raise ValueError('optimize_camera: pass smpl= (the reference re-reads the SMPL pickle per player, :3057)')
    dev = global_orient.device
    B = global_orient.shape[0]
    full = torch.cat([global_orient.reshape(B, 1, 3, 3), body_pose.reshape(B, 23, 3, 3)], dim=1).float().contiguous()
    b = betas.detach().float().contiguous().clone()
    cam = cam_wp.detach().float().contiguous().clone()
    j = torch.as_tensor(joints2D)
    fit_players(smpl, full, b, cam, j.reshape(B, 17, -1), variant=0, iters=iters, lr=lr)
    with torch.no_grad():
        global_orient.copy_(full[:, :1].reshape(global_orient.shape).to(dev))
        cam_wp.copy_(cam.to(dev))
    return None


FIT_SELECT_REFERENCE = 'reference'     # the reference's own rule: joints2D_l2es AND the rendered-silhouette metric (needs silhouette=)


def single_view_optimization(smpl, body_pose, global_orient, betas, pred_cam_wp, joints2D, interation=None, lr=None,
                             select_mode=_abi.FIT_SELECT_LAST, silhouette=None, renderer=None):
    """Fit body of `single_view_optimization` (`PR/player_recon.py:1172-1294`) for a batch of players.
    Returns the record the reference saves (:1293-1294) -- body_pose [B,23,3,3], global_orient [B,1,3,3],
    betas [B,10], translation [B,3] -- plus cam_wp.  Inputs are not modified.

    select_mode: FIT_SELECT_LAST (what `optimize_camera` leaves), FIT_SELECT_BEST_L2E (lowest joints2D_l2es), or
    FIT_SELECT_REFERENCE -- the rule the reference applies (`:1259-1271`): an iterate is kept iff its joints2D_l2es AND
    the "silhouette_iou" metric of its rendered silhouette against the detected `silhouette` [B,512,512] are both <= their
    values at the previous best.  The fit trajectory is recorded, every iterate's mesh is rendered (`renderer.
    SilhouetteRenderer`, the restated neural_renderer forward pass) and `spsp_fit_select` applies the rule; the result
    also carries best_iter [B] and the two metric histories.  Unlike the reference (:1264, which saves the body pose the
    REGRESSOR produced: the optimised tensor is a copy) the optimised body pose of the chosen iterate is returned."""
    B = betas.shape[0]
    full = torch.cat([rt.dev_f32(global_orient).reshape(B, 1, 3, 3), rt.dev_f32(body_pose).reshape(B, 23, 3, 3)], dim=1).contiguous()
    b = rt.dev_f32(betas).clone()
    cam = rt.dev_f32(pred_cam_wp).clone()
    # (the reference's default: `interation=player_recon_single_view_iteration` = 100, PR/player_recon.py:1051, S/global_var.py:75)
    iters = interation if interation is not None else config.player_recon_single_view_iteration
    extra = {}
    if select_mode == FIT_SELECT_REFERENCE:
        if silhouette is None or renderer is None:
            raise ValueError('single_view_optimization: select_mode "reference" needs silhouette= [B,512,512] and renderer=')
        r = rt.get(full.device)
        loss, l2e, traj = fit_players(smpl, full, b, cam, joints2D, variant=1, iters=iters, lr=lr,
                                      select_mode=_abi.FIT_SELECT_LAST, return_trajectory=True)
        tgt = torch.as_tensor(silhouette).to(device=r.device, dtype=torch.uint8).reshape(B, config.proxy_rep_input_wh,
                                                                                          config.proxy_rep_input_wh).contiguous()
        sil = torch.empty((B, iters), dtype=torch.float32, device=r.device)
        for it in range(iters):                                      # every iterate: SMPL forward -> render -> metric
            rec = traj[:, it]
            R_it = rec[:, 3:219].reshape(B, 24, 3, 3).contiguous()
            verts = smpl(full_pose=R_it, betas=rec[:, 219:].contiguous(), return_joints=False).vertices
            t_it = convert_weak_perspective_to_camera_translation_torch(rec[:, :3].contiguous(), config.FOCAL_LENGTH, config.proxy_rep_input_wh)
            sil[:, it] = renderer(verts, t_it, target=tgt, return_image=False).metric
        best_iter = torch.empty((B,), dtype=torch.int32, device=r.device)
        r.check(r.lib.spsp_fit_select(r.ctx, rt.ptr(l2e), rt.ptr(sil), iters, B, rt.ptr(traj), rt.ptr(full), rt.ptr(b), rt.ptr(cam),
                                      rt.ptr(best_iter), r.stream()), 'spsp_fit_select')
        extra = dict(best_iter=best_iter, l2e_history=l2e, silhouette_metric_history=sil)
    else:
        fit_players(smpl, full, b, cam, joints2D, variant=1, iters=iters, lr=lr, select_mode=select_mode)
    translation = convert_weak_perspective_to_camera_translation_torch(cam, config.FOCAL_LENGTH, config.proxy_rep_input_wh)
    return SimpleNamespace(body_pose=full[:, 1:], global_orient=full[:, :1], betas=b, translation=translation, cam_wp=cam, **extra)


def broad_view_optimization(smpl, body_pose, global_orient, betas, pred_cam_wp, joints2D, interation=None, lr=None,
                            select_mode=_abi.FIT_SELECT_BEST_L2E, silhouette=None, renderer=None):
    """Fit body of `broad_view_optimization` (`PR/player_recon.py:2128-2229`): the shared body pose and betas (the multi-view
    result) stay fixed; the broadcast view's camera and global orientation (initialised by the regressor on that view) are
    fitted with `Adam([pred_cam_wp, global_orient], lr=1e-2)` for 100 epochs (`S/global_var.py:83-84`).  Unlike the single-view
    loop every epoch has a TRAIN pass (loss, backward, step) and then a VALIDATION pass on the stepped parameters; the epoch
    whose validation metrics are the best so far is saved (`:2207-2221`) -- so the candidates are the parameters AFTER steps
    1..N, not the ones each step started from.
    select_mode: FIT_SELECT_BEST_L2E (validation joints2D_l2es alone), FIT_SELECT_REFERENCE (the reference's rule over both
    tracked metrics, joints2D_l2es AND the rendered-silhouette metric: needs silhouette= and renderer=), FIT_SELECT_LAST.
    Returns the record the reference saves (`:2207-2211`): body_pose, global_orient [B,1,3,3], betas, translation, plus cam_wp and
    best_epoch [B] (1-based).  Inputs are not modified."""
    B = betas.shape[0]
    full = torch.cat([rt.dev_f32(global_orient).reshape(B, 1, 3, 3), rt.dev_f32(body_pose).reshape(B, 23, 3, 3)], dim=1).contiguous()
    r = rt.get(full.device)
    b = rt.dev_f32(betas, r.device).clone()
    cam = rt.dev_f32(pred_cam_wp, r.device).clone()
    iters = int(interation if interation is not None else config.player_recon_broad_view_iteration)
    lr = float(lr if lr is not None else config.player_recon_broad_view_learning_rate)
    # iters + 1 evaluation points: point i >= 1 = the parameters after step i = what epoch i's validation pass sees
    loss, l2e, traj = fit_players(smpl, full, b, cam, joints2D, variant=0, iters=iters + 1, lr=lr, select_mode=_abi.FIT_SELECT_LAST,
                                  return_trajectory=True)
    l2e_val, traj_val = l2e[:, 1:].contiguous(), traj[:, 1:].contiguous()
    best_iter = torch.full((B,), iters - 1, dtype=torch.int32, device=r.device)
    sil = None
    if select_mode == FIT_SELECT_REFERENCE:
        if silhouette is None or renderer is None:
            raise ValueError('broad_view_optimization: select_mode "reference" needs silhouette= [B,512,512] and renderer=')
        wh = config.proxy_rep_input_wh
        tgt = torch.as_tensor(silhouette).to(device=r.device, dtype=torch.uint8).reshape(B, wh, wh).contiguous()
        sil = torch.empty((B, iters), dtype=torch.float32, device=r.device)
        for it in range(iters):
            rec = traj_val[:, it]
            verts = smpl(full_pose=rec[:, 3:219].reshape(B, 24, 3, 3).contiguous(), betas=rec[:, 219:].contiguous(), return_joints=False).vertices
            t_it = convert_weak_perspective_to_camera_translation_torch(rec[:, :3].contiguous(), config.FOCAL_LENGTH, wh)
            sil[:, it] = renderer(verts, t_it, target=tgt, return_image=False).metric
    if select_mode == _abi.FIT_SELECT_LAST:
        rec = traj_val[:, -1]
        full, b, cam = rec[:, 3:219].reshape(B, 24, 3, 3).contiguous(), rec[:, 219:].contiguous(), rec[:, :3].contiguous()
    else:
        r.check(r.lib.spsp_fit_select(r.ctx, rt.ptr(l2e_val), rt.ptr(sil), iters, B, rt.ptr(traj_val), rt.ptr(full), rt.ptr(b), rt.ptr(cam),
                                      rt.ptr(best_iter), r.stream()), 'spsp_fit_select')
    translation = convert_weak_perspective_to_camera_translation_torch(cam, config.FOCAL_LENGTH, config.proxy_rep_input_wh)
    return SimpleNamespace(body_pose=full[:, 1:], global_orient=full[:, :1], betas=b, translation=translation, cam_wp=cam,
                           best_epoch=best_iter + 1, val_l2e_history=l2e_val, val_silhouette_metric_history=sil)


def single_view_optimization_test(smpl, renderer, body_pose, global_orient, betas, pred_cam_wp, joints2D, silhouette, interation=50,
                                  lr=None, joints2D_weight=1.0, silhouette_weight=1e6, return_history=False):
    """Fit body of `single_view_optimization_test` (`PR/player_recon.py:1429-1557`) for a batch of players: the silhouette is IN
    the loss (`init_loss_and_metric(True, True)`, weights 1.0 / 1e6, `:851`) next to the joints2D term, both log-variances of
    the criterion are optimiser parameters (`:1453`), 50 epochs of Adam (lr 1e-3), the last iterate is left behind; the
    gradient of the rendered silhouette comes from the restated neural_renderer backward pass (csrc/silhouette_grad.hip,
    parity unpinned).  `renderer`: a `renderer.SilhouetteRenderer`; `silhouette` [B,512,512] (or one [512,512]) 0/1.
    Returns body_pose, global_orient, betas, translation, cam_wp, log_vars [B,2]; with return_history also loss / joints2D_l2es /
    silhouette-metric histories [B,iters], param_history [B,iters,231] ([cam | R | betas | log_vars] each epoch evaluated) and
    grad_history (the loss gradient there)."""
    import math
    B = betas.shape[0]
    full = torch.cat([rt.dev_f32(global_orient).reshape(B, 1, 3, 3), rt.dev_f32(body_pose).reshape(B, 23, 3, 3)], dim=1).contiguous()
    r = rt.get(full.device)
    b = rt.dev_f32(betas, r.device).clone()
    cam = rt.dev_f32(pred_cam_wp, r.device).clone()
    smpl.to(r.device)
    renderer.to(r.device)
    if B > smpl.max_batch or B > renderer.max_batch:
        raise ValueError(f'single_view_optimization_test: batch {B} exceeds max_batch of the SMPL / renderer handle')
    cfg = _abi.FitSilhouetteConfig()
    r.lib.spsp_fit_silhouette_config_default(C.byref(cfg))
    cfg.iters = int(interation)
    if lr is not None:
        cfg.lr = float(lr)
    cfg.joints2d_weight, cfg.silhouette_weight = float(joints2D_weight), float(silhouette_weight)
    lv = torch.tensor([-math.log(joints2D_weight + 1e-6), -math.log(silhouette_weight + 1e-6)], dtype=torch.float32,
                      device=r.device).repeat(B, 1).contiguous()           # multi_task_loss.py:37-46
    j2d = rt.dev_f32(torch.as_tensor(joints2D)[..., :2], r.device).reshape(B, 17, 2).contiguous()
    wh = config.proxy_rep_input_wh
    tgt = torch.as_tensor(silhouette).to(device=r.device, dtype=torch.uint8).contiguous()
    shared = 1 if tuple(tgt.shape) == (wh, wh) else 0
    if not shared and tuple(tgt.shape) != (B, wh, wh):
        raise ValueError(f'single_view_optimization_test: silhouette must be [{wh},{wh}] or [B,{wh},{wh}], got {tuple(tgt.shape)}')
    hist = [None] * 5
    if return_history:
        hist = [torch.empty((B, cfg.iters), dtype=torch.float32, device=r.device) for _ in range(3)]
        hist += [torch.empty((B, cfg.iters, _abi.RECORD_FLOATS + 2), dtype=torch.float32, device=r.device) for _ in range(2)]
    r.check(r.lib.spsp_fit_silhouette(r.ctx, smpl.handle, renderer.handle, C.byref(cfg), rt.ptr(full), rt.ptr(b), rt.ptr(cam), rt.ptr(lv),
                                      rt.ptr(j2d), rt.ptr(tgt), shared, B, rt.ptr(hist[0]), rt.ptr(hist[1]), rt.ptr(hist[2]), rt.ptr(hist[3]),
                                      rt.ptr(hist[4]), r.stream()), 'spsp_fit_silhouette')
    translation = convert_weak_perspective_to_camera_translation_torch(cam, config.FOCAL_LENGTH, wh)
    out = SimpleNamespace(body_pose=full[:, 1:], global_orient=full[:, :1], betas=b, translation=translation, cam_wp=cam, log_vars=lv)
    if return_history:
        out.loss_history, out.l2e_history, out.silhouette_metric_history, out.param_history, out.grad_history = hist
    return out


def multi_view_orders(V, interation, rounds, seed):
    """The view orders of a `multi_view_optimization` run as the reference draws them: `indexes = list(range(V));
    random.shuffle(indexes)` before EVERY pass -- train, then validation -- of every epoch of both blocks of every round
    (`PR/player_recon.py:1746-1748, 1874-1876`), from Python's `random` seeded with `seed`.  Returns the TRAIN orders
    [rounds * 2 * interation, V] int32 (the validation pass only sums its metrics: its order does not matter)."""
    import random
    rng = random.Random(seed)
    out = []
    for _ in range(rounds * 2 * interation):
        idx = list(range(V))
        rng.shuffle(idx)
        out.append(idx)
        val = list(range(V))
        rng.shuffle(val)                       # the draw the validation pass consumes
    return torch.tensor(out, dtype=torch.int32)


def multi_view_optimization(smpl, body_pose, global_orient, betas, translation, joints2D, interation=None, lr=None,
                            rounds=3, optimise_body_pose=False, return_history=False, view_order=None,
                            select_mode=_abi.FIT_SELECT_BEST_L2E, silhouettes=None, renderer=None):
    """Optimisation body of `multi_view_optimization` (`PR/player_recon.py:1691-1979`) for a batch of players seen in
    V <= 4 views each.  Inputs are the per-view single-view results the reference loads from its npz records
    (`:1673-1689`; `records.load_record`): body_pose [B,V,23,3,3], global_orient [B,V,1,3,3], betas [B,V,10],
    translation [B,V,3], and the detected joints2D [B,V,17,>=2] (512-px space).

    As in the reference the shared body pose and betas start from the MEAN over the views (`:1691-1694`), every view
    keeps its own camera (recovered from its translation, `:1689`) and orientation, and `rounds` x {camera block, pose
    block} of `interation` epochs run with the best iterate kept (csrc/fit_mv_core.hpp).
    optimise_body_pose=False reproduces the reference, whose pose block only ever moves betas (the posed tensor is
    concatenated before requires_grad is set, `:1724-1731`); True optimises body_pose idx {0-5, 8-20} as intended.
    view_order: [rounds * 2 * interation, V] -- the view each train step of each epoch visits (`multi_view_orders` draws
    them as the reference's `random.shuffle` does); None = 0 .. V-1.
    select_mode: FIT_SELECT_BEST_L2E -- the record is the epoch with the lowest mean joints2D_l2es, everything in ONE launch;
    FIT_SELECT_REFERENCE -- the reference's rule (`:1820, :1937`: joints2D_l2es AND the mean rendered-silhouette metric both <=
    their values at the previous record; needs silhouettes= [B,V,512,512] and renderer=): one launch per block, every epoch's
    V meshes rendered, the record and the block's best epoch decided on both metrics.

    Returns what the reference saves (`:1976-1979`): body_pose [B,23,3,3], betas [B,10] (-> data.npz) and per view
    translation [B,V,3], global_orient [B,V,1,3,3] (-> <view>.npz), plus cam_wp [B,V,3]."""
    from .utils.cam_utils import convert_camera_translation_to_weak_perspective_torch
    bp = rt.dev_f32(body_pose)
    r = rt.get(bp.device)
    if bp.dim() != 5 or tuple(bp.shape[2:]) != (23, 3, 3):
        raise ValueError(f'multi_view_optimization: body_pose must be [B,V,23,3,3], got {tuple(bp.shape)}')
    B, V = bp.shape[:2]
    if not 1 <= V <= 4:
        raise ValueError(f'multi_view_optimization: {V} views (the kernel holds 1..4)')
    go = rt.dev_f32(global_orient, r.device).reshape(B, V, 9).contiguous().clone()
    be = rt.dev_f32(betas, r.device).reshape(B, V, 10)
    tr = rt.dev_f32(translation, r.device).reshape(B, V, 3)
    j2d = rt.dev_f32(torch.as_tensor(joints2D)[..., :2], r.device).reshape(B, V, 17, 2).contiguous()
    cam = convert_camera_translation_to_weak_perspective_torch(tr.reshape(B * V, 3), config.FOCAL_LENGTH,
                                                               config.proxy_rep_input_wh).reshape(B, V, 3).contiguous()
    body = bp.mean(dim=1).reshape(B, 207).contiguous()           # body_pose_mean, betas_mean (:1691-1694)
    bmean = be.mean(dim=1).contiguous()
    cfg = _abi.FitMultiviewConfig()
    r.lib.spsp_fit_multiview_config_default(C.byref(cfg))
    cfg.iters = int(interation if interation is not None else config.player_recon_multi_view_iteration)
    cfg.rounds = int(rounds)
    if lr is not None:
        cfg.lr = float(lr)
    cfg.optimise_body_pose = 1 if optimise_body_pose else 0
    order = None
    if view_order is not None:
        order = torch.as_tensor(view_order).to(device=r.device, dtype=torch.int32).contiguous()
        if tuple(order.shape) != (cfg.rounds * 2 * cfg.iters, V) or not bool((order.sort(dim=1).values.cpu() == torch.arange(V)).all()):
            raise ValueError(f'multi_view_optimization: view_order must hold {cfg.rounds * 2 * cfg.iters} permutations of 0..{V - 1}')
    smpl.to(r.device)
    extra = {}
    if select_mode == FIT_SELECT_REFERENCE:
        if silhouettes is None or renderer is None:
            raise ValueError('multi_view_optimization: select_mode "reference" needs silhouettes= [B,V,512,512] and renderer=')
        extra = _multi_view_reference_rule(r, smpl, renderer, cfg, B, V, body, bmean, cam, go, j2d, order, silhouettes)
        hist = extra.pop('val_l2e')
    else:
        hist = torch.empty((B, cfg.rounds * 2 * cfg.iters), dtype=torch.float32, device=r.device) if return_history else None
        r.check(r.lib.spsp_fit_multiview_ordered(r.ctx, smpl.handle, C.byref(cfg), B, V, rt.ptr(body), rt.ptr(bmean), rt.ptr(cam),
                                                 rt.ptr(go), rt.ptr(j2d), rt.ptr(order), rt.ptr(hist), r.stream()), 'spsp_fit_multiview')
    trans = convert_weak_perspective_to_camera_translation_torch(cam.reshape(B * V, 3), config.FOCAL_LENGTH,
                                                                 config.proxy_rep_input_wh).reshape(B, V, 3)
    out = SimpleNamespace(body_pose=body.reshape(B, 23, 3, 3), betas=bmean, translation=trans,
                          global_orient=go.reshape(B, V, 1, 3, 3), cam_wp=cam, **extra)
    if return_history or select_mode == FIT_SELECT_REFERENCE:
        out.val_l2e_history = hist.reshape(B, cfg.rounds, 2, cfg.iters)
    return out


def _multi_view_reference_rule(r, smpl, renderer, cfg, B, V, body, betas, cam, go, j2d, order, silhouettes):
    """The multi-view fit under the reference's record rule over BOTH tracked metrics (in place on body / betas / cam / go =
    the record).  One `spsp_fit_multiview_block` launch per block (the optimisation inside a block does not depend on the
    record), every epoch's V meshes rendered for the silhouette metric, then the sequential rule over the block's epochs with the
    record's metrics carried along; the next block starts from the block's best epoch (`:1843-1845, :1954-1957`) -- from the
    block's start if none of its epochs made the record."""
    wh, it = config.proxy_rep_input_wh, cfg.iters
    tgt = torch.as_tensor(silhouettes).to(device=r.device, dtype=torch.uint8).reshape(B, V, wh, wh).contiguous()
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=r.device)
    st_body, st_betas, st_cam, st_go = body.clone(), betas.clone(), cam.clone(), go.clone()       # current state of the loop
    best_l2e = torch.full((B,), float('inf'))
    best_sil = torch.full((B,), float('inf'))
    l2e_all, sil_all = [], []
    for rd in range(cfg.rounds):
        for block in (0, 1):
            w_body, w_betas, w_cam, w_go = st_body.clone(), st_betas.clone(), st_cam.clone(), st_go.clone()
            val, sv, ss = f32(B, it), f32(B, it, V, 12), f32(B, it, 217)
            ordb = order[(rd * 2 + block) * it:(rd * 2 + block + 1) * it].contiguous() if order is not None else None
            r.check(r.lib.spsp_fit_multiview_block(r.ctx, smpl.handle, C.byref(cfg), block, B, V, rt.ptr(w_body), rt.ptr(w_betas), rt.ptr(w_cam),
                                                   rt.ptr(w_go), rt.ptr(j2d), rt.ptr(ordb), rt.ptr(val), rt.ptr(sv), rt.ptr(ss), r.stream()),
                    'spsp_fit_multiview_block')
            sil = torch.zeros((B, it), dtype=torch.float32, device=r.device)
            for ep in range(it):                                  # every epoch: the V meshes its validation pass saw
                for v in range(V):
                    full = torch.cat([sv[:, ep, v, 3:].reshape(B, 1, 3, 3), ss[:, ep, :207].reshape(B, 23, 3, 3)], dim=1).contiguous()
                    verts = smpl(full_pose=full, betas=ss[:, ep, 207:].contiguous(), return_joints=False).vertices
                    t_v = convert_weak_perspective_to_camera_translation_torch(sv[:, ep, v, :3].contiguous(), config.FOCAL_LENGTH, wh)
                    sil[:, ep] += renderer(verts, t_v, target=tgt[:, v].contiguous(), return_image=False).metric
            sil /= V
            vl, sl = val.cpu(), sil.cpu()
            l2e_all.append(val)
            sil_all.append(sil)
            blk_best = torch.full((B,), -1, dtype=torch.long)
            for b in range(B):                                    # determine_save_model_weights_this_epoch, epoch by epoch
                for ep in range(it):
                    if not (vl[b, ep] > best_l2e[b]) and not (sl[b, ep] > best_sil[b]):
                        best_l2e[b], best_sil[b], blk_best[b] = vl[b, ep], sl[b, ep], ep
            for b in range(B):
                e = int(blk_best[b])
                if e < 0:
                    continue                                      # no record in this block: its parameters return to the block's start
                # the record = the whole state of that epoch; the block's parameter group continues from there
                body[b], betas[b] = ss[b, e, :207], ss[b, e, 207:]
                cam[b], go[b] = sv[b, e, :, :3], sv[b, e, :, 3:]
                if block == 0:
                    st_cam[b], st_go[b] = sv[b, e, :, :3], sv[b, e, :, 3:]
                else:
                    st_body[b], st_betas[b] = ss[b, e, :207], ss[b, e, 207:]
    return dict(val_l2e=torch.cat(l2e_all, dim=1), val_silhouette_metric_history=torch.cat(sil_all, dim=1).reshape(B, cfg.rounds, 2, it),
                record_l2e=best_l2e, record_silhouette_metric=best_sil)


def reconstruct_players(regressor, smpl, proxy_rep, joints2D, *, fit='pose_shape', iters=50, lr=None,
                        select_mode=_abi.FIT_SELECT_LAST, return_vertices2d=False, pose_relation=None, boxes=None,
                        return_record=False):
    """The hot path end to end for one batch (`evaluate_model_2d` core, `PR/PyTorch3DTest.py:3738-3776`):
        proxy_rep [B,18,256,256] -> regressor -> rot6d -> fit ('camera' = optimize_camera per player,
        'pose_shape' = single_view_optimization per player, None = skip) -> SMPL forward.
    `pose_relation` (a `models.regressor_relate.PoseRelationModule`) + `boxes` insert the reference's optional
    cross-player refinement between rot6d and the fit.
    Returns cam [B,3], rotmats [B,24,3,3], betas [B,10], vertices [B,6890,3], joints [B,90,3], translation; with
    `return_record` also record [B,229] = [cam | R | betas], the all-gather payload of the multi-GPU path."""
    x, fmt = regressor.prepare_input(proxy_rep)
    r = rt.get(x.device)
    B = x.shape[0]
    smpl.to(r.device)
    if B > smpl.max_batch:
        raise ValueError(f'reconstruct_players: batch {B} exceeds the SMPL handle\'s max_batch {smpl.max_batch}')
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=r.device)
    cam, pose6d, betas, R = f32(B, 3), f32(B, 144), f32(B, 10), f32(B, 24, 3, 3)
    vertices, joints, translation = f32(B, 6890, 3), f32(B, 90, 3), f32(B, 3)
    out = _abi.PipelineOut(cam=rt.ptr(cam).value, pose6d=rt.ptr(pose6d).value, betas=rt.ptr(betas).value, rotmats=rt.ptr(R).value,
                           vertices=rt.ptr(vertices).value, joints=rt.ptr(joints).value, translation=rt.ptr(translation).value)
    record = None
    if return_record:
        record = f32(B, _abi.RECORD_FLOATS)
        out.record = rt.ptr(record).value
    rel_handle = bx = None
    if pose_relation is not None:       # `use_relate`, PR/PyTorch3DTest.py:3758-3759: the batch is ONE frame's players
        if boxes is None:
            raise ValueError('reconstruct_players: pose_relation needs boxes= [B,4] (x_min, y_min, x_max, y_max)')
        pose_relation.to(r.device)
        if B > pose_relation.max_players:
            raise ValueError(f'reconstruct_players: {B} players exceed the relation module\'s max_players')
        bx = rt.dev_f32(torch.as_tensor(boxes), r.device).reshape(B, 4).contiguous()
        tmp = f32(B, 216)
        out.relate_tmp = rt.ptr(tmp).value
        rel_handle = pose_relation.handle
    cfg = j2d = None
    if fit is not None:
        variant = {'camera': 0, 'pose_shape': 1}[fit]
        cfg = _fit_config(r, variant, iters, lr, None, select_mode)
        j2d = rt.dev_f32(torch.as_tensor(joints2D)[..., :2], r.device).reshape(B, 17, 2).contiguous()
    # one C call issues the whole chain of launches (csrc/pipeline.hip)
    r.check(r.lib.spsp_pipeline(r.ctx, regressor.handle, smpl.handle, rel_handle, rt.ptr(x), fmt, rt.ptr(bx), rt.ptr(j2d),
                                C.byref(cfg) if cfg is not None else None, B, config.FOCAL_LENGTH,
                                config.proxy_rep_input_wh, C.byref(out), r.stream()), 'spsp_pipeline')
    res = SimpleNamespace(cam=cam, rotmats=R, betas=betas, vertices=vertices, joints=joints, translation=translation,
                          pose6d=pose6d, record=record)
    if return_vertices2d:   # PR/PyTorch3DTest.py:3774-3776
        res.vertices2d = orthographic_project_to_pixels(vertices, cam, config.proxy_rep_input_wh)
    return res


def reconstruct_players_group(regressor, smpl, proxy_reps, joints2D, *, fit='pose_shape', iters=50, lr=None,
                              select_mode=_abi.FIT_SELECT_LAST, return_record=False):
    """`reconstruct_players` for a GROUP of consecutive batches with one tail (`spsp_pipeline_group`): the backbone runs batch
    by batch -- the launches of `reconstruct_players` at the batch size of `proxy_reps[i]` --, then IEF -> fit -> SMPL forward
    run once over all the group's players.  Those stages are latency-bound chains of small launches whose duration does not
    depend on the number of players (the fit is one workgroup per player), so a pipeline slot spends half as long in them per
    batch; every player's results are bit-identical to `reconstruct_players`'.
    proxy_reps: list of equally shaped batches; joints2D: list of [B,17,>=2] (one per batch) or one [G*B,17,>=2] tensor.
    Both handles need max_batch >= G*B.  Returns one result namespace per batch: views of the group's output tensors."""
    xs, fmt = [], None
    for x in proxy_reps:
        x, f = regressor.prepare_input(x)
        if fmt is not None and (f != fmt or x.shape != xs[0].shape):
            raise ValueError('reconstruct_players_group: the batches of a group must have one shape and layout')
        fmt = f
        xs.append(x)
    G, B = len(xs), xs[0].shape[0]
    N = G * B
    r = rt.get(xs[0].device)
    smpl.to(r.device)
    if N > smpl.max_batch or N > regressor.max_batch:
        raise ValueError(f'reconstruct_players_group: {G} x {B} players exceed a handle\'s max_batch ({regressor.max_batch} / {smpl.max_batch})')
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=r.device)
    cam, pose6d, betas, R = f32(N, 3), f32(N, 144), f32(N, 10), f32(N, 24, 3, 3)
    vertices, joints, translation = f32(N, 6890, 3), f32(N, 90, 3), f32(N, 3)
    out = _abi.PipelineOut(cam=rt.ptr(cam).value, pose6d=rt.ptr(pose6d).value, betas=rt.ptr(betas).value, rotmats=rt.ptr(R).value,
                           vertices=rt.ptr(vertices).value, joints=rt.ptr(joints).value, translation=rt.ptr(translation).value)
    record = None
    if return_record:
        record = f32(N, _abi.RECORD_FLOATS)
        out.record = rt.ptr(record).value
    cfg = j2d = None
    if fit is not None:
        variant = {'camera': 0, 'pose_shape': 1}[fit]
        cfg = _fit_config(r, variant, iters, lr, None, select_mode)
        if isinstance(joints2D, (list, tuple)):
            joints2D = torch.cat([rt.dev_f32(torch.as_tensor(j)[..., :2], r.device).reshape(B, 17, 2) for j in joints2D])
        j2d = rt.dev_f32(torch.as_tensor(joints2D)[..., :2], r.device).reshape(N, 17, 2).contiguous()
    ptrs = (C.c_void_p * G)(*[rt.ptr(x).value for x in xs])
    r.check(r.lib.spsp_pipeline_group(r.ctx, regressor.handle, smpl.handle, ptrs, G, fmt, rt.ptr(j2d),
                                      C.byref(cfg) if cfg is not None else None, B, config.FOCAL_LENGTH,
                                      config.proxy_rep_input_wh, C.byref(out), r.stream()), 'spsp_pipeline_group')
    res = []
    for g in range(G):
        sl = slice(g * B, (g + 1) * B)
        res.append(SimpleNamespace(cam=cam[sl], rotmats=R[sl], betas=betas[sl], vertices=vertices[sl], joints=joints[sl],
                                   translation=translation[sl], pose6d=pose6d[sl], record=None if record is None else record[sl],
                                   _keep=(j2d, xs)))
    return res


def pack_record(cam, R, betas):
    """[cam(3) | R(216) | betas(10)] = 229 fp32 per player: the all-gather payload."""
    B = cam.shape[0]
    r = rt.get(cam.device)
    rec = torch.empty((B, _abi.RECORD_FLOATS), dtype=torch.float32, device=cam.device)
    r.check(r.lib.spsp_pack_record(r.ctx, rt.ptr(cam.contiguous()), rt.ptr(R.contiguous()), rt.ptr(betas.contiguous()), B,
                                   rt.ptr(rec), r.stream()), 'spsp_pack_record')
    return rec


def unpack_record(rec):
    B = rec.shape[0]
    r = rt.get(rec.device)
    cam = torch.empty((B, 3), dtype=torch.float32, device=rec.device)
    R = torch.empty((B, 24, 3, 3), dtype=torch.float32, device=rec.device)
    betas = torch.empty((B, 10), dtype=torch.float32, device=rec.device)
    r.check(r.lib.spsp_unpack_record(r.ctx, rt.ptr(rec.contiguous()), B, rt.ptr(cam), rt.ptr(R), rt.ptr(betas), r.stream()),
            'spsp_unpack_record')
    return cam, R, betas


def _with_wait(res):
    """Gives a pipeline result its `wait()`: the way to consume the tensors on ANOTHER stream than the slot's.  The
    caller's current stream waits for the batch (`done`), and the result tensors -- allocated on the slot's stream --
    are marked as used by the caller's stream, so that the caching allocator does not hand their memory to the slot's
    next batch while a consumer kernel of the caller is still reading it.  (Done here and not in submit(): recording
    the stream makes every free of a result go through an event, which costs throughput when results are only
    synchronised on, as in bench.py.)
    `res.hold_until(event)` (set by the pipeline) is the other direction: work on another stream that READS this result
    (e.g. the record all-gather on the exchange's stream) registers its completion event, and the slot's next batch --
    which rewrites the slot's buffers; with HIP graphs the very same addresses -- waits for it before it starts."""
    def wait(stream=None):
        st = stream if stream is not None else torch.cuda.current_stream(res.stream.device)
        st.wait_event(res.done)
        for t in vars(res).values():
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(st)
        return res
    res.wait = wait
    return res


class PlayerPipeline:
    """Throughput mode of `reconstruct_players`: `depth` independent slots, each with its own HIP stream and its
    own regressor / SMPL handles (a handle's workspace serves one call in flight).  Batches are issued round-robin,
    so the latency-bound tail of batch i (IEF -> fit -> SMPL: a few wavefronts per player) runs on the GPU
    underneath the MFMA-bound backbone of batch i+1.  Every batch still goes through the complete path; results
    are returned with the event that marks their completion."""

    def __init__(self, make_regressor, make_smpl, device, depth=2, use_graphs=False, streams=None, plan='auto'):
        """`streams`: run on these HIP streams (one per slot) instead of new ones -- a second pipeline of a process
        should take over an idle pipeline's streams: past four active hardware queues the GPU time-slices them.
        `plan`: the regressor launch plan of the slots' handles ('auto': 'throughput' for two or more slots -- several batches
        in flight --, 'latency' for one; None: whatever make_regressor chose)."""
        self.device = torch.device(device)
        self.use_graphs = bool(use_graphs)
        self.slots = []
        from . import hw_queues
        if depth >= 2 and hw_queues() < depth + 1:
            import warnings
            warnings.warn(f'PlayerPipeline: {depth} slots + the caller\'s stream on {hw_queues()} hardware queues (GPU_MAX_HW_QUEUES as the GPU '
                          f'runtime saw it when it started): slots will share a queue and serialise (measured: 87 k instead of 129 k '
                          f'players/s at four slots).  Set GPU_MAX_HW_QUEUES=8 before the first GPU call of the process.', RuntimeWarning)
        if streams is not None and len(streams) != depth:
            raise ValueError(f'{len(streams)} streams for {depth} slots')
        for i in range(depth):
            st = streams[i] if streams is not None else torch.cuda.Stream(device=self.device)
            self.slots.append(SimpleNamespace(stream=st, regressor=make_regressor().to(self.device),
                                              smpl=make_smpl().to(self.device), done=None, graphs={}, warm=False, pending={}))
            if plan is not None:
                self.slots[-1].regressor.set_plan(('throughput' if depth >= 2 else 'latency') if plan == 'auto' else plan)
        self._next = 0
        self._nudge_in, self._nudge_keep = None, None

    def submit(self, proxy_rep, joints2D, buffer_set=0, **kw):
        """Issue one batch on the next slot.  With `use_graphs` the slot's ~60 launches are captured once into a
        HIP graph (keyed by the input buffers and options) and replayed: ~20 us of host time per batch instead of
        ~180 us, bit-identical results; the returned tensors are then the graph's static outputs, valid until
        the same slot is submitted to again (depth submits later).
        `buffer_set` (graphs): a slot keeps one set of static result buffers per value -- a caller whose results are read by
        work on ANOTHER stream (`res.hold_until(event)`: the record exchange of a multi-GPU run) alternates two sets, so that
        the slot's next batch does not have to wait for the reader of the previous one (bench.py: 100.8 k -> see DESIGN 6)."""
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        caller = torch.cuda.current_stream(self.device)
        self._nudge(caller)
        # inputs produced on the caller's stream: the slot must not start before they are there
        if self.use_graphs and not os.environ.get('SPSP_PIPE_EVENT_WAIT'):
            # ... on the HOST when a graph is about to be replayed.  Round 3, ROCm 7.2: a replay issued behind
            # hipStreamWaitEvent on a REAL event of another stream (one recorded behind a kernel; an event on an idle stream
            # completes on the spot and the wait is dropped) ran 13 % "faster" and gave WRONG results -- vertices centimetres off
            # while every run without such a wait is bit-identical to the serial path; an ordinary dispatch between the wait and
            # the replay did not help.  bench.py's parity gate found it.  So no event wait precedes a replay:
            # an idle caller stream (the common case) costs one hipStreamQuery, a busy one is waited for on the host; the same
            # goes for the readers registered through `hold_until` above.  SPSP_PIPE_EVENT_WAIT=1 restores the event waits (A/B).
            if not caller.query():
                caller.synchronize()
        else:
            slot.stream.wait_stream(caller)
        pending = slot.pending.setdefault(buffer_set if self.use_graphs else 0, [])
        for ev in pending:                                                  # readers of this buffer set's previous results on other streams
            if self.use_graphs and not os.environ.get('SPSP_PIPE_EVENT_WAIT'):
                ev.synchronize()                                            # (on the host in front of a graph replay: see above)
            else:
                slot.stream.wait_event(ev)                                  # (res.hold_until): the slot's buffers are about to be rewritten
        pending.clear()
        # The inputs were allocated on the caller's stream and are read on the slot's: tell the caching allocator, or a
        # buffer the caller drops right after submit() is handed to the caller's next allocation while this batch still
        # reads it.  The results are allocated on the slot's stream: consume them on another stream through res.wait().
        for t in (proxy_rep, joints2D, kw.get('boxes')):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(slot.stream)
        if self.use_graphs:
            return self._submit_graph(slot, proxy_rep, joints2D, kw, buffer_set, pending)
        with torch.cuda.stream(slot.stream):
            res = reconstruct_players(slot.regressor, slot.smpl, proxy_rep, joints2D, **kw)
            res.done = torch.cuda.Event()
            res.done.record(slot.stream)
        res.stream = slot.stream
        res.hold_until = pending.append
        return _with_wait(res)

    def _nudge(self, caller):
        """TEST HOOK (SPSP_PIPE_TEST_NUDGE=1): a 64-float fill on the caller's stream `2 * depth` submits after the last graph
        capture, then every SPSP_PIPE_NUDGE_PERIOD submits -- the smallest stand-in for an application that runs kernels of
        its own on the stream it submits from.  It is what exposed the hazard `submit()` now avoids (see there):
        tests/test_gpu_parity.py::test_player_pipeline_is_correct_next_to_kernels_on_the_callers_stream."""
        if self._nudge_in is None or not os.environ.get('SPSP_PIPE_TEST_NUDGE'):
            return
        self._nudge_in -= 1
        if self._nudge_in <= 0:
            with torch.cuda.stream(caller):
                if self._nudge_keep is None:
                    self._nudge_keep = torch.zeros(64, dtype=torch.float32, device=self.device)
                else:
                    self._nudge_keep.zero_()
            period = int(os.environ.get('SPSP_PIPE_NUDGE_PERIOD', '0'))
            self._nudge_in = period if period > 0 else None

    def _submit_graph(self, slot, proxy_rep, joints2D, kw, buffer_set, pending):
        if not (torch.is_tensor(proxy_rep) and proxy_rep.is_cuda and torch.is_tensor(joints2D) and joints2D.is_cuda):
            raise ValueError('PlayerPipeline(use_graphs=True): inputs must be device tensors (a graph replays fixed buffers)')
        key = (proxy_rep.data_ptr(), tuple(proxy_rep.shape), proxy_rep.dtype, joints2D.data_ptr(), tuple(joints2D.shape),
               joints2D.dtype, tuple(sorted(kw.items())), buffer_set)
        with torch.cuda.stream(slot.stream):
            entry = slot.graphs.get(key)
            if entry is None:
                if not slot.warm:       # first use of the slot: run eagerly once (one-time kernel attribute set-up)
                    reconstruct_players(slot.regressor, slot.smpl, proxy_rep, joints2D, **kw)
                    slot.warm = True
                slot.stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=slot.stream):
                    res = reconstruct_players(slot.regressor, slot.smpl, proxy_rep, joints2D, **kw)
                if len(slot.graphs) >= 16:
                    slot.graphs.pop(next(iter(slot.graphs)))
                # the inputs are kept alive with the graph that reads them
                entry = slot.graphs[key] = (g, res, proxy_rep, joints2D)
                self._nudge_in = 2 * len(self.slots)
            g, res = entry[0], entry[1]
            slot.regressor.check_status()        # a replay makes no library call: device-side failures of earlier batches surface here
            g.replay()
            out = SimpleNamespace(**vars(res))
            out.done = torch.cuda.Event()
            out.done.record(slot.stream)
        out.stream = slot.stream
        out.hold_until = pending.append
        return _with_wait(out)

    def submit_group(self, proxy_reps, joints2Ds, buffer_set=0, **kw):
        """`submit` for a GROUP of consecutive batches on ONE slot with one tail (`reconstruct_players_group`): the backbone
        batch by batch, IEF -> fit -> SMPL once over all the group's players.  A slot's stream then spends half (a third, ..)
        as long per batch in those latency-bound stages, during which it has no convolution in flight -- the throughput form
        for a caller that can hand over two batches at a time (bench.py --tail-group).  The handles need max_batch >= the
        group's players.  Returns one result per batch (views of the group's tensors; all complete at the same event)."""
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        caller = torch.cuda.current_stream(self.device)
        if self.use_graphs:
            if not caller.query():
                caller.synchronize()                                        # (on the host in front of a graph replay: see submit)
        else:
            slot.stream.wait_stream(caller)
        pending = slot.pending.setdefault(buffer_set if self.use_graphs else 0, [])
        for ev in pending:
            if self.use_graphs:
                ev.synchronize()
            else:
                slot.stream.wait_event(ev)
        pending.clear()
        for t in list(proxy_reps) + list(joints2Ds):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(slot.stream)
        with torch.cuda.stream(slot.stream):
            if not self.use_graphs:
                res = reconstruct_players_group(slot.regressor, slot.smpl, proxy_reps, joints2Ds, **kw)
            else:
                if not all(torch.is_tensor(t) and t.is_cuda for t in list(proxy_reps) + list(joints2Ds)):
                    raise ValueError('PlayerPipeline(use_graphs=True): inputs must be device tensors (a graph replays fixed buffers)')
                key = ('group', tuple(t.data_ptr() for t in proxy_reps), tuple(proxy_reps[0].shape), proxy_reps[0].dtype,
                       tuple(t.data_ptr() for t in joints2Ds), tuple(sorted(kw.items())), buffer_set)
                entry = slot.graphs.get(key)
                if entry is None:
                    if not slot.warm:
                        reconstruct_players_group(slot.regressor, slot.smpl, proxy_reps, joints2Ds, **kw)
                        slot.warm = True
                    slot.stream.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=slot.stream):
                        res = reconstruct_players_group(slot.regressor, slot.smpl, proxy_reps, joints2Ds, **kw)
                    if len(slot.graphs) >= 16:
                        slot.graphs.pop(next(iter(slot.graphs)))
                    entry = slot.graphs[key] = (g, res, list(proxy_reps), list(joints2Ds))
                g, res = entry[0], entry[1]
                slot.regressor.check_status()
                g.replay()
            done = torch.cuda.Event()
            done.record(slot.stream)
        outs = []
        for one in res:
            o = SimpleNamespace(**vars(one))
            o.done, o.stream, o.hold_until = done, slot.stream, pending.append
            outs.append(_with_wait(o))
        return outs

    def synchronize(self):
        for s in self.slots:
            s.stream.synchronize()
