"""The per-player fit loops and the batched inference pipeline, re-designed for gfx950.

Mirrors (same names, argument meaning, in-place contract):
  optimize_camera           PR/PyTorch3DTest.py:3055-3124   cam + global-orientation fit, 50 Adam steps, lr 0.01
  single_view_optimization  PR/player_recon.py:1172-1294    pose + shape + camera fit ("theta fit"), lr 1e-3
  evaluate_model_2d core    PR/PyTorch3DTest.py:3738-3776   regressor -> rot6d -> per-player fit -> SMPL -> 2-D verts
The reference runs these one player at a time with a full SMPL.forward + autograd per iteration; here a
whole batch of players is ONE kernel launch (csrc/fit.hip), one wavefront per player."""
import ctypes as C
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
    iters = interation if interation is not None else config.texture_single_opt_interation
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


def multi_view_optimization(smpl, body_pose, global_orient, betas, translation, joints2D, interation=None, lr=None,
                            rounds=3, optimise_body_pose=False, return_history=False):
    """Optimisation body of `multi_view_optimization` (`PR/player_recon.py:1691-1979`) for a batch of players seen in
    V <= 4 views each.  Inputs are the per-view single-view results the reference loads from its npz records
    (`:1673-1689`; `records.load_record`): body_pose [B,V,23,3,3], global_orient [B,V,1,3,3], betas [B,V,10],
    translation [B,V,3], and the detected joints2D [B,V,17,>=2] (512-px space).

    As in the reference the shared body pose and betas start from the MEAN over the views (`:1691-1694`), every view
    keeps its own camera (recovered from its translation, `:1689`) and orientation, and `rounds` x {camera block, pose
    block} of `interation` epochs run with the best mean joints2D_l2es iterate kept (csrc/fit_mv_core.hpp).
    optimise_body_pose=False reproduces the reference, whose pose block only ever moves betas (the posed tensor is
    concatenated before requires_grad is set, `:1724-1731`); True optimises body_pose idx {0-5, 8-20} as intended.

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
    hist = torch.empty((B, cfg.rounds * 2 * cfg.iters), dtype=torch.float32, device=r.device) if return_history else None
    smpl.to(r.device)
    r.check(r.lib.spsp_fit_multiview(r.ctx, smpl.handle, C.byref(cfg), B, V, rt.ptr(body), rt.ptr(bmean), rt.ptr(cam),
                                     rt.ptr(go), rt.ptr(j2d), rt.ptr(hist), r.stream()), 'spsp_fit_multiview')
    trans = convert_weak_perspective_to_camera_translation_torch(cam.reshape(B * V, 3), config.FOCAL_LENGTH,
                                                                 config.proxy_rep_input_wh).reshape(B, V, 3)
    out = SimpleNamespace(body_pose=body.reshape(B, 23, 3, 3), betas=bmean, translation=trans,
                          global_orient=go.reshape(B, V, 1, 3, 3), cam_wp=cam)
    if return_history:
        out.val_l2e_history = hist.reshape(B, cfg.rounds, 2, cfg.iters)
    return out


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

    def __init__(self, make_regressor, make_smpl, device, depth=2, use_graphs=False, streams=None):
        """`streams`: run on these HIP streams (one per slot) instead of new ones -- a second pipeline of a process
        should take over an idle pipeline's streams: past four active hardware queues the GPU time-slices them."""
        self.device = torch.device(device)
        self.use_graphs = bool(use_graphs)
        self.slots = []
        if streams is not None and len(streams) != depth:
            raise ValueError(f'{len(streams)} streams for {depth} slots')
        for i in range(depth):
            st = streams[i] if streams is not None else torch.cuda.Stream(device=self.device)
            self.slots.append(SimpleNamespace(stream=st, regressor=make_regressor().to(self.device),
                                              smpl=make_smpl().to(self.device), done=None, graphs={}, warm=False, pending=[]))
        self._next = 0

    def submit(self, proxy_rep, joints2D, **kw):
        """Issue one batch on the next slot.  With `use_graphs` the slot's ~60 launches are captured once into a
        HIP graph (keyed by the input buffers and options) and replayed: ~20 us of host time per batch instead of
        ~180 us, bit-identical results; the returned tensors are then the graph's static outputs, valid until
        the same slot is submitted to again (depth submits later)."""
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        caller = torch.cuda.current_stream(self.device)
        slot.stream.wait_stream(caller)                                     # inputs produced on the caller's stream
        for ev in slot.pending:                                             # readers of the slot's previous results on other streams
            slot.stream.wait_event(ev)                                      # (res.hold_until): the slot's buffers are about to be rewritten
        slot.pending.clear()
        # The inputs were allocated on the caller's stream and are read on the slot's: tell the caching allocator, or a
        # buffer the caller drops right after submit() is handed to the caller's next allocation while this batch still
        # reads it.  The results are allocated on the slot's stream: consume them on another stream through res.wait().
        for t in (proxy_rep, joints2D, kw.get('boxes')):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(slot.stream)
        if self.use_graphs:
            return self._submit_graph(slot, proxy_rep, joints2D, kw)
        with torch.cuda.stream(slot.stream):
            res = reconstruct_players(slot.regressor, slot.smpl, proxy_rep, joints2D, **kw)
            res.done = torch.cuda.Event()
            res.done.record(slot.stream)
        res.stream = slot.stream
        res.hold_until = slot.pending.append
        return _with_wait(res)

    def _submit_graph(self, slot, proxy_rep, joints2D, kw):
        if not (torch.is_tensor(proxy_rep) and proxy_rep.is_cuda and torch.is_tensor(joints2D) and joints2D.is_cuda):
            raise ValueError('PlayerPipeline(use_graphs=True): inputs must be device tensors (a graph replays fixed buffers)')
        key = (proxy_rep.data_ptr(), tuple(proxy_rep.shape), proxy_rep.dtype, joints2D.data_ptr(), tuple(joints2D.shape),
               joints2D.dtype, tuple(sorted(kw.items())))
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
                if len(slot.graphs) >= 8:
                    slot.graphs.pop(next(iter(slot.graphs)))
                # the inputs are kept alive with the graph that reads them
                entry = slot.graphs[key] = (g, res, proxy_rep, joints2D)
            g, res = entry[0], entry[1]
            slot.regressor.check_status()        # a replay makes no library call: device-side failures of earlier batches surface here
            g.replay()
            out = SimpleNamespace(**vars(res))
            out.done = torch.cuda.Event()
            out.done.record(slot.stream)
        out.stream = slot.stream
        out.hold_until = slot.pending.append
        return _with_wait(out)

    def synchronize(self):
        for s in self.slots:
            s.stream.synchronize()
