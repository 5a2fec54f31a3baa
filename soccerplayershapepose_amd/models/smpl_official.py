"""Mirror of `PR/models/smpl_official.py`: the SMPL body model with the 90-joint output, on gfx950.

`SMPL(model, batch_size=1).to(device)` then `smpl(body_pose=[B,23,3,3], global_orient=[B,1,3,3],
betas=[B,10], pose2rot=False)` or `smpl(betas=...)` -> object with `.vertices [B,6890,3]`,
`.joints [B,90,3]`, `.betas`, `.body_pose`, `.global_orient`, `.full_pose`; attribute `.faces`.
The arithmetic (smplx.lbs.lbs [EXT] + the three extra joint regressors, smpl_official.py:27-41) runs in
csrc/smpl.hip through `spsp_smpl_forward`."""
import ctypes as C
import os
import pickle
from types import SimpleNamespace

import numpy as np
import torch

from .. import _abi, config
from .. import runtime as rt

SMPLOutput = SimpleNamespace


def load_smpl_assets(model_dir):
    """Best-effort loader for the real (licensed, not shipped) assets named in `PR/config.py:3-8`:
    `<model_dir>/SMPL_NEUTRAL.pkl` + the three extra regressors.  Written against the documented
    formats; a .npz with the same array names is accepted too."""
    npz = model_dir if model_dir.endswith('.npz') else None
    if npz:
        d = dict(np.load(npz))
    else:
        path = os.path.join(model_dir, 'SMPL_NEUTRAL.pkl') if os.path.isdir(model_dir) else model_dir
        with open(path, 'rb') as f:
            raw = pickle.load(f, encoding='latin1')
        dense = lambda a: np.asarray(a.todense() if hasattr(a, 'todense') else a)
        posedirs = np.asarray(raw['posedirs'])                               # [6890,3,207]
        d = {'v_template': np.asarray(raw['v_template']), 'shapedirs': np.asarray(raw['shapedirs'])[:, :, :10],
             'posedirs': posedirs.reshape(-1, posedirs.shape[-1]).T, 'J_regressor': dense(raw['J_regressor']),
             'lbs_weights': np.asarray(raw['weights']), 'faces': np.asarray(raw['f']).astype(np.int32)}
        kt = np.asarray(raw['kintree_table'])[0].astype(np.int64)
        kt[0] = -1
        d['parents'] = kt.astype(np.int32)
    if 'vertex_joint_ids' not in d:
        from ..synthetic import VERTEX_JOINT_IDS
        d['vertex_joint_ids'] = VERTEX_JOINT_IDS
    for key, path in (('J_regressor_extra', config.J_REGRESSOR_EXTRA_PATH),
                      ('J_regressor_cocoplus', config.COCOPLUS_REGRESSOR_PATH),
                      ('J_regressor_h36m', config.H36M_REGRESSOR_PATH)):
        if key not in d:
            d[key] = np.load(path)
    return d


class SMPL:
    def __init__(self, model, batch_size=1, max_batch=None, **_ignored):
        """model: dict of arrays (names of `synthetic.smpl_model`) or a path handled by `load_smpl_assets`."""
        self._model = load_smpl_assets(model) if isinstance(model, str) else model
        self.batch_size = batch_size
        self.max_batch = max_batch or max(256, batch_size)
        self.faces = np.asarray(self._model['faces'])
        self._handle = None
        self._rt = None

    # -- reference-style plumbing ---------------------------------------------------------------
    def to(self, device):
        self._load(rt.get(device))
        return self

    def eval(self):
        return self

    def _load(self, r):
        if self._handle is not None and self._rt is r:
            return
        arrs, keep = _abi.smpl_arrays(self._model)
        h = C.c_void_p()
        with torch.cuda.device(r.device):
            r.check(r.lib.spsp_smpl_load(r.ctx, C.byref(arrs), self.max_batch, C.byref(h)), 'spsp_smpl_load')
        self._free()
        self._handle, self._rt = h, r

    def _free(self):
        if self._handle is not None:
            self._rt.lib.spsp_smpl_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    @property
    def handle(self):
        if self._handle is None:
            self._load(rt.get())
        return self._handle

    def debug(self, what, value):
        """Test hook (`spsp_smpl_debug`): e.g. `_abi.DEBUG_FIT_SCHEDULE`, 2 | 1 | 0 = which form of the fit kernel runs on this handle."""
        self._rt.check(self._rt.lib.spsp_smpl_debug(self.handle, what, value), 'spsp_smpl_debug')

    # -- forward ------------------------------------------------------------------------------------
    def full_pose_from(self, body_pose, global_orient, B, device):
        eye = None
        if body_pose is None or global_orient is None:
            eye = torch.eye(3, dtype=torch.float32, device=device)
        bp = rt.dev_f32(body_pose, device).reshape(-1, 23, 3, 3) if body_pose is not None else eye.expand(B, 23, 3, 3)
        go = rt.dev_f32(global_orient, device).reshape(-1, 1, 3, 3) if global_orient is not None else eye.expand(B, 1, 3, 3)
        if bp.shape[0] != B or go.shape[0] != B:
            raise ValueError(f'SMPL.forward: batch mismatch body_pose {tuple(bp.shape)}, global_orient {tuple(go.shape)}, B={B}')
        return torch.cat([go, bp], dim=1).contiguous(), bp, go

    def forward(self, body_pose=None, global_orient=None, betas=None, pose2rot=False, get_skin=True,
                return_verts=True, return_joints=True, full_pose=None):
        """`full_pose` [B,24,3,3] (contiguous CUDA fp32) may be given instead of body_pose + global_orient: it is
        used as is, without the torch.cat the reference-style arguments need."""
        if pose2rot and (body_pose is not None or global_orient is not None):
            raise NotImplementedError('axis-angle input (pose2rot=True) is not on the hot path: the reference always '
                                      'passes rotation matrices (PR/PyTorch3DTest.py:3082, PR/player_recon.py:1207)')
        if betas is None:
            raise ValueError('SMPL.forward: betas is required')
        if self._handle is None:
            dev = betas.device if isinstance(betas, torch.Tensor) and betas.is_cuda else None
            self._load(rt.get(dev))
        r = self._rt
        betas = rt.dev_f32(betas, r.device).reshape(-1, 10)
        B = betas.shape[0]
        if B > self.max_batch:
            raise ValueError(f'SMPL.forward: batch {B} exceeds max_batch {self.max_batch}')
        if full_pose is not None:
            full_pose = rt.dev_f32(full_pose, r.device).reshape(B, 24, 3, 3)
            bp, go = full_pose[:, 1:], full_pose[:, :1]
        else:
            full_pose, bp, go = self.full_pose_from(body_pose, global_orient, B, r.device)
        verts = torch.empty((B, 6890, 3), dtype=torch.float32, device=r.device) if return_verts else None
        joints = torch.empty((B, 90, 3), dtype=torch.float32, device=r.device) if return_joints else None
        r.check(r.lib.spsp_smpl_forward(r.ctx, self.handle, rt.ptr(full_pose), rt.ptr(betas), B, rt.ptr(verts),
                                        rt.ptr(joints), r.stream()), 'spsp_smpl_forward')
        return SMPLOutput(vertices=verts, joints=joints, betas=betas, body_pose=bp, global_orient=go, full_pose=full_pose)

    __call__ = forward

    def vertices_vjp(self, full_pose, betas, grad_vertices):
        """Reverse mode of `forward(...).vertices` (what `loss.backward()` does to the mesh in the reference's silhouette fit,
        `PR/player_recon.py:1429-1537`): full_pose [B,24,3,3], betas [B,10], grad_vertices [B,6890,3] -> (grad_full_pose
        [B,24,3,3], grad_betas [B,10])."""
        if self._handle is None:
            self._load(rt.get(grad_vertices.device if isinstance(grad_vertices, torch.Tensor) and grad_vertices.is_cuda else None))
        r = self._rt
        betas = rt.dev_f32(betas, r.device).reshape(-1, 10)
        B = betas.shape[0]
        if B > self.max_batch:
            raise ValueError(f'SMPL.vertices_vjp: batch {B} exceeds max_batch {self.max_batch}')
        R = rt.dev_f32(full_pose, r.device).reshape(B, 24, 3, 3).contiguous()
        g = rt.dev_f32(grad_vertices, r.device).reshape(B, 6890, 3).contiguous()
        gR = torch.empty((B, 24, 3, 3), dtype=torch.float32, device=r.device)
        gb = torch.empty((B, 10), dtype=torch.float32, device=r.device)
        r.check(r.lib.spsp_smpl_vertices_vjp(r.ctx, self.handle, rt.ptr(R), rt.ptr(betas), B, rt.ptr(g), rt.ptr(gR), rt.ptr(gb), r.stream()),
                'spsp_smpl_vertices_vjp')
        return gR, gb
