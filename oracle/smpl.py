"""Oracle: SMPL forward.  Test infrastructure only.  PARITY UNPINNED.

The reference's `SMPL.forward` (`PR/models/smpl_official.py:27-41`) delegates to the third-party
`smplx` package (`smplx.SMPL.forward` -> `smplx.lbs.lbs`), which is neither vendored under
/root/reference nor installed here, and is un-pinned in `PR/requirements.txt:6`.  This module
restates smplx's published algorithm op-for-op (same einsum / matmul formulation, so that
`torch.autograd` differentiates the same graph the reference's fit loops differentiate):

  blend_shapes      v_shaped = v_template + einsum('bl,mkl->bmk', betas, shapedirs)
  vertices2joints   J = einsum('bik,ji->bjk', v_shaped, J_regressor)
  pose2rot=False    pose_feature = (R[:,1:] - I).view(B,207); v_posed = v_shaped + pose_feature @ posedirs
  batch_rigid_transform   chain of 4x4 transforms over `parents`, A_k = W_k with t -= W_k.R @ J_k
  skinning          T = lbs_weights @ A.view(B,24,16); verts = (T @ [v_posed;1])[:3]
  VertexJointSelector     joints = cat(24 posed chain joints, 21 picked vertices)  (face, feet, finger tips)
  wrapper (`smpl_official.py:30-34`)  + J_regressor_extra(9) + cocoplus(19) + h36m(17) from the VERTICES -> 90

and carries a float64 numpy twin (`smpl_forward_np`) used to cross-check the fp32 path.
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F


class OracleSMPL:
    """Holds the model arrays as fp32 torch tensors (what `smplx.SMPL.__init__` registers)."""

    def __init__(self, model):
        f = lambda k: torch.from_numpy(np.ascontiguousarray(model[k])).float()
        self.v_template = f('v_template')                    # [6890,3]
        self.shapedirs = f('shapedirs')                      # [6890,3,10]
        self.posedirs = f('posedirs')                        # [207,20670]
        self.J_regressor = f('J_regressor')                  # [24,6890]
        self.lbs_weights = f('lbs_weights')                  # [6890,24]
        self.parents = [int(p) for p in model['parents']]
        self.vertex_joint_ids = torch.from_numpy(np.asarray(model['vertex_joint_ids'], dtype=np.int64))
        self.J_regressor_extra = f('J_regressor_extra')
        self.J_regressor_cocoplus = f('J_regressor_cocoplus')
        self.J_regressor_h36m = f('J_regressor_h36m')
        self.faces = np.asarray(model['faces'])

    def __call__(self, body_pose=None, global_orient=None, betas=None, pose2rot=False):
        """Mirrors `SMPL.forward(body_pose=[B,23,3,3], global_orient=[B,1,3,3], betas=[B,10],
        pose2rot=False)`; `__call__(betas=)` alone gives the T-pose mesh (`PR/predict/predict_3D.py:148`)."""
        assert not pose2rot or (body_pose is None and global_orient is None)
        B = betas.shape[0]
        if body_pose is None:
            body_pose = torch.eye(3).expand(B, 23, 3, 3)
        if global_orient is None:
            global_orient = torch.eye(3).expand(B, 1, 3, 3)
        full_pose = torch.cat([global_orient.reshape(-1, 1, 3, 3), body_pose.reshape(-1, 23, 3, 3)], dim=1)
        verts, J_t = self._lbs(betas, full_pose)
        picked = torch.index_select(verts, 1, self.vertex_joint_ids)
        joints45 = torch.cat([J_t, picked], dim=1)
        v2j = lambda R: torch.einsum('bik,ji->bjk', [verts, R])
        joints = torch.cat([joints45, v2j(self.J_regressor_extra), v2j(self.J_regressor_cocoplus),
                            v2j(self.J_regressor_h36m)], dim=1)
        return SimpleNamespace(vertices=verts, joints=joints, betas=betas, body_pose=body_pose,
                               global_orient=global_orient, full_pose=full_pose)

    def _lbs(self, betas, rot_mats):
        B = betas.shape[0]
        v_shaped = self.v_template + torch.einsum('bl,mkl->bmk', [betas, self.shapedirs])
        J = torch.einsum('bik,ji->bjk', [v_shaped, self.J_regressor])
        ident = torch.eye(3, dtype=torch.float32)
        pose_feature = (rot_mats[:, 1:].reshape(B, -1, 3, 3) - ident).reshape(B, -1)
        v_posed = torch.matmul(pose_feature, self.posedirs).view(B, -1, 3) + v_shaped
        J_t, A = self._rigid(rot_mats, J)
        W = self.lbs_weights.unsqueeze(0).expand([B, -1, -1])
        T = torch.matmul(W, A.view(B, 24, 16)).view(B, -1, 4, 4)
        homo = torch.cat([v_posed, torch.ones([B, v_posed.shape[1], 1])], dim=2)
        v_homo = torch.matmul(T, homo.unsqueeze(-1))
        return v_homo[:, :, :3, 0], J_t

    def _rigid(self, rot_mats, joints):
        joints = joints.unsqueeze(-1)
        rel = joints.clone()
        rel[:, 1:] = rel[:, 1:] - joints[:, self.parents[1:]]
        tm = torch.cat([F.pad(rot_mats.reshape(-1, 3, 3), [0, 0, 0, 1]),
                        F.pad(rel.reshape(-1, 3, 1), [0, 0, 0, 1], value=1)], dim=2).reshape(-1, 24, 4, 4)
        chain = [tm[:, 0]]
        for i in range(1, 24):
            chain.append(torch.matmul(chain[self.parents[i]], tm[:, i]))
        transforms = torch.stack(chain, dim=1)
        posed = transforms[:, :, :3, 3]
        jh = F.pad(joints, [0, 0, 0, 1])
        rel_t = transforms - F.pad(torch.matmul(transforms, jh), [3, 0, 0, 0, 0, 0, 0, 0])
        return posed, rel_t


def smpl_forward_np(model, R, betas):
    """float64 numpy twin: R [B,24,3,3], betas [B,10] -> (verts [B,6890,3], joints [B,90,3])."""
    R = np.asarray(R, np.float64)
    betas = np.asarray(betas, np.float64)
    B = R.shape[0]
    m = {k: np.asarray(v, np.float64) for k, v in model.items() if k not in ('parents', 'faces', 'vertex_joint_ids')}
    parents = model['parents']
    v_shaped = m['v_template'][None] + np.einsum('bl,mkl->bmk', betas, m['shapedirs'])
    J = np.einsum('bik,ji->bjk', v_shaped, m['J_regressor'])
    pf = (R[:, 1:] - np.eye(3)).reshape(B, 207)
    v_posed = v_shaped + (pf @ m['posedirs']).reshape(B, -1, 3)
    Wr = np.zeros((B, 24, 3, 3))
    Wt = np.zeros((B, 24, 3))
    Wr[:, 0], Wt[:, 0] = R[:, 0], J[:, 0]
    for k in range(1, 24):
        p = parents[k]
        Wr[:, k] = Wr[:, p] @ R[:, k]
        Wt[:, k] = np.einsum('bij,bj->bi', Wr[:, p], J[:, k] - J[:, p]) + Wt[:, p]
    At = Wt - np.einsum('bkij,bkj->bki', Wr, J)
    Tr = np.einsum('vk,bkij->bvij', m['lbs_weights'], Wr)
    Tt = np.einsum('vk,bki->bvi', m['lbs_weights'], At)
    verts = np.einsum('bvij,bvj->bvi', Tr, v_posed) + Tt
    picked = verts[:, model['vertex_joint_ids']]
    v2j = lambda Rg: np.einsum('bik,ji->bjk', verts, Rg)
    joints = np.concatenate([Wt, picked, v2j(m['J_regressor_extra']), v2j(m['J_regressor_cocoplus']),
                             v2j(m['J_regressor_h36m'])], axis=1)
    return verts, joints
