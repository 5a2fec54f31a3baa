"""Mirror of the 3-D part of `PR/predict/predict_3D.py` (:67-81 proxy builder, :128-149 inference)."""
from types import SimpleNamespace

import numpy as np
import torch

from .. import _abi, config
from .. import runtime as rt
from ..utils.cam_utils import orthographic_project_to_pixels
from ..utils.label_conversions import convert_2Djoints_to_gaussian_heatmaps
from ..utils.rigid_transform_utils import rot6d_to_rotmat


def create_proxy_representation(silhouette, joints2D, in_wh, out_wh):
    """`PR/predict/predict_3D.py:67-81`: nearest-resize the silhouette (cv2.INTER_NEAREST at an integer factor
    = strided slice), scale + int16-truncate the joints, rasterise 17 Gaussian heat-maps -> [18,out_wh,out_wh]."""
    f = in_wh // out_wh
    if f * out_wh != in_wh:
        raise ValueError('create_proxy_representation: in_wh must be an integer multiple of out_wh (reference: 512 -> 256)')
    silhouette = np.asarray(silhouette)[::f, ::f]
    j = np.asarray(joints2D)[:, :2] * np.array([out_wh / float(in_wh), out_wh / float(in_wh)])
    heatmaps = convert_2Djoints_to_gaussian_heatmaps(j.astype(np.int16), out_wh)
    proxy = np.concatenate([silhouette[:, :, None], heatmaps], axis=-1)
    return np.transpose(proxy, [2, 0, 1])


def create_proxy_representation_batch(silhouettes, joints2D, precision='f16'):
    """The same routine for a batch ON THE GPU, emitting the backbone's native input directly:
    silhouettes [B,512,512] (uint8 / bool / float32, values 0/1) and joints2D [B,17,2|3] (x, y[, conf]; 512-px
    space) -> [B,262,264,18] bf16 / fp16 device tensor, bit-identical to
    `regressor.pack(stack(create_proxy_representation(sil_b, joints_b, 512, 256)))`; feed it to
    `SingleInputRegressor(precision=...)` as is.  Fixed to the reference's 512 -> 256 geometry."""
    if precision not in ('bf16', 'f16'):
        raise ValueError(f"precision must be 'bf16' or 'f16', got {precision!r}")
    sil = silhouettes if isinstance(silhouettes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(silhouettes))
    r = rt.get(sil.device if sil.is_cuda else None)
    if sil.dim() != 3 or tuple(sil.shape[1:]) != (512, 512):
        raise ValueError(f'silhouettes must be [B,512,512], got {tuple(sil.shape)}')
    if sil.dtype == torch.bool:
        sil = sil.to(torch.uint8)
    if sil.dtype not in (torch.uint8, torch.float32):
        sil = sil.to(torch.float32)
    sil = sil.to(r.device).contiguous()
    j = rt.dev_f32(torch.as_tensor(joints2D), r.device).contiguous()
    B = sil.shape[0]
    if j.dim() != 3 or j.shape[0] != B or j.shape[1] != 17 or j.shape[2] < 2:
        raise ValueError(f'joints2D must be [B,17,>=2] with B={B}, got {tuple(j.shape)}')
    prec = _abi.PRECISION_F16 if precision == 'f16' else _abi.PRECISION_BF16
    out = torch.empty((B, 262, 264, 18), dtype=torch.float16 if precision == 'f16' else torch.bfloat16, device=r.device)
    fmt = _abi.SIL_U8 if sil.dtype == torch.uint8 else _abi.SIL_F32
    r.check(r.lib.spsp_proxy_build(r.ctx, rt.ptr(sil), fmt, rt.ptr(j), j.shape[2], B, prec, rt.ptr(out), r.stream()),
            'spsp_proxy_build')
    return out


def predict_3D(proxy_rep, regressor, smpl):
    """`PR/predict/predict_3D.py:128-149` for an already-built proxy batch: regressor -> rot6d -> SMPL mesh,
    2-D vertex projection in 512-px space, and the T-pose "reposed" mesh `smpl(betas=pred_shape)`."""
    pred_cam_wp, pred_pose, pred_shape = regressor(proxy_rep)
    B = pred_cam_wp.shape[0]
    R = rot6d_to_rotmat(pred_pose).view(B, 24, 3, 3)
    out = smpl(body_pose=R[:, 1:], global_orient=R[:, 0].unsqueeze(1), betas=pred_shape, pose2rot=False)
    verts2d = orthographic_project_to_pixels(out.vertices, pred_cam_wp, config.proxy_rep_input_wh)
    reposed = smpl(betas=pred_shape, return_joints=False)
    return SimpleNamespace(pred_cam_wp=pred_cam_wp, pred_pose_rotmats=R, pred_shape=pred_shape,
                           pred_vertices=out.vertices, pred_joints=out.joints, pred_vertices2d=verts2d,
                           pred_reposed_vertices=reposed.vertices)
