"""Mirror of the 3-D part of `PR/predict/predict_3D.py` (:67-81 proxy builder, :128-149 inference)."""
from types import SimpleNamespace

import numpy as np
import torch

from .. import config
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
