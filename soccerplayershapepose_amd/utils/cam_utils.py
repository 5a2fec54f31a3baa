"""Mirror of `PR/utils/cam_utils.py` (hot-path part: :5-42)."""
import torch

from .. import runtime as rt


def _project(points3D, cam_params, img_wh):
    points3D = rt.dev_f32(points3D)
    cam_params = rt.dev_f32(cam_params, points3D.device)
    if points3D.dim() != 3 or points3D.shape[-1] != 3 or cam_params.shape != (points3D.shape[0], 3):
        raise ValueError(f'orthographic_project_torch: bad shapes {tuple(points3D.shape)}, {tuple(cam_params.shape)}')
    B, n = points3D.shape[:2]
    r = rt.get(points3D.device)
    out = torch.empty((B, n, 2), dtype=torch.float32, device=points3D.device)
    r.check(r.lib.spsp_project(r.ctx, rt.ptr(points3D), rt.ptr(cam_params), B, n, float(img_wh), rt.ptr(out),
                               r.stream()), 'spsp_project')
    return out


def orthographic_project_torch(points3D, cam_params):
    """Weak-perspective projection u = s(x+tx), v = s(y+ty) (`PR/utils/cam_utils.py:5-26`)."""
    return _project(points3D, cam_params, 0.0)


def orthographic_project_to_pixels(points3D, cam_params, img_wh):
    """Fused `undo_keypoint_normalisation(orthographic_project_torch(p, cam), img_wh)`."""
    return _project(points3D, cam_params, img_wh)


def convert_weak_perspective_to_camera_translation_torch(cam_wp, focal_length, resolution):
    """`PR/utils/cam_utils.py:28-34`: [s, tx, ty] -> [tx, ty, 2 f / (res s + 1e-9)]."""
    cam_wp = rt.dev_f32(cam_wp)
    r = rt.get(cam_wp.device)
    out = torch.empty_like(cam_wp)
    r.check(r.lib.spsp_wp_to_translation(r.ctx, rt.ptr(cam_wp), cam_wp.shape[0], float(focal_length), float(resolution),
                                         rt.ptr(out), r.stream()), 'spsp_wp_to_translation')
    return out


def convert_camera_translation_to_weak_perspective_torch(translation, focal_length, resolution):
    """`PR/utils/cam_utils.py:36-42` (pure tensor arithmetic, off the hot path)."""
    s = 2 * focal_length / (resolution * translation[:, 2] + 1e-9)
    return torch.stack([s, translation[:, 0], translation[:, 1]], dim=-1)
