"""Mirror of `PR/utils/rigid_transform_utils.py` (hot-path part)."""
import torch

from .. import runtime as rt


def rot6d_to_rotmat(x):
    """6-D rotation representation -> rotation matrices, `PR/utils/rigid_transform_utils.py:27-41`.
    x: [..., 6k] tensor (viewed as (-1, 3, 2) like the reference) -> [N, 3, 3] on the GPU."""
    x = rt.dev_f32(x)
    n = x.numel() // 6
    if x.numel() != n * 6:
        raise ValueError(f'rot6d_to_rotmat: {x.numel()} elements is not a multiple of 6')
    r = rt.get(x.device)
    out = torch.empty((n, 3, 3), dtype=torch.float32, device=x.device)
    r.check(r.lib.spsp_rot6d_to_rotmat(r.ctx, rt.ptr(x), n, rt.ptr(out), r.stream()), 'spsp_rot6d_to_rotmat')
    return out
