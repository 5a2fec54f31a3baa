"""Mirror of the part of `S/global_utils.py` that sits immediately in front of the hot path: `crop_image` (:257-290),
fused with the `cv2.resize(..., player_crop_size, cv2.INTER_LINEAR)` that always follows it
(`S/PlayerDetection/player_detection.py:181-184`), for all detected boxes of a frame in one kernel launch."""
import numpy as np
import torch

from . import runtime as rt

player_crop_border = 40             # S/global_var.py:50
player_crop_size = (512, 512)       # S/global_var.py:51
player_crop_border_broad = 15       # S/global_var.py:53


def crop_players(image, boxes, border=player_crop_border, size=player_crop_size):
    """image: [H,W,3] uint8 BGR frame (numpy or tensor); boxes: [N,4] (x0, y0, x1, y1), cast to int32 as
    `player_detection.predict` does (:30).  Returns the N crops as a uint8 device tensor [N,size,size,3]: each box grown by
    `border`, clamped to the frame, padded to a square with BGR (53, 88, 83), resized to `size` with OpenCV's 8-bit
    INTER_LINEAR arithmetic -- `cv2.resize(crop_image(image, box, border), size, interpolation=cv2.INTER_LINEAR)` per box."""
    if size[0] != size[1]:
        raise ValueError(f'crop_players: square outputs only (the reference uses (512, 512)), got {size}')
    img = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
    if img.dim() != 3 or img.shape[2] != 3 or img.dtype != torch.uint8:
        raise ValueError(f'crop_players: image must be [H,W,3] uint8, got {tuple(img.shape)} {img.dtype}')
    r = rt.get(img.device if img.is_cuda else None)
    img = img.to(r.device).contiguous()
    bx = torch.as_tensor(np.asarray(boxes.cpu() if isinstance(boxes, torch.Tensor) else boxes).astype(np.int32)).reshape(-1, 4)
    bx = bx.to(r.device).contiguous()
    N = bx.shape[0]
    out = torch.empty((N, size[0], size[0], 3), dtype=torch.uint8, device=r.device)
    r.check(r.lib.spsp_crop_players(r.ctx, rt.ptr(img), img.shape[0], img.shape[1], rt.ptr(bx), N, int(border), int(size[0]),
                                    rt.ptr(out), r.stream()), 'spsp_crop_players')
    return out
