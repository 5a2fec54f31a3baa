"""Oracle: small geometry / loss pieces of the hot path (SURVEY 8a rows a1,a5,a7,a8,a11,a12).
Test infrastructure only.  Every function is pinned against the imported reference function by
`oracle/gen_golden.py` -> `tests/golden/geometry.npz`."""
import numpy as np
import torch
import torch.nn.functional as F

FOCAL_LENGTH = 5000.0          # PR/config.py:15
REGRESSOR_IMG_WH = 256         # PR/config.py:16
PROXY_REP_INPUT_WH = 512       # S/global_var.py:32
SMPL_TO_KPRCNN_MAP = [24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]  # PR/config.py:38


def rot6d_to_rotmat(x):
    """`PR/utils/rigid_transform_utils.py:27-41`: [B,144] (or [N,6]) -> [N,3,3], columns b1,b2,b3."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def orthographic_project(points3D, cam):
    """`PR/utils/cam_utils.py:5-26`: u = s(x+tx), v = s(y+ty)."""
    s = cam[:, 0:1]
    return torch.stack([s * (points3D[:, :, 0] + cam[:, 1:2]), s * (points3D[:, :, 1] + cam[:, 2:3])], dim=-1)


def undo_keypoint_normalisation(kp, img_wh):
    """`PR/utils/joints2d_utils.py:5-10`."""
    return (kp + 1) * (img_wh / 2.0)


def wp_to_translation(cam_wp, focal_length=FOCAL_LENGTH, resolution=PROXY_REP_INPUT_WH):
    """`PR/utils/cam_utils.py:28-34`."""
    t = 2 * focal_length / (resolution * cam_wp[:, 0] + 1e-9)
    return torch.stack([cam_wp[:, 1], cam_wp[:, 2], t], dim=-1)


def translation_to_wp(translation, focal_length=FOCAL_LENGTH, resolution=PROXY_REP_INPUT_WH):
    """`PR/utils/cam_utils.py:36-42`."""
    s = 2 * focal_length / (resolution * translation[:, 2] + 1e-9)
    return torch.stack([s, translation[:, 0], translation[:, 1]], dim=-1)


JOINTS2D_LOG_VAR = float(np.float32(-np.log(1.0 + 1e-6)))   # PR/losses/multi_task_loss.py:37-38


def joints2d_loss(pred_px, label_px):
    """joints2D branch of `HomoscedasticUncertaintyWeightedMultiTaskLoss.forward`
    (`PR/losses/multi_task_loss.py:97-113`) with init weight 1.0 (`PR/PyTorch3DTest.py:3032`):
    both sides normalised with REGRESSOR_IMG_WH=256 although the pixels live in 512-space."""
    lv = torch.tensor(JOINTS2D_LOG_VAR, dtype=torch.float32)
    lab = (2.0 * label_px) / REGRESSOR_IMG_WH - 1.0
    pred = (2.0 * pred_px) / REGRESSOR_IMG_WH - 1.0
    mse = F.mse_loss(pred, lab, reduction='mean')
    return mse * torch.exp(-lv) + lv


def joints2d_l2e(pred_px, label_px):
    """`TrainingLossesAndMetricsTracker` joints2D_l2es (`PR/metrics/train_loss_and_metrics_tracker.py:
    211-222`, per-epoch division by 17 at :262-273): mean over the 17 joints of the pixel L2 error."""
    p, l = np.asarray(pred_px, np.float32), np.asarray(label_px, np.float32)
    return float(np.sum(np.linalg.norm(p - l, axis=-1)) / 17)


def gaussian_heatmaps(joints2D, img_wh, std=4):
    """`convert_2Djoints_to_gaussian_heatmaps` (`PR/utils/label_conversions.py:58-87`)."""
    nj = joints2D.shape[0]
    size = 2 * std
    hm = np.zeros((img_wh, img_wh, nj), dtype=np.float32)
    for i in range(nj):
        if np.all(joints2D[i] > -size) and np.all(joints2D[i] < img_wh - 1 + size):
            gx, gy = np.meshgrid(np.linspace(-size, size, 2 * size), np.linspace(-size, size, 2 * size))
            g = np.exp(-(np.sqrt(gx * gx + gy * gy) ** 2 / (2.0 * std ** 2)))
            c = joints2D[i]
            hs_x, he_x = max(0, c[0] - size), min(img_wh - 1, c[0] + size)
            hs_y, he_y = max(0, c[1] - size), min(img_wh - 1, c[1] + size)
            gs_x, ge_x = max(0, size - c[0]), min(2 * size, 2 * size - (size + c[0] - (img_wh - 1)))
            gs_y, ge_y = max(0, size - c[1]), min(2 * size, 2 * size - (size + c[1] - (img_wh - 1)))
            hm[hs_y:he_y, hs_x:he_x, i] = g[gs_y:ge_y, gs_x:ge_x]
    return hm


def create_proxy_representation(silhouette, joints2D, in_wh, out_wh):
    """`PR/predict/predict_3D.py:67-81`.  cv2.INTER_NEAREST at an integer factor f=in_wh/out_wh
    samples source pixel floor(dst*f), i.e. a strided slice; joints are scaled then TRUNCATED to int16."""
    f = in_wh // out_wh
    assert f * out_wh == in_wh, 'oracle restates the integer-factor case used by the reference (512->256)'
    sil = silhouette[::f, ::f]
    j = joints2D[:, :2] * np.array([out_wh / float(in_wh), out_wh / float(in_wh)])
    hm = gaussian_heatmaps(j.astype(np.int16), out_wh)
    proxy = np.concatenate([sil[:, :, None], hm], axis=-1)
    return np.transpose(proxy, [2, 0, 1])


# ---------------------------------------------------------------- front half of the proxy builder
CROP_FILL_BGR = (53, 88, 83)          # S/global_utils.py:288


def crop_image(image, box, border):
    """`S/global_utils.py:257-290` with `cv2.copyMakeBorder(..., BORDER_CONSTANT, value=(53, 88, 83))` written out."""
    left, top, right, bottom = box[0] - border, box[1] - border, box[2] + border, box[3] + border
    left, top = max(left, 0), max(top, 0)
    right, bottom = min(right, image.shape[1]), min(bottom, image.shape[0])
    croped = image[top:bottom, left:right]
    size = max(croped.shape[0], croped.shape[1])
    if size == croped.shape[0]:
        diff = size - croped.shape[1]
        pad_top = pad_bottom = 0
        pad_left = pad_right = diff // 2
        if diff % 2 != 0:
            pad_left += 1
    else:
        diff = size - croped.shape[0]
        pad_left = pad_right = 0
        pad_top = pad_bottom = diff // 2
        if diff % 2 != 0:
            pad_top += 1
    ans = np.empty((croped.shape[0] + pad_top + pad_bottom, croped.shape[1] + pad_left + pad_right, 3), dtype=image.dtype)
    ans[:] = np.array(CROP_FILL_BGR, dtype=image.dtype)
    ans[pad_top:pad_top + croped.shape[0], pad_left:pad_left + croped.shape[1]] = croped
    return ans


def resize_linear_u8(src, dsize):
    """`cv2.resize(src, (dsize, dsize), interpolation=cv2.INTER_LINEAR)` for 8-bit images, restated from OpenCV's
    published source (modules/imgproc/src/resize.cpp; OpenCV is absent here and un-pinned in the reference: unpinned):
    coordinate f = (float)((d + 0.5) * scale - 0.5) with scale = 1 / (dsize / ssize) in double, clamped taps, 11-bit
    fixed-point coefficients (cvRound = round half to even), rows first in exact integers, columns in the 16-bit SIMD
    form ((b0 (H0 >> 4)) >> 16) + ((b1 (H1 >> 4)) >> 16) + 2) >> 2; an exact 2 x 2 shrink takes the INTER_AREA path."""
    ssize = src.shape[0]
    assert src.shape[0] == src.shape[1] and src.dtype == np.uint8
    if ssize == 2 * dsize:
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= ssize - 1
    s = np.where(lo, 0, np.where(hi, ssize - 1, s))
    f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)              # np.rint: half to even, like cvRound
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    s1 = np.minimum(s + 1, ssize - 1)
    S = src.astype(np.int64)
    H = S[:, s, :] * a0[None, :, None] + S[:, s1, :] * a1[None, :, None]          # [ssize, dsize, 3]
    H0, H1 = H[s], H[s1]
    out = (((a0[:, None, None] * (H0 >> 4)) >> 16) + ((a1[:, None, None] * (H1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def crop_players(image, boxes, border, dsize=512):
    """What `S/PlayerDetection/player_detection.py:181-184` produces for every detected box of a frame."""
    return np.stack([resize_linear_u8(crop_image(image, [int(v) for v in b], border), dsize) for b in boxes])
