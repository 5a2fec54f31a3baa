"""Oracle: the silhouette render and metric of the fit drivers.  Test infrastructure only.  PARITY UNPINNED:
`neural_renderer` (the reference's `nr.Renderer`, `PR/player_recon.py:1075-1080`) is an un-vendored, un-pinned dependency
(PR/requirements.txt) that is not installed here; its forward pass is restated from the published implementation:

  projection (camera_mode='projection', R = I, no distortion)
      p = v + t;  x' = x / (z + 1e-9), y' = y / (z + 1e-9);  (u, v) = (f x' + cx, f y' + cy)   [K = intrinsics(512, 512, f)]
      v = orig_size - v;  u_n = 2 (u - orig_size / 2) / orig_size,  v_n likewise                (fp32 throughout)
  rasterize_silhouettes (anti_aliasing=True, fill_back=True, near 0.1, far 100)
      grid is = 2 * 512; sample xp = (2 xi + 1 - is) / is, yp likewise, output row r <-> yi = is - 1 - r;
      face (p0, p1, p2) covers a sample unless (yp - y0)(x1 - x0) < (xp - x0)(y1 - y0) or the two cyclic successors hold;
      the reversed copy (fill_back) covers it unless the same three hold with '>'; depth (perspective-correct, from clamped
      barycentric weights) must lie in (near, far); silhouette = 2 x 2 mean of the coverage
  metric  `silhouette_iou` (`PR/metrics/train_loss_and_metrics_tracker.py:224-229`), loss (`PR/losses/multi_task_loss.py:132-144`)
  rule    `determine_save_model_weights_this_epoch` (:291-298) over ['joints2D_l2es', 'silhouette_iou'] as
          `PR/player_recon.py:1259-1271` applies it.
"""
import numpy as np

F32 = np.float32


def _fma(a, b, c):
    return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(F32)


def project(verts, translation, focal=5000.0, wh=512):
    v = verts.astype(F32) + translation.astype(F32)[None, :]
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    xd, yd = x / (z + F32(1e-9)), y / (z + F32(1e-9))
    half, fwh = F32(0.5 * wh), F32(wh)
    u = _fma(xd, F32(focal), half)
    vv = fwh - _fma(yd, F32(focal), half)
    return F32(2) * (u - half) / fwh, F32(2) * (vv - half) / fwh, z


def render_silhouette(verts, faces, translation, focal=5000.0, wh=512):
    """[6890,3], [F,3], [3] -> [wh,wh] fp32 in {0,.25,.5,.75,1}."""
    is_ = 2 * wh
    fis = F32(is_)
    xn, yn, zz = project(verts, translation, focal, wh)
    cover = np.zeros((is_, is_), dtype=bool)
    for f in np.asarray(faces):
        x0, x1, x2 = xn[f[0]], xn[f[1]], xn[f[2]]
        y0, y1, y2 = yn[f[0]], yn[f[1]], yn[f[2]]
        z0, z1, z2 = zz[f[0]], zz[f[1]], zz[f[2]]
        xmin, xmax, ymin, ymax = min(x0, x1, x2), max(x0, x1, x2), min(y0, y1, y2), max(y0, y1, y2)
        if not (xmax >= -1 and xmin <= 1 and ymax >= -1 and ymin <= 1):
            continue
        c0 = max(int(np.floor((xmin * fis + fis - F32(1)) * F32(0.5))) - 1, 0)
        c1 = min(int(np.ceil((xmax * fis + fis - F32(1)) * F32(0.5))) + 1, is_ - 1)
        q0 = max(int(np.floor((ymin * fis + fis - F32(1)) * F32(0.5))) - 1, 0)
        q1 = min(int(np.ceil((ymax * fis + fis - F32(1)) * F32(0.5))) + 1, is_ - 1)
        if c1 < c0 or q1 < q0:
            continue
        xp = ((F32(2) * np.arange(c0, c1 + 1, dtype=F32) + F32(1) - fis) / fis)[None, :]
        yp = ((F32(2) * np.arange(q0, q1 + 1, dtype=F32) + F32(1) - fis) / fis)[:, None]
        e0x, e0y, e1x, e1y, e2x, e2y = x1 - x0, y1 - y0, x2 - x1, y2 - y1, x0 - x2, y0 - y2
        l0, r0 = (yp - y0) * e0x, (xp - x0) * e0y
        l1, r1 = (yp - y1) * e1x, (xp - x1) * e1y
        l2, r2 = (yp - y2) * e2x, (xp - x2) * e2y
        inside = (~(l0 < r0) & ~(l1 < r1) & ~(l2 < r2)) | (~(l0 > r0) & ~(l1 > r1) & ~(l2 > r2))
        area = e0x * (y2 - y0) - e0y * (x2 - x0)
        if area != 0:
            w0 = ((xp - x1) * (y2 - y1) - (yp - y1) * (x2 - x1)) / area
            w1 = ((xp - x2) * (y0 - y2) - (yp - y2) * (x0 - x2)) / area
            w2 = F32(1) - w0 - w1
            w0, w1, w2 = np.clip(w0, 0, 1), np.clip(w1, 0, 1), np.clip(w2, 0, 1)
            with np.errstate(divide='ignore', invalid='ignore'):
                zp = (w0 + w1 + w2) / (w0 / z0 + w1 / z1 + w2 / z2)
            inside &= (zp > F32(0.1)) & (zp < F32(100))
        rows = is_ - 1 - np.arange(q0, q1 + 1)
        cover[rows[:, None], np.arange(c0, c1 + 1)[None, :]] |= inside
    c = cover.astype(F32)
    return F32(0.25) * (c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2])


def silhouette_metric(pred, target):
    """`silhouette_iou` of the tracker: -sum [p ln(t + 1e-6) + (1 - p) ln(1 - t + 1e-6)] (fp32 terms, float64 sum)."""
    p, t = pred.astype(F32), target.astype(F32)
    term = p * np.log(t + F32(1e-6)) + (F32(1) - p) * np.log(F32(1) - t + F32(1e-6))
    return float(-np.sum(term.astype(np.float64)))


def select_best(l2e, sil):
    """Index of the iterate the reference ends up saving (`PR/player_recon.py:1259-1271`): both tracked metrics <= best."""
    best, bl, bs = 0, np.inf, np.inf
    for it in range(len(l2e)):
        if not (l2e[it] > bl) and not (sil[it] > bs):
            best, bl, bs = it, l2e[it], sil[it]
    return best
