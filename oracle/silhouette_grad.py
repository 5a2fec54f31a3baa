"""Oracle: the DIFFERENTIABLE half of the silhouette term -- neural_renderer's silhouette rasteriser with its
approximate backward pass, and the fit that puts the silhouette into the loss (`single_view_optimization_test`,
`PR/player_recon.py:1429-1557`).  Test infrastructure only.

PARITY UNPINNED.  `neural_renderer` (`nr.Renderer(camera_mode='projection', ...)`, `PR/player_recon.py:1075-1080`) is an
un-vendored, un-pinned dependency (`PR/requirements.txt`) that is not installed here, and the reference holds no outputs
of it.  What follows restates the published implementation (Kato et al., "Neural 3D Mesh Renderer", CVPR 2018; the PyTorch
port's `rasterize_cuda_kernel.cu`: `forward_face_index_map`, `forward_texture_sampling` for alpha, `backward_pixel_map`):

  forward   faces doubled by fill_back (each face also with its vertex order reversed); a copy takes part iff it is not
            back-facing, (y2-y0)(x1-x0) >= (y1-y0)(x2-x0); per sample of the 2x super-sampled grid (is = 1024) the covering
            copy with the smallest perspective-correct depth zp in (near, far) owns the sample (`face_index_map`), barycentric
            weights from the inverse of the face's pixel-coordinate matrix, clamped to [0,1] and renormalised;
            alpha = 1 on owned samples; vertical flip; 2 x 2 average -> [512, 512].
  backward  per front-facing copy, per edge (p0, p1; p2 the third vertex), per axis (0: columns d0 = x, scan rows d1 = y;
            1: the transpose): for every integer d0 under the edge, the edge crosses at d1_cross; "in" = the pixel on the
            face's side, "out" = its neighbour across the edge.
              out-part (only if the in-pixel is OWNED by this copy): every pixel from the out-pixel to the image border whose
                  alpha differs from alpha_in in the direction the loss asks for, diff = (alpha[d1] - alpha_in) * grad[d1] > 0,
                  pulls the edge's two vertices along the scan axis by -diff / dist, dist = the (signed, eps-padded) distance
                  the vertex would have to move for the edge to reach that pixel;
              in-part: the same for the pixels between the in-pixel and the face's far boundary that this copy owns, against
                  alpha_out.
            Only x / y receive gradient, z none (silhouettes).  eps = 1e-4 (rasterize.py DEFAULT_EPS), near / far = 0.1 / 100.
  The gradient then flows through the 2 x 2 average, the flip, the vertex gather and the projection by ordinary autograd.

Loss and optimiser as the reference: `HomoscedasticUncertaintyWeightedMultiTaskLoss(['joints2D', 'silhouette'],
init_loss_weights={'joints2D': 1.0, 'silhouette': 1e6})` (`PR/player_recon.py:842-872`, `PR/losses/multi_task_loss.py:97-113,
132-144`), both log-variances IN the Adam parameter list (`:1453`), 50 epochs, lr 1e-3; the camera translation handed to the
renderer is recomputed from cam_wp after every step (`:1539`), so from the second epoch on the silhouette loss also reaches the
camera (in the first epoch the translation was computed under no_grad, `:1414-1427`)."""
import numpy as np
import torch
from torch import optim

from .fit import _project17
from .geometry import (FOCAL_LENGTH, PROXY_REP_INPUT_WH, REGRESSOR_IMG_WH, joints2d_l2e, wp_to_translation)
from .silhouette import silhouette_metric

F32 = np.float32
NEAR, FAR, EPS = F32(0.1), F32(100.0), F32(1e-4)


def _to_px(v, is_):
    return F32(0.5) * (v * F32(is_) + F32(is_) - F32(1))


def rasterize_face_index(fx, fy, fz, is_):
    """fx, fy, fz: [NF,3] fp32 normalised x / y (y up) and depth of the DOUBLED face list -> face_index_map [is, is] int32
    (row = yi, unflipped; -1 = empty)."""
    NF = fx.shape[0]
    fis = F32(is_)
    index = np.full((is_, is_), -1, dtype=np.int32)
    depth = np.full((is_, is_), np.inf, dtype=F32)
    back = (fy[:, 2] - fy[:, 0]) * (fx[:, 1] - fx[:, 0]) < (fy[:, 1] - fy[:, 0]) * (fx[:, 2] - fx[:, 0])
    px, py = _to_px(fx, is_), _to_px(fy, is_)
    for fn in np.nonzero(~back)[0]:
        x0, x1, x2 = fx[fn]
        y0, y1, y2 = fy[fn]
        xmin, xmax, ymin, ymax = min(x0, x1, x2), max(x0, x1, x2), min(y0, y1, y2), max(y0, y1, y2)
        if not (xmax >= -1 and xmin <= 1 and ymax >= -1 and ymin <= 1):
            continue
        c0 = max(int(np.floor((xmin * fis + fis - F32(1)) * F32(0.5))) - 1, 0)
        c1 = min(int(np.ceil((xmax * fis + fis - F32(1)) * F32(0.5))) + 1, is_ - 1)
        q0 = max(int(np.floor((ymin * fis + fis - F32(1)) * F32(0.5))) - 1, 0)
        q1 = min(int(np.ceil((ymax * fis + fis - F32(1)) * F32(0.5))) + 1, is_ - 1)
        if c1 < c0 or q1 < q0:
            continue
        xi = np.arange(c0, c1 + 1, dtype=F32)[None, :]
        yi = np.arange(q0, q1 + 1, dtype=F32)[:, None]
        xp = (F32(2) * xi + F32(1) - fis) / fis
        yp = (F32(2) * yi + F32(1) - fis) / fis
        inside = ~(((yp - y0) * (x1 - x0) < (xp - x0) * (y1 - y0)) | ((yp - y1) * (x2 - x1) < (xp - x1) * (y2 - y1)) |
                   ((yp - y2) * (x0 - x2) < (xp - x2) * (y0 - y2)))
        if not inside.any():
            continue
        p0x, p1x, p2x = px[fn]
        p0y, p1y, p2y = py[fn]
        den = p2x * (p0y - p1y) + p0x * (p1y - p2y) + p1x * (p2y - p0y)
        if den == 0:
            continue
        inv = [((p1y - p2y) / den, (p2x - p1x) / den, (p1x * p2y - p2x * p1y) / den),
               ((p2y - p0y) / den, (p0x - p2x) / den, (p2x * p0y - p0x * p2y) / den),
               ((p0y - p1y) / den, (p1x - p0x) / den, (p0x * p1y - p1x * p0y) / den)]
        w = [np.clip(a * xi + b * yi + c, F32(0), F32(1)).astype(F32) for a, b, c in inv]
        ws = w[0] + w[1] + w[2]
        with np.errstate(divide='ignore', invalid='ignore'):
            w = [wk / ws for wk in w]
            zp = (F32(1) / (w[0] / fz[fn, 0] + w[1] / fz[fn, 1] + w[2] / fz[fn, 2])).astype(F32)
        ok = inside & (zp > NEAR) & (zp < FAR)
        sub_d = depth[q0:q1 + 1, c0:c1 + 1]
        take = ok & (zp < sub_d)
        sub_d[take] = zp[take]
        index[q0:q1 + 1, c0:c1 + 1][take] = fn
    return index


def backward_pixel_map(fx, fy, index, grad_alpha, is_):
    """fx, fy [NF,3]; index [is,is] (row = yi); grad_alpha [is,is] (same orientation) -> (gx, gy) [NF,3] fp32."""
    NF = fx.shape[0]
    alpha = (index >= 0).astype(F32)
    gx = np.zeros((NF, 3), dtype=np.float64)
    gy = np.zeros((NF, 3), dtype=np.float64)
    back = (fy[:, 2] - fy[:, 0]) * (fx[:, 1] - fx[:, 0]) < (fy[:, 1] - fy[:, 0]) * (fx[:, 2] - fx[:, 0])
    px, py = _to_px(fx, is_), _to_px(fy, is_)
    owners = np.zeros(NF, dtype=bool)
    owners[np.unique(index[index >= 0])] = True
    two_over_is = F32(2) / F32(is_)
    for fn in np.nonzero(~back)[0]:
        for edge in range(3):
            pi = [(edge + k) % 3 for k in range(3)]
            for axis in range(2):
                if axis == 0:
                    p = [(px[fn, k], py[fn, k]) for k in pi]
                else:
                    p = [(py[fn, k], px[fn, k]) for k in pi]
                (a0, b0), (a1, b1), (a2, b2) = p                  # a = scan-column coordinate (d0), b = scan coordinate (d1)
                if axis == 0:
                    direction = -1 if a0 < a1 else 1
                else:
                    direction = 1 if a0 < a1 else -1
                d0_from = max(int(np.ceil(min(a0, a1))), 0)
                d0_to = int(min(max(a0, a1), F32(is_ - 1)))
                for d0 in range(d0_from, d0_to + 1):
                    fd0 = F32(d0)
                    d1_cross = (b1 - b0) / (a1 - a0) * (fd0 - a0) + b0
                    if not np.isfinite(d1_cross):
                        continue
                    d1_in = int(np.floor(d1_cross)) if direction > 0 else int(np.ceil(d1_cross))
                    d1_out = d1_in + direction
                    if d1_in < 0 or d1_in >= is_ or d1_out < 0 or d1_out >= is_:
                        continue
                    col_a = alpha[:, d0] if axis == 0 else alpha[d0, :]
                    col_g = grad_alpha[:, d0] if axis == 0 else grad_alpha[d0, :]
                    col_i = index[:, d0] if axis == 0 else index[d0, :]
                    alpha_in, alpha_out = col_a[d1_in], col_a[d1_out]

                    def pull(d1s, diff):
                        sel = diff > 0
                        if not sel.any():
                            return
                        d1v, df = d1s[sel].astype(F32), diff[sel]
                        tgt = gy if axis == 0 else gx
                        if a1 != fd0:
                            dist = (a1 - a0) / (a1 - fd0) * (d1v - d1_cross) * two_over_is
                            dist = np.where(dist > 0, dist + EPS, dist - EPS).astype(F32)
                            tgt[fn, pi[0]] -= float(np.sum((df / dist).astype(np.float64)))
                        if a0 != fd0:
                            dist = (a1 - a0) / (fd0 - a0) * (d1v - d1_cross) * two_over_is
                            dist = np.where(dist > 0, dist + EPS, dist - EPS).astype(F32)
                            tgt[fn, pi[1]] -= float(np.sum((df / dist).astype(np.float64)))

                    # out-part
                    if owners[fn] and col_i[d1_in] == fn:
                        d1_limit = is_ - 1 if direction > 0 else 0
                        lo, hi = max(min(d1_out, d1_limit), 0), min(max(d1_out, d1_limit), is_ - 1)
                        d1s = np.arange(lo, hi + 1)
                        pull(d1s, (col_a[lo:hi + 1] - alpha_in) * col_g[lo:hi + 1])
                    # in-part
                    if owners[fn]:
                        if (fd0 - a0) * (fd0 - a2) < 0:
                            cross2 = (b2 - b0) / (a2 - a0) * (fd0 - a0) + b0
                        else:
                            cross2 = (b1 - b2) / (a1 - a2) * (fd0 - a2) + b2
                        if not np.isfinite(cross2):
                            continue
                        d1_limit = int(np.ceil(cross2)) if direction > 0 else int(np.floor(cross2))
                        lo, hi = max(min(d1_in, d1_limit), 0), min(max(d1_in, d1_limit), is_ - 1)
                        if hi >= lo:
                            d1s = np.arange(lo, hi + 1)
                            diff = np.where(col_i[lo:hi + 1] == fn, (col_a[lo:hi + 1] - alpha_out) * col_g[lo:hi + 1], F32(0))
                            pull(d1s, diff)
    return gx.astype(F32), gy.astype(F32)


class _RasterizeSilhouette(torch.autograd.Function):
    """faces [NF,3,3] (x_n, y_n, z per vertex, doubled list) -> alpha [is,is], row = yi (before the vertical flip)."""

    @staticmethod
    def forward(ctx, faces, is_):
        f = faces.detach().numpy().astype(F32)
        index = rasterize_face_index(f[:, :, 0], f[:, :, 1], f[:, :, 2], is_)
        ctx.save_for_backward(faces)
        ctx.index, ctx.is_ = index, is_
        return torch.from_numpy((index >= 0).astype(F32))

    @staticmethod
    def backward(ctx, grad_alpha):
        (faces,) = ctx.saved_tensors
        f = faces.detach().numpy().astype(F32)
        gx, gy = backward_pixel_map(f[:, :, 0], f[:, :, 1], ctx.index, grad_alpha.numpy().astype(F32), ctx.is_)
        g = np.zeros_like(f)
        g[:, :, 0], g[:, :, 1] = gx, gy
        return torch.from_numpy(g), None


def project_torch(verts, translation, focal=FOCAL_LENGTH, wh=PROXY_REP_INPUT_WH):
    """neural_renderer's `projection` (R = I, no distortion) + the [-1,1] normalisation, differentiable: [V,3], [3] -> [V,3]."""
    v = verts + translation[None, :]
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    xd, yd = x / (z + 1e-9), y / (z + 1e-9)
    half = 0.5 * wh
    # K applied as ONE fused multiply-add per coordinate (oracle/silhouette.py `project` and the kernels do the same): through
    # float64, rounded once -- coverage decisions and the backward pass's pixel distances hang on the last bit of these
    u = (xd.double() * focal + half).float()
    vv = wh - (yd.double() * focal + half).float()
    return torch.stack([2 * (u - half) / wh, 2 * (vv - half) / wh, z], dim=-1)


def render_silhouette_torch(verts, faces_idx, translation, focal=FOCAL_LENGTH, wh=PROXY_REP_INPUT_WH, return_index=False):
    """Differentiable `nmr(vertices, faces, t=translation, mode='silhouettes')` for ONE mesh: [V,3], [F,3] int, [3] -> [wh,wh]."""
    fi = torch.as_tensor(np.asarray(faces_idx), dtype=torch.long)
    fi2 = torch.cat([fi, fi.flip(-1)], dim=0)                       # fill_back
    proj = project_torch(verts, translation, focal, wh)
    faces = proj[fi2]                                               # vertices_to_faces
    alpha = _RasterizeSilhouette.apply(faces, 2 * wh)
    img = torch.nn.functional.avg_pool2d(alpha.flip(0)[None, None], kernel_size=2, stride=2)[0, 0]
    return img


SIL_LOSS_WEIGHT = 1e6                                               # PR/player_recon.py:851


def multitask_loss(pred_j2d, label_j2d, pred_sil, label_sil, lv_j2d, lv_sil):
    """`HomoscedasticUncertaintyWeightedMultiTaskLoss.forward` with losses_on = ['joints2D', 'silhouette']."""
    lab = (2.0 * label_j2d) / REGRESSOR_IMG_WH - 1.0
    pred = (2.0 * pred_j2d) / REGRESSOR_IMG_WH - 1.0
    total = torch.nn.functional.mse_loss(pred, lab, reduction='mean') * torch.exp(-lv_j2d) + lv_j2d
    sil = torch.sum(pred_sil * torch.log(label_sil + 1e-6) + (1 - pred_sil) * torch.log(1 - label_sil + 1e-6))
    total = total + (-sil * torch.exp(-lv_sil) + lv_sil)
    return total, sil


def fit_pose_shape_silhouette(smpl, faces_idx, body_pose, global_orient, betas, joints2D, silhouette, cam_wp, lr=1e-3, iters=50,
                              record=None):
    """`single_view_optimization_test` (`PR/player_recon.py:1429-1557`) for ONE player: parameters {global_orient, body-19,
    cam_wp, betas, joints2D_log_var, silhouette_log_var}.  Returns dict(last=(body, G, betas, cam, lv_j2d, lv_sil), ...)."""
    body_pose = body_pose.detach().clone()
    global_orient = global_orient.detach().clone()
    betas = betas.detach().clone()
    cam_wp = cam_wp.detach().clone()
    opt_body = torch.cat([body_pose[:, :6], body_pose[:, 8:21]], dim=1).detach().clone()
    lv_j2d = torch.tensor(float(-np.log(1.0 + 1e-6)), dtype=torch.float32)
    lv_sil = torch.tensor(float(-np.log(SIL_LOSS_WEIGHT + 1e-6)), dtype=torch.float32)
    with torch.no_grad():
        translation = wp_to_translation(cam_wp)                    # :1414-1427: outside the graph in the first epoch
    for p in (global_orient, betas, opt_body, cam_wp, lv_j2d, lv_sil):
        p.requires_grad = True
    # criterion.parameters(): joints2D_log_var, silhouette_log_var in registration order (multi_task_loss.py:60-84)
    optimiser = optim.Adam([global_orient, opt_body, cam_wp, betas, lv_j2d, lv_sil], lr=lr)
    target = torch.from_numpy(np.asarray(joints2D)[:, :2]).float()
    sil_t = torch.from_numpy(np.asarray(silhouette)).float()

    def assemble():
        return torch.cat([opt_body[:, :6], body_pose[:, 6:8], opt_body[:, 6:19], body_pose[:, 21:]], dim=1)

    for epoch in range(1, iters + 1):
        out = smpl(body_pose=assemble(), global_orient=global_orient, betas=betas, pose2rot=False)
        pred = _project17(out, cam_wp)
        sil = render_silhouette_torch(out.vertices[0], faces_idx, translation[0])
        optimiser.zero_grad()
        loss, sil_sum = multitask_loss(pred[0], target, sil, sil_t, lv_j2d, lv_sil)
        if record is not None:
            record.setdefault('loss', []).append(float(loss.item()))
            record.setdefault('l2e', []).append(joints2d_l2e(pred[0].detach().numpy(), target.numpy()))
            record.setdefault('sil_metric', []).append(silhouette_metric(sil.detach().numpy(), np.asarray(silhouette)))
        loss.backward()
        if record is not None and epoch <= 2:
            record[f'grad_G_{epoch}'] = global_orient.grad.detach().numpy().copy()
            record[f'grad_body_{epoch}'] = opt_body.grad.detach().numpy().copy()
            record[f'grad_cam_{epoch}'] = cam_wp.grad.detach().numpy().copy()
            record[f'grad_betas_{epoch}'] = betas.grad.detach().numpy().copy()
            record[f'grad_lv_{epoch}'] = np.array([lv_j2d.grad.item(), lv_sil.grad.item()], np.float32)
        optimiser.step()
        translation = wp_to_translation(cam_wp)                    # :1539 -- inside the graph from now on
        if record is not None:
            with torch.no_grad():
                record.setdefault('body', []).append(assemble().numpy().copy())
                record.setdefault('G', []).append(global_orient.detach().numpy().copy())
                record.setdefault('betas', []).append(betas.detach().numpy().copy())
                record.setdefault('cam', []).append(cam_wp.detach().numpy().copy())
                record.setdefault('lv', []).append(np.array([lv_j2d.item(), lv_sil.item()], np.float32))
    with torch.no_grad():
        return {'last': (assemble().clone(), global_orient.detach().clone(), betas.detach().clone(), cam_wp.detach().clone(),
                         float(lv_j2d.item()), float(lv_sil.item()))}
