"""The on-disk result record of the fit drivers and its use as a warm start.

Reference layout (all fp32, leading batch dimension 1):
  single view   `np.savez(opt_full, body_pose=[1,23,3,3], global_orient=[1,1,3,3], betas=[1,10], translation=[1,3])`
                (`PR/player_recon.py:1293-1294`), read back as the initialisation of the multi-view fit
                (`:1673-1689`; cam_wp is recovered from `translation` with `convert_camera_translation_to_weak_perspective`).
  multi view    one `data.npz{body_pose, betas}` per player + one `<view>.npz{translation, global_orient}` per view
                (`:1976-1979`); `data.npz` is what the broad-view fit starts from (`:2088-2099`).
The functions take / return host numpy arrays or device tensors; nothing here touches the GPU except `warm_start`, which
places the loaded record on the device in the layout `fit_players` / `single_view_optimization` take."""
import os

import numpy as np
import torch

from . import config

_SINGLE_KEYS = ('body_pose', 'global_orient', 'betas', 'translation')
_SHAPES = {'body_pose': (1, 23, 3, 3), 'global_orient': (1, 1, 3, 3), 'betas': (1, 10), 'translation': (1, 3)}


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def _one(name, a):
    a = np.ascontiguousarray(_np(a), dtype=np.float32)
    if a.shape != _SHAPES[name]:
        raise ValueError(f'record field {name}: shape {a.shape}, the reference layout is {_SHAPES[name]} (one player per file)')
    return a


def save_record(path, body_pose, global_orient, betas, translation):
    """Write ONE player's single-view record exactly as `PR/player_recon.py:1293-1294` does (numpy appends `.npz`)."""
    np.savez(path, body_pose=_one('body_pose', body_pose), global_orient=_one('global_orient', global_orient),
             betas=_one('betas', betas), translation=_one('translation', translation))


def save_records(dirname, names, fit):
    """A batch of players (the namespace `single_view_optimization` returns) -> one reference-layout file per player."""
    os.makedirs(dirname, exist_ok=True)
    B = fit.betas.shape[0]
    if len(names) != B:
        raise ValueError(f'{len(names)} names for {B} players')
    bp, go, be, tr = (_np(t) for t in (fit.body_pose, fit.global_orient, fit.betas, fit.translation))
    paths = []
    for i, n in enumerate(names):
        p = os.path.join(dirname, n if n.endswith('.npz') else n + '.npz')
        save_record(p, bp[i:i + 1], go[i:i + 1], be[i:i + 1], tr[i:i + 1])
        paths.append(p)
    return paths


def load_record(path):
    """The four arrays of a single-view record, validated against the reference layout."""
    with np.load(path) as f:
        missing = [k for k in _SINGLE_KEYS if k not in f.files]
        if missing:
            raise KeyError(f'{path}: missing {missing} (single-view record: {_SINGLE_KEYS})')
        return {k: _one(k, f[k]) for k in _SINGLE_KEYS}


def translation_to_cam_wp(translation, focal_length=config.FOCAL_LENGTH, resolution=config.proxy_rep_input_wh):
    """`convert_camera_translation_to_weak_perspective` (`PR/utils/cam_utils.py:36-42`) on host arrays: [tx,ty,tz] -> [s,tx,ty]."""
    t = np.asarray(translation, dtype=np.float32)
    s = np.float32(2.0) * np.float32(focal_length) / (np.float32(resolution) * t[:, 2] + np.float32(1e-9))
    return np.stack([s, t[:, 0], t[:, 1]], axis=-1).astype(np.float32)


def warm_start(records, device):
    """Loaded single-view records (a list of `load_record` dicts, or paths) -> device tensors for the next fit:
    full_pose [B,24,3,3] (joint 0 = global orientation), betas [B,10], cam_wp [B,3] -- the initialisation the
    reference builds at `PR/player_recon.py:1673-1689`."""
    recs = [load_record(r) if isinstance(r, (str, os.PathLike)) else r for r in records]
    full = np.concatenate([np.concatenate([r['global_orient'], r['body_pose']], axis=1) for r in recs], axis=0)
    betas = np.concatenate([r['betas'] for r in recs], axis=0)
    cam = translation_to_cam_wp(np.concatenate([r['translation'] for r in recs], axis=0))
    dev = torch.device(device)
    return (torch.from_numpy(full).to(dev).contiguous(), torch.from_numpy(betas).to(dev).contiguous(),
            torch.from_numpy(cam).to(dev).contiguous())


# ---------------------------------------------------------------- multi-view layout (PR/player_recon.py:1976-1979)
def save_multiview_record(dirname, body_pose, betas, views):
    """`data.npz{body_pose [1,23,3,3], betas [1,10]}` + `<view>.npz{translation [1,3], global_orient [1,1,3,3]}` per view;
    `views` maps a view name (the reference uses the crop's file name, '.png' -> '.npz') to (translation, global_orient)."""
    os.makedirs(dirname, exist_ok=True)
    np.savez(os.path.join(dirname, 'data.npz'), body_pose=_one('body_pose', body_pose), betas=_one('betas', betas))
    for name, (translation, global_orient) in views.items():
        stem = name[:-4] if name.endswith(('.png', '.npz')) else name
        if stem == 'data':
            raise ValueError("a view cannot be called 'data' (that file holds the shared pose)")
        np.savez(os.path.join(dirname, stem + '.npz'), translation=_one('translation', translation),
                 global_orient=_one('global_orient', global_orient))


def load_multiview_record(dirname):
    """-> (body_pose, betas, {view: (translation, global_orient)})."""
    with np.load(os.path.join(dirname, 'data.npz')) as f:
        body_pose, betas = _one('body_pose', f['body_pose']), _one('betas', f['betas'])
    views = {}
    for fn in sorted(os.listdir(dirname)):
        if fn.endswith('.npz') and fn != 'data.npz':
            with np.load(os.path.join(dirname, fn)) as f:
                views[fn[:-4]] = (_one('translation', f['translation']), _one('global_orient', f['global_orient']))
    return body_pose, betas, views
