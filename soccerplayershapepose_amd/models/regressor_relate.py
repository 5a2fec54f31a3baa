"""Mirror of `PR/models/regressor_relate.py`: `PoseRelationModule`, the reference's only cross-player operator
(27 relation units over the players of one frame), on gfx950.

    pose_relation = PoseRelationModule()
    pose_relation.load_state_dict(checkpoint['best_model_state_dict'])   # PR/PyTorch3DTest.py:3527-3534
    pose_relation.to(device); pose_relation.eval()
    pred_pose_rotmats = pose_relation([pred_pose_rotmats, boxes_batch])  # [N,24,3,3], [N,4] -> [N,24,3,3]  (:3758-3759)

The arithmetic runs in csrc/relate.hip through `spsp_relate_forward`.  All N players of the frame must be in one
call (the operator couples them), so multi-GPU sharding is by frame when it is used."""
import ctypes as C

import numpy as np
import torch

from .. import _abi
from .. import runtime as rt


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


class PoseRelationModule:
    def __init__(self, max_players=64):
        self.max_players = max_players
        self._sd = None
        self._handle = None
        self._rt = None

    def load_state_dict(self, state_dict, strict=True):
        need = [f'relation_module.relation.{r}.{w}.{k}' for r in range(_abi.NUM_RELATIONS)
                for w in ('WG', 'WK', 'WQ', 'WV') for k in ('weight', 'bias')]
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError(f'load_state_dict: missing keys {missing[:4]}{"..." if len(missing) > 4 else ""}')
        self._sd = {k: np.ascontiguousarray(_np(state_dict[k]), dtype=np.float32) for k in need}
        for k, v in self._sd.items():
            gate, weight = '.WG.' in k, k.endswith('.weight')
            want = ((1, 8) if weight else (1,)) if gate else ((8, 216) if weight else (8,))
            if v.shape != want:
                raise ValueError(f'load_state_dict: {k} has shape {v.shape}, expected {want}')
        self._drop()
        return self

    def to(self, device):
        self._load(rt.get(device))
        return self

    def eval(self):
        return self

    def _drop(self):
        if self._handle is not None:
            self._rt.lib.spsp_relate_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass

    def _load(self, r):
        if self._handle is not None and self._rt is r:
            return
        if self._sd is None:
            raise RuntimeError('PoseRelationModule: call load_state_dict() before the first forward')
        w = _abi.RelateWeights()
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        for i in range(_abi.NUM_RELATIONS):
            for m in ('wg', 'wk', 'wq', 'wv'):
                for k in ('weight', 'bias'):
                    setattr(w.unit[i], f'{m}_{k}', fp(self._sd[f'relation_module.relation.{i}.{m.upper()}.{k}']))
        h = C.c_void_p()
        self._drop()
        with torch.cuda.device(r.device):
            r.check(r.lib.spsp_relate_load(r.ctx, C.byref(w), self.max_players, C.byref(h)), 'spsp_relate_load')
        self._handle, self._rt = h, r

    @property
    def handle(self):
        return self._handle

    def forward(self, input):
        pose_params, bboxes = input
        pose = rt.dev_f32(pose_params)
        r = rt.get(pose.device)
        self._load(r)
        pose = pose.reshape(-1, 24, 3, 3).contiguous()
        boxes = rt.dev_f32(torch.as_tensor(bboxes), r.device).reshape(-1, 4).contiguous()
        n = pose.shape[0]
        if boxes.shape[0] != n:
            raise ValueError(f'PoseRelationModule: {n} poses but {boxes.shape[0]} boxes')
        if n > self.max_players:
            raise ValueError(f'PoseRelationModule: {n} players exceed max_players={self.max_players}')
        out = torch.empty_like(pose)
        r.check(r.lib.spsp_relate_forward(r.ctx, self._handle, rt.ptr(pose), rt.ptr(boxes), n, rt.ptr(out), r.stream()),
                'spsp_relate_forward')
        return out

    __call__ = forward
