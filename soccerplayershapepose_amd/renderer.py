"""Mirror of the one renderer call on the fit path: `nmr(vertices=..., faces=..., t=translation_nmr, mode='silhouettes')`
with `nr.Renderer(camera_mode='projection', K, R=I, image_size=512, orig_size=512)` (`PR/player_recon.py:1075-1080,
1226-1229`) and the tracker's "silhouette_iou" metric of the result (`PR/metrics/train_loss_and_metrics_tracker.py:
224-229`) -- csrc/silhouette.hip; `loss_grad` adds the differentiable half (the silhouette LOSS of `single_view_optimization_test`,
`PR/player_recon.py:1429`, and neural_renderer's approximate backward pass) -- csrc/silhouette_grad.hip."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import config
from . import runtime as rt


class SilhouetteRenderer:
    def __init__(self, faces, max_batch=64, image_size=config.proxy_rep_input_wh, focal_length=config.FOCAL_LENGTH):
        f = np.ascontiguousarray(np.asarray(faces).astype(np.int32))
        if f.ndim != 2 or f.shape[1] != 3:
            raise ValueError(f'SilhouetteRenderer: faces must be [F,3], got {f.shape}')
        self._faces, self.max_batch, self.image_size, self.focal_length = f, int(max_batch), int(image_size), float(focal_length)
        self._handle = self._rt = None

    def to(self, device):
        r = rt.get(device)
        if self._handle is not None and self._rt is r:
            return self
        self._drop()
        h = C.c_void_p()
        with torch.cuda.device(r.device):
            r.check(r.lib.spsp_rasterizer_load(r.ctx, self._faces.ctypes.data_as(C.c_void_p), self._faces.shape[0], self.max_batch,
                                               self.image_size, C.byref(h)), 'spsp_rasterizer_load')
        self._handle, self._rt = h, r
        return self

    def _drop(self):
        if self._handle is not None:
            self._rt.lib.spsp_rasterizer_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass

    def __call__(self, vertices, translation, target=None, return_image=True):
        """vertices [B,6890,3], translation [B,3] (or [B,1,3] as the reference passes it) -> namespace(silhouettes
        [B,512,512] fp32 in {0,.25,.5,.75,1}); with target (detected silhouette, 0/1: [B,512,512] or one [512,512] for all)
        also metric [B] = -sum_px [p ln(t + 1e-6) + (1 - p) ln(1 - t + 1e-6)]."""
        v = rt.dev_f32(vertices)
        self.to(v.device)
        r = self._rt
        B = v.shape[0]
        if v.dim() != 3 or tuple(v.shape[1:]) != (6890, 3):
            raise ValueError(f'SilhouetteRenderer: vertices must be [B,6890,3], got {tuple(v.shape)}')
        if B > self.max_batch:
            raise ValueError(f'SilhouetteRenderer: batch {B} exceeds max_batch {self.max_batch}')
        t = rt.dev_f32(translation, r.device).reshape(B, 3).contiguous()
        wh = self.image_size
        img = torch.empty((B, wh, wh), dtype=torch.float32, device=r.device) if return_image else None
        tgt = metric = None
        shared = 0
        if target is not None:
            tgt = torch.as_tensor(target).to(device=r.device, dtype=torch.uint8).contiguous()
            if tuple(tgt.shape) == (wh, wh):
                shared = 1
            elif tuple(tgt.shape) != (B, wh, wh):
                raise ValueError(f'SilhouetteRenderer: target must be [{wh},{wh}] or [B,{wh},{wh}], got {tuple(tgt.shape)}')
            metric = torch.empty((B,), dtype=torch.float32, device=r.device)
        elif not return_image:
            raise ValueError('SilhouetteRenderer: nothing to compute (no target and return_image=False)')
        r.check(r.lib.spsp_render_silhouettes(r.ctx, self._handle, rt.ptr(v), rt.ptr(t), self.focal_length, B, rt.ptr(tgt), shared,
                                              rt.ptr(img), rt.ptr(metric), r.stream()), 'spsp_render_silhouettes')
        return SimpleNamespace(silhouettes=img, metric=metric)

    def loss_grad(self, vertices, translation, target, weight=None, return_image=False):
        """The silhouette term of `HomoscedasticUncertaintyWeightedMultiTaskLoss` (`PR/losses/multi_task_loss.py:132-144`) on the
        rendered silhouette, and its gradient through the restated neural_renderer backward pass.
        vertices [B,6890,3], translation [B,3], target 0/1 ([B,512,512] or [512,512]), weight [B] = exp(-silhouette_log_var)
        (None: value only) -> namespace(sil_sum [B] = sum_px [p ln(t + 1e-6) + (1 - p) ln(1 - t + 1e-6)], grad_vertices [B,6890,3]
        and grad_translation [B,3] of L = -sil_sum * weight, silhouettes [B,512,512] if asked for)."""
        v = rt.dev_f32(vertices)
        self.to(v.device)
        r = self._rt
        B = v.shape[0]
        if v.dim() != 3 or tuple(v.shape[1:]) != (6890, 3):
            raise ValueError(f'SilhouetteRenderer: vertices must be [B,6890,3], got {tuple(v.shape)}')
        if B > self.max_batch:
            raise ValueError(f'SilhouetteRenderer: batch {B} exceeds max_batch {self.max_batch}')
        t = rt.dev_f32(translation, r.device).reshape(B, 3).contiguous()
        wh = self.image_size
        tgt = torch.as_tensor(target).to(device=r.device, dtype=torch.uint8).contiguous()
        shared = 1 if tuple(tgt.shape) == (wh, wh) else 0
        if not shared and tuple(tgt.shape) != (B, wh, wh):
            raise ValueError(f'SilhouetteRenderer: target must be [{wh},{wh}] or [B,{wh},{wh}], got {tuple(tgt.shape)}')
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=r.device)
        sil_sum = f32(B)
        w = gv = gt = None
        if weight is not None:
            w = rt.dev_f32(weight, r.device).reshape(B).contiguous()
            gv, gt = f32(B, 6890, 3), f32(B, 3)
        img = f32(B, wh, wh) if return_image else None
        r.check(r.lib.spsp_silhouette_loss_grad(r.ctx, self._handle, rt.ptr(v), rt.ptr(t), self.focal_length, B, rt.ptr(tgt), shared, rt.ptr(w),
                                                rt.ptr(sil_sum), rt.ptr(gv), rt.ptr(gt), rt.ptr(img), r.stream()), 'spsp_silhouette_loss_grad')
        return SimpleNamespace(sil_sum=sil_sum, grad_vertices=gv, grad_translation=gt, silhouettes=img)

    @property
    def handle(self):
        return self._handle
