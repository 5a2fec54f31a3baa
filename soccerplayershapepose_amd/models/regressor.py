"""Mirror of `PR/models/regressor.py`: SingleInputRegressor = ResNet-18 encoder + IEF, on gfx950.

    regressor = SingleInputRegressor(resnet_in_channels=18, resnet_layers=18, ief_iters=3)
    regressor.load_state_dict(checkpoint['best_model_state_dict'])     # 132-key schema, SURVEY Appendix C
    regressor.to(device); regressor.eval()
    cam, pose6d, shape = regressor(proxy_rep)                            # [B,18,256,256] fp32 NCHW

Eval-mode BatchNorm is folded into the conv weights inside `spsp_regressor_load`; the forward is
`spsp_regressor_forward` (csrc/regressor.hip, csrc/conv_*.hip)."""
import ctypes as C
import os

import numpy as np
import torch

from .. import _abi, config
from .. import runtime as rt


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


class SingleInputRegressor:
    def __init__(self, resnet_in_channels=18, resnet_layers=18, ief_iters=3, mean_params=None, max_batch=32,
                 precision='f16', plan='latency'):
        if resnet_in_channels != 18 or resnet_layers != 18:
            raise NotImplementedError('only the configuration the reference ever instantiates is built: '
                                      'SingleInputRegressor(18, 18, 3) (PR/player_recon.py:134-136)')
        self.ief_iters = ief_iters
        self.max_batch = max_batch
        # launch plan (include/spsp.h): 'latency' (default) for one batch in flight -- chip-filling grids; 'throughput' for several
        # batches in flight on different streams (PlayerPipeline selects it for its slots): narrow persistent grids, least CU time
        if plan not in ('throughput', 'latency'):
            raise ValueError(f"plan must be 'throughput' or 'latency', got {plan!r}")
        self.plan = plan
        # storage / MFMA element type of the backbone (include/spsp.h): 'f16' (default; precision 1) keeps the END-TO-END
        # vertices within 1e-3 m of the all-fp32 reference path; 'bf16' (precision 0) has the same bytes and MFMA rate but
        # 3 mantissa bits fewer (~1.5e-3 m)
        if precision not in ('bf16', 'f16'):
            raise ValueError(f"precision must be 'bf16' or 'f16', got {precision!r}")
        self.precision = precision
        self._prec = _abi.PRECISION_F16 if precision == 'f16' else _abi.PRECISION_BF16
        self.native_dtype = torch.float16 if precision == 'f16' else torch.bfloat16
        self._sd = None
        self._handle = None
        self._rt = None
        if mean_params is None and os.path.exists(config.SMPL_MEAN_PARAMS_PATH):
            mean_params = np.load(config.SMPL_MEAN_PARAMS_PATH)         # PR/models/ief_module.py:31-35
        self.initial_params_estimate = None
        if mean_params is not None:
            self.set_mean_params(mean_params)

    def set_mean_params(self, mean_params):
        """`IEFModule.load_mean_params_6d_pose` (`PR/models/ief_module.py:33-46`)."""
        mp = np.zeros(3 + 24 * 6 + 10, dtype=np.float32)
        mp[3:] = np.concatenate((np.asarray(mean_params['pose']), np.asarray(mean_params['shape'])))
        mp[0] = 0.9
        self.initial_params_estimate = mp
        self._drop()

    # -- reference-style plumbing ---------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        lib = _abi.load()
        need = []
        for i in range(_abi.NUM_CONVS):
            need.append(lib.spsp_conv_name(i).decode() + '.weight')
            bn = lib.spsp_conv_bn_name(i).decode()
            need += [f'{bn}.{k}' for k in ('weight', 'bias', 'running_mean', 'running_var')]
        need += [f'ief_module.fc{n}.{k}' for n in (1, 2, 3) for k in ('weight', 'bias')]
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError(f'load_state_dict: missing keys {missing[:4]}{"..." if len(missing) > 4 else ""}')
        self._sd = {k: np.ascontiguousarray(_np(state_dict[k]), dtype=np.float32) for k in need}
        shapes = {'image_encoder.conv1.weight': (64, 18, 7, 7), 'ief_module.fc1.weight': (512, 669),
                  'ief_module.fc2.weight': (512, 512), 'ief_module.fc3.weight': (157, 512)}
        for k, shp in shapes.items():
            if self._sd[k].shape != shp:
                raise ValueError(f'load_state_dict: {k} has shape {self._sd[k].shape}, expected {shp}')
        self._drop()
        return self

    def to(self, device):
        self._load(rt.get(device))
        return self

    def eval(self):
        return self

    def fix(self):                       # PR/models/regressor.py:50-56 (training-time switch; BN is always eval here)
        return self

    def _drop(self):
        if self._handle is not None:
            self._rt.lib.spsp_regressor_free(self._handle)
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
            raise RuntimeError('SingleInputRegressor: call load_state_dict() before the first forward')
        if self.initial_params_estimate is None:
            raise RuntimeError('SingleInputRegressor: no mean parameters (pass mean_params= or provide '
                               f'{config.SMPL_MEAN_PARAMS_PATH})')
        lib = r.lib
        w = _abi.RegressorWeights()
        keep = []
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        for i in range(_abi.NUM_CONVS):
            conv, bn = lib.spsp_conv_name(i).decode(), lib.spsp_conv_bn_name(i).decode()
            w.conv[i].weight = fp(self._sd[conv + '.weight'])
            w.conv[i].bn_weight = fp(self._sd[bn + '.weight'])
            w.conv[i].bn_bias = fp(self._sd[bn + '.bias'])
            w.conv[i].bn_mean = fp(self._sd[bn + '.running_mean'])
            w.conv[i].bn_var = fp(self._sd[bn + '.running_var'])
        for j in range(3):
            w.fc_weight[j] = fp(self._sd[f'ief_module.fc{j + 1}.weight'])
            w.fc_bias[j] = fp(self._sd[f'ief_module.fc{j + 1}.bias'])
        init = np.ascontiguousarray(self.initial_params_estimate, dtype=np.float32)
        keep.append(init)
        w.init_params = fp(init)
        w.ief_iters = self.ief_iters
        h = C.c_void_p()
        self._drop()
        with torch.cuda.device(r.device):
            r.check(lib.spsp_regressor_load(r.ctx, C.byref(w), self.max_batch, self._prec, C.byref(h)), 'spsp_regressor_load')
        self._handle, self._rt = h, r
        self.set_plan(self.plan)

    # -- forward ------------------------------------------------------------------------------------
    def pack(self, proxy_nchw):
        """NCHW fp32 [B,18,256,256] -> backbone-native [B,262,264,18] bf16 / fp16 (zero border) device tensor."""
        x = rt.dev_f32(proxy_nchw)
        r = rt.get(x.device)
        if x.dim() != 4 or tuple(x.shape[1:]) != (18, 256, 256):
            raise ValueError(f'proxy representation must be [B,18,256,256], got {tuple(x.shape)}')
        out = torch.empty((x.shape[0], 262, 264, 18), dtype=self.native_dtype, device=x.device)
        r.check(r.lib.spsp_proxy_pack(r.ctx, rt.ptr(x), x.shape[0], self._prec, rt.ptr(out), r.stream()), 'spsp_proxy_pack')
        return out

    def prepare_input(self, input):
        """Validated device tensor + proxy_format for spsp_regressor_forward / spsp_pipeline."""
        native = isinstance(input, torch.Tensor) and input.dtype in (torch.bfloat16, torch.float16)
        if native and input.dtype != self.native_dtype:
            raise ValueError(f'native proxy dtype {input.dtype} does not match this regressor (precision={self.precision!r}: {self.native_dtype})')
        if native:
            if input.dim() != 4 or tuple(input.shape[1:]) != (262, 264, 18) or not input.is_cuda:
                raise ValueError(f'native proxy must be a CUDA {self.precision} [B,262,264,18] tensor, got {tuple(input.shape)}')
            x = input.contiguous()
        else:
            x = rt.dev_f32(input)
            if x.dim() != 4 or tuple(x.shape[1:]) != (18, 256, 256):
                raise ValueError(f'proxy representation must be [B,18,256,256], got {tuple(x.shape)}')
        self._load(rt.get(x.device))
        if x.shape[0] > self.max_batch:
            raise ValueError(f'batch {x.shape[0]} exceeds max_batch {self.max_batch} given at construction')
        return x, (_abi.PROXY_NATIVE if native else _abi.PROXY_NCHW_F32)

    @property
    def handle(self):
        return self._handle

    def set_plan(self, plan):
        if plan not in ('throughput', 'latency'):
            raise ValueError(f"plan must be 'throughput' or 'latency', got {plan!r}")
        self.plan = plan
        if self._handle is not None:
            self._rt.check(self._rt.lib.spsp_regressor_set_plan(self._handle, _abi.PLAN_LATENCY if plan == 'latency' else _abi.PLAN_THROUGHPUT),
                           'spsp_regressor_set_plan')
        return self

    def check_status(self):
        """Raises RuntimeError if a kernel of an earlier forward on this handle reported a device-side failure (the fused
        IEF's bounded hand-off wait; its outputs are NaN).  No synchronisation: a host-visible status word is read.  Every
        forward checks it on entry; callers that replay a captured HIP graph (no library call) check it themselves."""
        if self._handle is not None:
            self._rt.check(self._rt.lib.spsp_regressor_status(self._handle), 'spsp_regressor_status')

    def forward(self, input, return_feats=False):
        native = isinstance(input, torch.Tensor) and input.dtype in (torch.bfloat16, torch.float16)
        if native and input.dtype != self.native_dtype:
            raise ValueError(f'native proxy dtype {input.dtype} does not match this regressor (precision={self.precision!r}: {self.native_dtype})')
        if native:
            if input.dim() != 4 or tuple(input.shape[1:]) != (262, 264, 18) or not input.is_cuda:
                raise ValueError(f'native proxy must be a CUDA {self.precision} [B,262,264,18] tensor, got {tuple(input.shape)}')
            x = input.contiguous()
        else:
            x = rt.dev_f32(input)
            if x.dim() != 4 or tuple(x.shape[1:]) != (18, 256, 256):
                raise ValueError(f'proxy representation must be [B,18,256,256], got {tuple(x.shape)}')
        r = rt.get(x.device)
        self._load(r)
        B = x.shape[0]
        if B > self.max_batch:
            raise ValueError(f'batch {B} exceeds max_batch {self.max_batch} given at construction')
        cam = torch.empty((B, 3), dtype=torch.float32, device=x.device)
        pose = torch.empty((B, 144), dtype=torch.float32, device=x.device)
        shape = torch.empty((B, 10), dtype=torch.float32, device=x.device)
        feats = torch.empty((B, 512), dtype=torch.float32, device=x.device) if return_feats else None
        fmt = _abi.PROXY_NATIVE if native else _abi.PROXY_NCHW_F32
        r.check(r.lib.spsp_regressor_forward(r.ctx, self._handle, rt.ptr(x), fmt, B, rt.ptr(cam), rt.ptr(pose),
                                             rt.ptr(shape), rt.ptr(feats), r.stream()), 'spsp_regressor_forward')
        return (cam, pose, shape, feats) if return_feats else (cam, pose, shape)

    __call__ = forward

    def read_stage(self, stage, B):
        """Parity tap: stage 1 maxpool, 2..5 layer1..4 of the last forward, as fp32 NCHW (stage 0, conv1+relu, is never
        materialised: conv1 and the max-pool are one kernel -> ValueError)."""
        S, Cc = [(128, 64), (64, 64), (64, 64), (32, 128), (16, 256), (8, 512)][stage]
        r = self._rt
        out = torch.empty((B, Cc, S, S), dtype=torch.float32, device=r.device)
        r.check(r.lib.spsp_regressor_read_stage(r.ctx, self._handle, stage, B, rt.ptr(out), r.stream()),
                'spsp_regressor_read_stage')
        return out
