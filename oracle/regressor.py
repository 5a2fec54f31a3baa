"""Oracle: ResNet-18 + IEF regressor.  Test infrastructure only.

Restates `ResNet.forward` (`PR/models/resnet.py:201-216`), `BasicBlock.forward` (:61-77),
`resnet18(in_channels=18)` (:228-236), `IEFModule.forward` (`PR/models/ief_module.py:48-64`)
and `SingleInputRegressor.forward` (`PR/models/regressor.py:43-48`) functionally on a plain
state dict (SURVEY Appendix C schema), eval-mode BatchNorm (`regressor.eval()`,
`PR/predict/predict_3D.py:129`).  Pinned against the imported reference module by
`oracle/gen_golden.py` -> `tests/golden/regressor.npz`.

`emulate_bf16=True` additionally rounds the BN-folded weights and every stored activation to
bfloat16 at exactly the points where the HIP backbone does (fp32 accumulation), so the MFMA
kernels can be checked to ~1e-3 instead of against the much looser bf16-vs-fp32 gap.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default [EXT]


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _f16(x):
    return x.to(torch.float16).to(torch.float32)


def _quantiser(emulate):
    """emulate: False/None = fp32; True or 'bf16' = the HIP backbone's default storage type; 'f16' = its precision-1 mode."""
    if not emulate:
        return None
    return _f16 if emulate == 'f16' else _bf16


def fold_bn(sd, conv, bn):
    """Eval-mode BN folded into the preceding bias-free conv: returns (w', b') fp32."""
    w = _t(sd[conv + '.weight']).double()
    g, b = _t(sd[bn + '.weight']).double(), _t(sd[bn + '.bias']).double()
    m, v = _t(sd[bn + '.running_mean']).double(), _t(sd[bn + '.running_var']).double()
    s = g / torch.sqrt(v + BN_EPS)
    return (w * s[:, None, None, None]).float(), (b - m * s).float()


def _conv_bn(sd, x, conv, bn, stride, pad, emulate_bf16):
    qz = _quantiser(emulate_bf16)
    if qz is not None:
        w, b = fold_bn(sd, conv, bn)
        return F.conv2d(x, qz(w), b, stride=stride, padding=pad)
    y = F.conv2d(x, _t(sd[conv + '.weight']), None, stride=stride, padding=pad)
    return F.batch_norm(y, _t(sd[bn + '.running_mean']), _t(sd[bn + '.running_var']),
                        _t(sd[bn + '.weight']), _t(sd[bn + '.bias']), False, 0.0, BN_EPS)


def encoder_forward(sd, x, emulate_bf16=False, return_stages=False):
    """[B,18,256,256] -> [B,512] pooled features (`PR/models/resnet.py:201-216`)."""
    q = _quantiser(emulate_bf16) or (lambda t: t)
    stages = {}
    x = q(_t(x).float())
    p = 'image_encoder.'
    x = q(F.relu(_conv_bn(sd, x, p + 'conv1', p + 'bn1', 2, 3, emulate_bf16)))
    stages['conv1'] = x
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    stages['maxpool'] = x
    for li in range(1, 5):
        for b in range(2):
            blk = f'{p}layer{li}.{b}.'
            stride = 2 if (li > 1 and b == 0) else 1
            identity = x
            out = q(F.relu(_conv_bn(sd, x, blk + 'conv1', blk + 'bn1', stride, 1, emulate_bf16)))
            out = _conv_bn(sd, out, blk + 'conv2', blk + 'bn2', 1, 1, emulate_bf16)
            if (blk + 'downsample.0.weight') in sd:
                identity = q(_conv_bn(sd, x, blk + 'downsample.0', blk + 'downsample.1', stride, 0, emulate_bf16))
            pre = F.relu(out + identity)
            x = q(pre)
        stages[f'layer{li}'] = x
    # the HIP backbone fuses the global average pool into the last conv's epilogue: it averages the fp32
    # values BEFORE they are rounded to bf16 for storage (`pre`; identical to x in fp32 mode)
    feats = torch.flatten(F.adaptive_avg_pool2d(pre, (1, 1)), 1)
    return (feats, stages) if return_stages else feats


def initial_params(mean_params):
    """`IEFModule.load_mean_params_6d_pose` (`PR/models/ief_module.py:33-46`): [0.9,0,0 | pose | shape]."""
    mp = np.zeros(3 + 24 * 6 + 10)
    mp[3:] = np.concatenate((mean_params['pose'], mean_params['shape']))
    mp[0] = 0.9
    return torch.from_numpy(mp.astype(np.float32)).float()


def ief_forward(sd, feats, mean_params, iterations=3):
    """`IEFModule.forward` (`PR/models/ief_module.py:48-64`)."""
    B = feats.shape[0]
    params = initial_params(mean_params).repeat([B, 1])
    w1, b1 = _t(sd['ief_module.fc1.weight']), _t(sd['ief_module.fc1.bias'])
    w2, b2 = _t(sd['ief_module.fc2.weight']), _t(sd['ief_module.fc2.bias'])
    w3, b3 = _t(sd['ief_module.fc3.weight']), _t(sd['ief_module.fc3.bias'])
    state = torch.cat([feats, params], dim=1)
    for _ in range(iterations):
        h = F.relu(F.linear(state, w1, b1))
        h = F.relu(F.linear(h, w2, b2))
        params = params + F.linear(h, w3, b3)
        state = torch.cat([feats, params], dim=1)
    return params[:, :3], params[:, 3:3 + 144], params[:, 3 + 144:]


@torch.no_grad()
def regressor_forward(sd, x, mean_params, ief_iters=3, emulate_bf16=False):
    """`SingleInputRegressor.forward` (`PR/models/regressor.py:43-48`) -> cam[B,3], pose6d[B,144], shape[B,10]."""
    feats = encoder_forward(sd, x, emulate_bf16=emulate_bf16)
    return ief_forward(sd, feats, mean_params, ief_iters)
