"""Where the bf16 backbone loses its 1.5 mm -- CPU only, on the oracle.

The oracle's rounding-emulating encoder (oracle/regressor.py: weights and stored activations rounded where the HIP kernels round
them, fp32 accumulation; the GPU matches it to a few ulp per stage) is run with the element type chosen PER GROUP --
input tensor, conv1 (+pool), layer1..layer4, separately for weights and for stored activations -- and the vertices of
regressor -> SMPL are compared with the all-fp32 path on the bench's crops:
    only-one: that group in bf16, everything else fp32      leave-one-out: everything bf16 except that group (fp16 there)
usage: python tools/precision_breakdown.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from soccerplayershapepose_amd import synthetic as S  # noqa: E402
from oracle import regressor as OR, geometry as G, smpl as OS  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
GROUPS = ['input', 'conv1', 'layer1', 'layer2', 'layer3', 'layer4']
Q = {None: lambda t: t, 'bf16': OR._bf16, 'f16': OR._f16}


def encoder(sd, x, wq, aq):
    """wq / aq: dict group -> None | 'bf16' | 'f16' for the folded weights / the stored activations of that group."""
    def conv(x, conv_, bn, stride, pad, g):
        w, b = OR.fold_bn(sd, conv_, bn)
        return F.conv2d(x, Q[wq[g]](w), b, stride=stride, padding=pad)
    p = 'image_encoder.'
    x = Q[aq['input']](torch.from_numpy(x).float())
    x = Q[aq['conv1']](F.relu(conv(x, p + 'conv1', p + 'bn1', 2, 3, 'conv1')))
    x = F.max_pool2d(x, 3, 2, 1)
    for li in range(1, 5):
        g = f'layer{li}'
        for b in range(2):
            blk = f'{p}layer{li}.{b}.'
            stride = 2 if (li > 1 and b == 0) else 1
            identity = x
            out = Q[aq[g]](F.relu(conv(x, blk + 'conv1', blk + 'bn1', stride, 1, g)))
            out = conv(out, blk + 'conv2', blk + 'bn2', 1, 1, g)
            if (blk + 'downsample.0.weight') in sd:
                identity = Q[aq[g]](conv(x, blk + 'downsample.0', blk + 'downsample.1', stride, 0, g))
            pre = F.relu(out + identity)
            x = Q[aq[g]](pre)
    return torch.flatten(F.adaptive_avg_pool2d(pre, (1, 1)), 1)


@torch.no_grad()
def vertices(sd, model, osmpl, x, wq, aq):
    cam, pose, shape = OR.ief_forward(sd, encoder(sd, x, wq, aq), S.mean_params())
    R = G.rot6d_to_rotmat(pose).view(x.shape[0], 24, 3, 3)
    return osmpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=shape).vertices, torch.cat([cam, pose, shape], 1)


sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
osmpl = OS.OracleSMPL(model)
x, _ = S.proxy_crops(B, seed=1234)
none = {g: None for g in GROUPS}
all_ = lambda t: {g: t for g in GROUPS}
ref, pref = vertices(sd, model, osmpl, x, none, none)
dev = lambda w, a: tuple(float(v) for v in ((lambda r: ((r[0] - ref).abs().max(), (r[1] - pref).abs().max()))(vertices(sd, model, osmpl, x, w, a))))
print(f'{B} crops; max |vertex - fp32| in metres, max |cam / pose6d / betas - fp32|')
print('all bf16 (weights + activations): %.2e m  %.2e' % dev(all_('bf16'), all_('bf16')))
print('all f16                         : %.2e m  %.2e' % dev(all_('f16'), all_('f16')))
print('bf16 weights only               : %.2e m  %.2e' % dev(all_('bf16'), none))
print('bf16 activations only           : %.2e m  %.2e' % dev(none, all_('bf16')))
for g in GROUPS:
    w1, a1 = dict(none), dict(none)
    w1[g] = a1[g] = 'bf16'
    w2, a2 = all_('bf16'), all_('bf16')
    w2[g] = a2[g] = 'f16'
    print('%-7s only this group bf16: %.2e m  %.2e   |   all bf16 but this group f16: %.2e m  %.2e' % ((g,) + dev(w1, a1) + dev(w2, a2)))
w, a = all_('bf16'), all_('bf16')
for g in ('layer3', 'layer4'):
    w[g] = a[g] = 'f16'
print('bf16 up to layer2, f16 layer3 + layer4: %.2e m  %.2e' % dev(w, a))
w, a = all_('bf16'), all_('bf16')
a['layer4'] = 'f16'
print('bf16, layer4 ACTIVATIONS f16          : %.2e m  %.2e' % dev(w, a))
w, a = all_('bf16'), all_('bf16')
w['layer4'] = 'f16'
print('bf16, layer4 WEIGHTS f16              : %.2e m  %.2e' % dev(w, a))
