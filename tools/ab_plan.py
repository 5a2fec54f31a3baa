"""A/B of regressor plan switches inside ONE process (same box, same clocks): per-kernel durations (the library's own
profiling scopes, one batch in flight) and the whole forward, for every setting of
    stride-2 stages:  ring kernel of round 2 | parity-plane patch kernel (conv_s2.hip)
    layer4 3x3 convs: 64- | 32-channel blocks
    layer3 / layer4:  with | without four DMA-only producer waves per workgroup
usage: python tools/ab_plan.py [B] [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from soccerplayershapepose_amd import _abi, synthetic as S  # noqa: E402
from soccerplayershapepose_amd import runtime as rt  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(S.regressor_state_dict(seed=0)).to(dev)
x, _ = S.proxy_crops(B, seed=1)
native = reg.pack(torch.from_numpy(x).to(dev))
r = rt.get(dev)


def profile():
    nk = r.lib.spsp_profile_num_kernels()
    ms, cnt = (C.c_double * nk)(), (C.c_int * nk)()
    r.check(r.lib.spsp_profile_select(r.ctx, -1), 'select')
    r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'enable')
    for _ in range(reps):
        reg(native)
        torch.cuda.synchronize()
    r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'enable')
    r.check(r.lib.spsp_profile_read(r.ctx, nk, ms, cnt), 'read')
    return {r.lib.spsp_profile_kernel_name(i).decode(): (1e3 * ms[i] / cnt[i], cnt[i] / reps) for i in range(nk) if cnt[i]}


def whole():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        reg(native)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        reg(native)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


ref = None
for s2, nch, prod in ((1, 64, 0), (1, 64, 1), (1, 64, 2), (1, 32, 1)):      # prod 2: also layer2, as 128-pixel tiles
    if True:
        r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_S2, s2), 'debug')
        r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_LAYER4_CHANNELS, nch), 'debug')
        for L in (2, 3):       # layer3, layer4: four DMA-only producer waves in the 128-pixel patch kernels
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_PATCH_PRODUCER, L * 10 + (1 if prod else 0)), 'debug')
        r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_PATCH, 1 * 10000 + (128 if prod == 2 else 256) * 10 + 3 * 2 + (1 if prod == 2 else 0)), 'debug')
        r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_PATCH_PRODUCER, 10 + (1 if prod == 2 else 0)), 'debug')
        out = reg(native)
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        dev_ = max((a - b).abs().max().item() for a, b in zip(out, ref))
        k = profile()
        conv = sum(v[0] * v[1] for n, v in k.items() if n.startswith('conv'))
        print(f's2={s2} layer4_channels={nch} producer_waves={prod}: forward {whole():7.1f} us, conv kernels {conv:6.1f} us/forward, max |out - first setting| {dev_:.2e}')
        print('     ' + ', '.join(f'{n} {v[0]:.2f} x{v[1]:.0f}' for n, v in k.items() if n.startswith('conv')))
