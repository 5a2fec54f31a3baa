"""layer2's three 3x3 / 1 convolutions in the regressor's forward, one batch in flight: conv_p4.hip (weights through an LDS ring per tile)
against conv_rs.hip (round 5: a channel block's weights in registers, persistent) on 32 / 64 / 128 workgroups -- per-launch duration,
workgroups and CU time (duration x share of the 256 CUs).
usage: python tools/layer2_time.py [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from soccerplayershapepose_amd import _abi, synthetic as S  # noqa: E402
from soccerplayershapepose_amd import runtime as rt  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
r = rt.get(dev)
nk = r.lib.spsp_profile_num_kernels()
names = [r.lib.spsp_profile_kernel_name(i).decode() for i in range(nk)]
sd = S.regressor_state_dict(seed=0)
for B in (32, 22):
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16', plan='throughput').load_state_dict(sd).to(dev)
    x, _ = S.proxy_crops(B, seed=3)
    native = reg.pack(torch.from_numpy(x).to(dev))
    for rnd in range(2):
        for wgs in (0, 32, 64, 128, 256):
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_RS, wgs), 'debug')
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_P4, 2 if wgs == 0 else 6), 'debug')
            reg(native)
            torch.cuda.synchronize()
            ms, cnt, wg = (C.c_double * nk)(), (C.c_int * nk)(), (C.c_double * nk)()
            r.check(r.lib.spsp_profile_select(r.ctx, -1), 'select')
            r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'enable')
            for _ in range(reps):
                reg(native)
                torch.cuda.synchronize()
            r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'enable')
            r.check(r.lib.spsp_profile_read_grids(r.ctx, nk, ms, cnt, wg), 'read')
            for k in ('conv3x3rs_kernel', 'conv3x3_p4_kernel', 'conv3x3_patch_kernel<false>'):
                i = names.index(k)
                if cnt[i]:
                    us, w = 1e3 * ms[i] / cnt[i], wg[i] / cnt[i]
                    print(f'B={B} rs_wgs={wgs:3d}: {k:28s} {cnt[i] // reps} x {us:6.2f} us on {w:5.1f} workgroups = {us * min(1.0, w / 256):5.2f} CU-us each', flush=True)
