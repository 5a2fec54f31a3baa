"""A/B inside ONE process: the 3x3 stride-1 convolutions of layer2 (and optionally layer3) on conv_patch.hip (K split over wave
pairs, accumulators exchanged through LDS) against conv_p4.hip (four accumulating waves over the whole K + four producer waves):
per-launch durations, the whole forward, and how far the outputs move (the two kernels add the same products in another order,
so the comparison is a tolerance, not bits).
usage: python tools/ab_p4.py [reps] [B ...]      (env AB_P4_PLANS="0,2,4,6": bit 1 = layer2, bit 2 = layer3)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from soccerplayershapepose_amd import _abi, synthetic as S  # noqa: E402
from soccerplayershapepose_amd import runtime as rt  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batches = [int(v) for v in sys.argv[2:]] or [32, 22, 8]
plans = [int(v) for v in os.environ.get('AB_P4_PLANS', '0,2').split(',')]
dev = torch.device('cuda', 0)
r = rt.get(dev)
nk = r.lib.spsp_profile_num_kernels()
names = [r.lib.spsp_profile_kernel_name(i).decode() for i in range(nk)]

for prec in ([os.environ['AB_P4_ONLY']] if os.environ.get('AB_P4_ONLY') else ['f16', 'bf16']):
    for B in batches:
        reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(S.regressor_state_dict(seed=0)).to(dev)
        x, _ = S.proxy_crops(B, seed=1)
        native = reg.pack(torch.from_numpy(x).to(dev))
        res = {}
        for plan in plans:
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_P4, plan), 'debug')
            out = [t.clone() for t in reg(native)]
            stages = [reg.read_stage(s, B).clone() for s in (3, 4)]
            torch.cuda.synchronize()
            reg.check_status()
            ms, cnt = (C.c_double * nk)(), (C.c_int * nk)()
            r.check(r.lib.spsp_profile_select(r.ctx, -1), 'select')
            r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'enable')
            for _ in range(reps):
                reg(native)
                torch.cuda.synchronize()
            r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'enable')
            r.check(r.lib.spsp_profile_read(r.ctx, nk, ms, cnt), 'read')
            k = {names[i]: 1e3 * ms[i] / reps for i in range(nk) if cnt[i]}
            res[plan] = (out, stages, k)
        base = res[plans[0]]
        for plan in plans:
            o, st, k = res[plan]
            d_out = max(float((a - b).abs().max()) for a, b in zip(o, base[0]))
            d_st = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(st, base[1])]
            print(f'{prec} B={B} plan {plan}: patch<false> {k.get("conv3x3_patch_kernel<false>", 0):.2f} us, conv kernels '
                  f'{sum(v for n, v in k.items() if n.startswith("conv")):.1f} us, all {sum(k.values()):.1f} us; '
                  f'vs plan {plans[0]}: outputs max |d| {d_out:.3g}, layer2 / layer3 maps rel {d_st[0]:.3g} / {d_st[1]:.3g}, finite {all(torch.isfinite(t).all().item() for t in o)}', flush=True)
