"""A/B inside ONE process: layer2.0's stride-2 pair on conv_s2.hip (weights streamed through an LDS ring, one tile per workgroup)
against conv_s2rs.hip (weights stationary in registers, two tiles per workgroup): per-launch durations of the stride-2 kernels,
the whole forward, and a bitwise comparison of the regressor outputs and of layer2's stage map.
usage: python tools/ab_s2rs.py [reps] [B ...]"""
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
batches = [int(v) for v in sys.argv[2:]] or [32, 22, 8, 3]
dev = torch.device('cuda', 0)
r = rt.get(dev)
nk = r.lib.spsp_profile_num_kernels()
names = [r.lib.spsp_profile_kernel_name(i).decode() for i in range(nk)]

for prec in ('f16', 'bf16'):
    for B in batches:
        reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision=prec).load_state_dict(S.regressor_state_dict(seed=0)).to(dev)
        x, _ = S.proxy_crops(B, seed=1)
        native = reg.pack(torch.from_numpy(x).to(dev))
        res = {}
        for rs in (0, 1):
            r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_S2RS, rs), 'debug')
            out = [t.clone() for t in reg(native)]
            stage = reg.read_stage(3, B).clone() if hasattr(reg, 'read_stage') else None
            torch.cuda.synchronize()
            ms, cnt = (C.c_double * nk)(), (C.c_int * nk)()
            r.check(r.lib.spsp_profile_select(r.ctx, -1), 'select')
            r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'enable')
            for _ in range(reps):
                reg(native)
                torch.cuda.synchronize()
            r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'enable')
            r.check(r.lib.spsp_profile_read(r.ctx, nk, ms, cnt), 'read')
            k = {names[i]: 1e3 * ms[i] / reps for i in range(nk) if cnt[i]}
            res[rs] = (out, stage, k)
        same = all(torch.equal(a, b) for a, b in zip(res[0][0], res[1][0]))
        same_stage = None if res[0][1] is None else torch.equal(res[0][1], res[1][1])
        print(f'{prec} B={B}: stride-2 kernels per forward {res[0][2]["conv3x3s2_kernel"]:.2f} us (conv_s2) -> {res[1][2]["conv3x3s2_kernel"]:.2f} us (layer2.0 on conv_s2rs); '
              f'conv kernels {sum(v for n, v in res[0][2].items() if n.startswith("conv")):.1f} -> {sum(v for n, v in res[1][2].items() if n.startswith("conv")):.1f} us; '
              f'outputs bit-identical: {same}, layer2 stage map: {same_stage}', flush=True)
