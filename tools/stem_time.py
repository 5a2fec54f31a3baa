"""The stem kernel alone in the regressor's forward (one batch in flight): the 32x32x16 form (round 5) against the 16x16x32 form, both plans.
usage: python tools/stem_time.py [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from soccerplayershapepose_amd import _abi, synthetic as S  # noqa: E402
from soccerplayershapepose_amd import runtime as rt  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402

FORMS = ('16x16x32', '32x32x16, pooling in the K loop', '32x32x16, pooling behind the loop')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
r = rt.get(dev)
nk = r.lib.spsp_profile_num_kernels()
names = [r.lib.spsp_profile_kernel_name(i).decode() for i in range(nk)]
sd = S.regressor_state_dict(seed=0)
for B in (32, 22):
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd).to(dev)
    x, _ = S.proxy_crops(B, seed=3)
    native = reg.pack(torch.from_numpy(x).to(dev))
    for rnd in range(2):
        for plan in ('latency', 'throughput'):
            for m32 in (0, 2, 1):
                reg.set_plan(plan)
                r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_STEM32, m32), 'debug')
                reg(native)
                torch.cuda.synchronize()
                ms, cnt = (C.c_double * nk)(), (C.c_int * nk)()
                r.check(r.lib.spsp_profile_select(r.ctx, -1), 'select')
                r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'enable')
                for _ in range(reps):
                    reg(native)
                    torch.cuda.synchronize()
                r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'enable')
                r.check(r.lib.spsp_profile_read(r.ctx, nk, ms, cnt), 'read')
                i = names.index('conv1_pool_kernel')
                tot = sum(ms[j] for j in range(nk) if cnt[j] and names[j].startswith('conv')) * 1e3 / reps
                print(f'B={B} plan {plan:10s} stem on {FORMS[m32]:34s}: {1e3 * ms[i] / cnt[i]:6.2f} us; all convolutions {tot:6.1f} us', flush=True)
