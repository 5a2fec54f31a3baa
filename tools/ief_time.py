"""The fused IEF alone (csrc/regressor.hip, ief_fused_kernel): duration between the kernel's own events at B = 32 / 22 / 64 with
blocks of 32 players (rounds 2-4: 64 workgroups of 256 threads hand their hidden vectors to each other) and of 16 (round 5: two
independent halves per 32 players, 128 threads each), and whether the outputs are the same bits.
usage: python tools/ief_time.py [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from soccerplayershapepose_amd import _abi, synthetic as S  # noqa: E402
from soccerplayershapepose_amd import runtime as rt  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda', 0)
r = rt.get(dev)
nk = r.lib.spsp_profile_num_kernels()
names = [r.lib.spsp_profile_kernel_name(i).decode() for i in range(nk)]
sd = S.regressor_state_dict(seed=0)
for B in (32, 22, 64):
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd).to(dev)
    x, _ = S.proxy_crops(B, seed=3)
    native = reg.pack(torch.from_numpy(x).to(dev))
    outs = {}
    for poll, blk in ((0, 32), (0, 16), (1, 16), (2, 16), (0, 32), (0, 16), (1, 16), (2, 16)):
        r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_IEF_POLL, poll), 'debug')
        r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_IEF_BLOCK, blk), 'debug')
        out = [t.clone() for t in reg(native)]
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
        i = names.index('ief_fused_kernel')
        j = names.index('conv1_pool_kernel')
        ref = outs.setdefault(32, out)
        same = all(torch.equal(a, b) for a, b in zip(out, ref)) or f'no, max deviation {max(float((a - b).abs().max()) for a, b in zip(out, ref)):.1e}'
        form = ('hand-offs poll the data, 32 lanes per player pair' if poll == 2 else 'hand-offs poll the data (ief_poll_kernel)') if poll else f'arrival counters, blocks of {blk} players'
        print(f'B={B}, IEF {form:50s}: {1e3 * ms[i] / cnt[i]:.2f} us (stem {1e3 * ms[j] / cnt[j]:.2f} us); outputs equal to the 32-player blocks\': {same}', flush=True)
