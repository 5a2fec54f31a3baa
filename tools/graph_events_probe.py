"""Do HIP events recorded inside a captured graph give valid elapsed times after replay?"""
import os, sys, ctypes as C
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.getcwd())
import torch
from soccerplayershapepose_amd import synthetic as S, runtime as rt
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import PlayerPipeline
dev = torch.device('cuda', 0)
B = 32
sd = S.regressor_state_dict(seed=0); model = S.smpl_model(seed=0)
pipe = PlayerPipeline(lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(sd),
                      lambda: SMPL(model, max_batch=B), dev, depth=2, use_graphs=True)
r = rt.get(dev)
x, _ = S.proxy_crops(B, seed=1)
native = pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev))
tgt = torch.rand(B, 17, 2, device=dev) * 512
# warm every slot eagerly first (capture happens on the second use)
pipe.use_graphs = False
for _ in range(2): pipe.submit(native, tgt, fit='pose_shape', iters=50)
torch.cuda.synchronize()
for s in pipe.slots: s.warm = True
pipe.use_graphs = True
names = [r.lib.spsp_profile_kernel_name(i).decode() for i in range(r.lib.spsp_profile_num_kernels())]
kid = names.index('conv3x3_patch_kernel<false, 2, 5>')
r.check(r.lib.spsp_profile_select(r.ctx, kid), 'select')
r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'enable')
for _ in range(2): pipe.submit(native, tgt, fit='pose_shape', iters=50)      # capture (events become graph nodes)
torch.cuda.synchronize()
for _ in range(10): pipe.submit(native, tgt, fit='pose_shape', iters=50)     # replays
torch.cuda.synchronize()
r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'enable')
nk = len(names); ms = (C.c_double * nk)(); cnt = (C.c_int * nk)()
rc = r.lib.spsp_profile_read(r.ctx, nk, ms, cnt)
print('profile_read rc', rc, r.lib.spsp_last_error().decode() if rc else '')
print('records', cnt[kid], 'avg us', 1e3 * ms[kid] / max(1, cnt[kid]))
