"""Timeline of the multi-stream pipeline from the library's own HIP events: which kernels run when, per slot."""
import os, sys, ctypes as C, collections
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S, runtime as rt
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import PlayerPipeline
dev = torch.device('cuda', 0)
B, depth, steps = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 24
fit_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50          # 0 = no fit
plan = sys.argv[3] if len(sys.argv) > 3 else 'throughput'
kw = dict(fit='pose_shape' if fit_iters else None, iters=fit_iters or 50)
sd = S.regressor_state_dict(seed=0); model = S.smpl_model(seed=0)
pipe = PlayerPipeline(lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd),
                      lambda: SMPL(model, max_batch=B), dev, depth=depth)
r = rt.get(dev)
x, _ = S.proxy_crops(B, seed=1)
native = pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev))
tgt = torch.rand(B, 17, 2, device=dev) * 512
for _ in range(12): pipe.submit(native, tgt, **kw)
torch.cuda.synchronize()
r.check(r.lib.spsp_profile_select(r.ctx, -1), 'sel'); r.check(r.lib.spsp_profile_enable(r.ctx, 1), 'en')
for _ in range(steps): pipe.submit(native, tgt, **kw)
torch.cuda.synchronize()
r.check(r.lib.spsp_profile_enable(r.ctx, 0), 'en')
N = 8192
ids = (C.c_int * N)(); st = (C.c_ulonglong * N)(); a = (C.c_double * N)(); b = (C.c_double * N)(); cnt = C.c_int()
r.check(r.lib.spsp_profile_timeline(r.ctx, N, ids, st, a, b, C.byref(cnt)), 'timeline')
names = [r.lib.spsp_profile_kernel_name(i).decode() for i in range(r.lib.spsp_profile_num_kernels())]
recs = [(st[i], names[ids[i]], a[i] * 1e3, b[i] * 1e3) for i in range(cnt.value)]
streams = sorted(set(s for s, *_ in recs))
t0 = min(x[2] for x in recs); t1 = max(x[3] for x in recs)
print(f'{cnt.value} launches, {steps} steps, wall {t1 - t0:.0f} us -> {(t1 - t0) / steps:.0f} us/step (all launches bracketed: ~15 % slower than the real run)')
# steady-state window: drop the first and last quarter
w0, w1 = t0 + (t1 - t0) * 0.25, t0 + (t1 - t0) * 0.75
busy = collections.defaultdict(float); kern = collections.defaultdict(float); ncalls = collections.defaultdict(int)
for s, n, ta, tb in recs:
    o = max(0.0, min(tb, w1) - max(ta, w0))
    busy[s] += o
    if ta >= w0 and tb <= w1: kern[n] += tb - ta; ncalls[n] += 1
for i, s in enumerate(streams):
    print(f'  stream {i}: busy {100 * busy[s] / (w1 - w0):.0f} % of the window')
# concurrency histogram
ev = sorted([(max(ta, w0), 1) for s, n, ta, tb in recs if tb > w0 and ta < w1] + [(min(tb, w1), -1) for s, n, ta, tb in recs if tb > w0 and ta < w1])
hist = collections.defaultdict(float); cur = 0; last = w0
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print('  kernels in flight:', {k: f'{100 * v / (w1 - w0):.0f}%' for k, v in sorted(hist.items())})
print('  per kernel (steady window): avg us under contention')
for n in sorted(kern, key=lambda k: -kern[k]):
    print(f'    {n:44s} {kern[n] / ncalls[n]:7.1f} us x {ncalls[n]}')
# one slot's sequence for one step
s0 = streams[0]
seq = [x for x in recs if x[0] == s0 and x[2] >= w0][:70]
prev = None
print('  one stream, consecutive launches: (gap before, name, duration)')
for s, n, ta, tb in seq[:64]:
    print(f'    +{(ta - prev) if prev is not None else 0:6.1f}  {n:44s} {tb - ta:7.1f}')
    prev = tb
