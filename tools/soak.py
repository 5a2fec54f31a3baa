"""Soak: N steps of the four-slot pipeline (plan "throughput", HIP graphs, eight rotating input batches); EVERY result is compared
bit for bit with the serial path's result for its input (checked three submits later, before its slot's buffers are reused).
usage: python tools/soak.py [steps] [players per batch]"""
import os
import sys
import time
from collections import deque

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from soccerplayershapepose_amd import synthetic as S  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402
from soccerplayershapepose_amd.models.smpl_official import SMPL  # noqa: E402
from soccerplayershapepose_amd.player_recon import PlayerPipeline, reconstruct_players  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B, NIN, depth = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 8, 4
dev = torch.device('cuda', 0)
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd).to(dev)
mk_s = lambda: SMPL(model, max_batch=B)
pipe = PlayerPipeline(mk_r, mk_s, dev, depth=depth, use_graphs=True)
natives, tgts = [], []
for k in range(NIN):
    x, _ = S.proxy_crops(B, seed=300 + k)
    natives.append(pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev)))
    tgts.append((torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(k)) * 300 + 100).to(dev))
reg, smpl = mk_r().set_plan('throughput'), mk_s()
fields = ('cam', 'rotmats', 'betas', 'vertices', 'joints', 'translation')
ref = []
for k in range(NIN):
    r_ = reconstruct_players(reg, smpl, natives[k], tgts[k], fit='pose_shape', iters=50)
    ref.append({f: getattr(r_, f).clone() for f in fields})
torch.cuda.synchronize()
pending = deque()
bad = checked = 0
t0 = time.perf_counter()


def check(item):
    global bad, checked
    res, k, i = item
    res.done.synchronize()
    for f in fields:
        if not torch.equal(getattr(res, f), ref[k][f]):
            bad += 1
            if bad <= 5:
                print(f'MISMATCH step {i} input {k} {f}: max |d| {float((getattr(res, f) - ref[k][f]).abs().max()):.3e}', flush=True)
    checked += 1


for i in range(steps):
    k = i % NIN
    pending.append((pipe.submit(natives[k], tgts[k], fit='pose_shape', iters=50), k, i))
    if len(pending) >= depth:             # the oldest result's slot is the one the NEXT submit reuses
        check(pending.popleft())
    if (i + 1) % 1000 == 0:
        print(f'{i + 1} steps, {checked} results checked, {bad} mismatches, {B * (i + 1) / (time.perf_counter() - t0):.0f} players/s incl. the checks', flush=True)
while pending:
    check(pending.popleft())
for sl in pipe.slots:
    sl.regressor.check_status()
print(f'soak: {steps} steps, {checked} results checked bit for bit, {bad} mismatches')
sys.exit(1 if bad else 0)
