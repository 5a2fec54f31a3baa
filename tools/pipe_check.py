"""Correctness of the four-slot graph pipeline under load: every 7th result of a 200-step run (and the last four) against the serial
eager path on the same inputs, bit for bit.  usage: python tools/pipe_check.py [prod0|prod1] [eager] [soak]   (SPSP_PIPE_TEST_NUDGE=1 SPSP_PIPE_NUDGE_PERIOD=n: fills on the caller's stream)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S, _abi
from soccerplayershapepose_amd import runtime as rt
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import PlayerPipeline, reconstruct_players
prod = 0 if 'prod0' in sys.argv else 1
B, steps = 32, 200
rounds = 12 if 'soak' in sys.argv else 5
dev = torch.device('cuda', 0)
r = rt.get(dev)
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)
def mk_r():
    reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd).to(dev)
    for L in (2, 3):
        r.check(r.lib.spsp_regressor_debug(reg.handle, _abi.DEBUG_PLAN_PATCH_PRODUCER, L * 10 + prod), 'debug')
    return reg
mk_s = lambda: SMPL(model, max_batch=B)
pipe = PlayerPipeline(mk_r, mk_s, dev, depth=4, use_graphs='eager' not in sys.argv)
inputs, refs = [], []
ref_reg, ref_smpl = mk_r(), mk_s().to(dev)
for k in range(4):
    x, _ = S.proxy_crops(B, seed=1234 + 1000 * k)
    nat = ref_reg.pack(torch.from_numpy(x).to(dev))
    tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3 + k)) * 300 + 100).to(dev)
    ref = reconstruct_players(ref_reg, ref_smpl, nat, tgt, fit='pose_shape', iters=50)
    torch.cuda.synchronize()
    inputs.append((nat, tgt))
    refs.append((ref.vertices.clone(), ref.cam.clone(), ref.rotmats.clone()))
for rd in range(rounds):
    last = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        nat, tgt = inputs[i % 4]
        last[i % 4] = pipe.submit(nat, tgt, fit='pose_shape', iters=50)
    torch.cuda.synchronize()
    rate = B * steps / (time.perf_counter() - t0)
    bad, worst = 0, 0.0
    for k, res in last.items():
        rv, rc, rR = refs[k]
        d = max((res.vertices - rv).abs().max().item(), (res.cam - rc).abs().max().item(), (res.rotmats - rR).abs().max().item())
        worst = max(worst, d)
        bad += d != 0.0
    assert torch.isfinite(res.vertices).all()              # (a PyTorch kernel on the caller's stream between the rounds, as in tools/ab_pipeline.py)
    print(f'round {rd}: {rate:.0f} players/s; {bad} of 4 final results differ from the serial path, worst |d| = {worst:.3e}', flush=True)
