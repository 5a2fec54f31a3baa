"""A/B of regressor plans in the PIPELINED throughput mode (4 slots, HIP graphs) inside one process: players/s of the whole
path (regressor -> fit -> SMPL) for each plan, measured in interleaved rounds so that box / clock differences cancel.
usage: python tools/ab_pipeline.py [steps] [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402

from soccerplayershapepose_amd import _abi, synthetic as S  # noqa: E402
from soccerplayershapepose_amd import runtime as rt  # noqa: E402
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor  # noqa: E402
from soccerplayershapepose_amd.models.smpl_official import SMPL  # noqa: E402
from soccerplayershapepose_amd.player_recon import PlayerPipeline  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = 32
dev = torch.device('cuda', 0)
r = rt.get(dev)
sd, model = S.regressor_state_dict(seed=0), S.smpl_model(seed=0)


def patch(L, tile, ring, db):
    return (_abi.DEBUG_PLAN_PATCH, L * 10000 + tile * 10 + ring * 2 + db)


prod = lambda *layers: [(_abi.DEBUG_PLAN_PATCH_PRODUCER, L * 10 + 1) for L in layers]
W, SB, P4, P4C, S2C = _abi.DEBUG_PLAN_WRES_CUS, _abi.DEBUG_PLAN_STEM_BAND, _abi.DEBUG_PLAN_P4, _abi.DEBUG_PLAN_P4_CUS, _abi.DEBUG_PLAN_S2RS_CUS
base = [(_abi.DEBUG_PLAN_S2, 1), (_abi.DEBUG_PLAN_LAYER4_CHANNELS, 64)]
PLANS = {      # (PlayerPipeline puts its slots in plan "throughput": the hooks below override single choices of it)
    'plan "throughput" (layer1 on 64 workgroups)': base + [(P4, 6), (W, 4), (SB, 64), (S2C, 2)],
    'layer1 on 48 workgroups': base + [(P4, 6), (W, 48), (SB, 64), (S2C, 2)],
    'layer1 on 80 workgroups': base + [(P4, 6), (W, 80), (SB, 64), (S2C, 2)],
    'layer1 on 96 workgroups': base + [(P4, 6), (W, 96), (SB, 64), (S2C, 2)],
    'chip-filling grids, layer2 / layer3 on conv_patch (round 3)': base + [(P4, 0), (W, 1), (SB, 0), (S2C, 1)],
}
if os.environ.get('AB_OLD_PLANS'):
    PLANS = {
        's2 + layer4 64ch': [(_abi.DEBUG_PLAN_S2, 1), (_abi.DEBUG_PLAN_LAYER4_CHANNELS, 64)],
        's2 + layer4 64ch + producer waves in layer3 / layer4': [(_abi.DEBUG_PLAN_S2, 1), (_abi.DEBUG_PLAN_LAYER4_CHANNELS, 64)] + prod(2, 3),
        's2 + layer4 32ch + producer waves in layer3 / layer4': [(_abi.DEBUG_PLAN_S2, 1), (_abi.DEBUG_PLAN_LAYER4_CHANNELS, 32)] + prod(2, 3),
        's2 + 64ch + producers in layer3 / layer4 + layer2 as 128 px tiles with producers': [(_abi.DEBUG_PLAN_S2, 1), (_abi.DEBUG_PLAN_LAYER4_CHANNELS, 64), patch(1, 128, 3, 1)] + prod(1, 2, 3),
    }
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
x, _ = S.proxy_crops(B, seed=1234)
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
pipes = {}
for name, hooks in PLANS.items():
    def mk_r(hooks=hooks):
        reg = SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B, precision='f16').load_state_dict(sd).to(dev)
        for what, val in hooks:
            r.check(r.lib.spsp_regressor_debug(reg.handle, what, val), 'spsp_regressor_debug')
        return reg
    pipe = PlayerPipeline(mk_r, lambda: SMPL(model, max_batch=B), dev, depth=4, use_graphs=True, streams=streams)
    natives = [pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev)) for _ in range(4)]
    for i in range(8):
        pipe.submit(natives[i % 4], tgt, fit='pose_shape', iters=50)
    torch.cuda.synchronize()
    pipes[name] = (pipe, natives)
res = {k: [] for k in PLANS}
for rd in range(rounds):
    for name, (pipe, natives) in pipes.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            out = pipe.submit(natives[i % 4], tgt, fit='pose_shape', iters=50)
        torch.cuda.synchronize()
        res[name].append(B * steps / (time.perf_counter() - t0))
        assert torch.isfinite(out.vertices).all()
for name, v in res.items():
    print(f'{sorted(v)[len(v) // 2]:9.0f} players/s (median of {[round(a) for a in v]})  {name}')
