"""Duration of the 50-iteration pose+shape fit of 32 / 22 players, alone on the GPU: the v2 schedule (default) against the
ten-barrier schedule of rounds 1-3 with the tree walks and with the level sweeps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from soccerplayershapepose_amd import _abi, synthetic as S
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import fit_players
dev = torch.device('cuda', 0)
for plan in ('v2', 'walks', 'level sweeps'):
    for B in (32, 22):
        smpl = SMPL(S.smpl_model(seed=0), max_batch=32).to(dev)
        smpl.debug(_abi.DEBUG_FIT_SCHEDULE, {'v2': 2, 'walks': 1, 'level sweeps': 0}[plan])
        R, b, cam = S.pose_params(B, seed=1)
        tgt = torch.rand(B, 17, 2, device=dev) * 512
        args = lambda: (torch.from_numpy(R).to(dev).contiguous(), torch.from_numpy(b).to(dev).contiguous(), torch.from_numpy(cam).to(dev).contiguous())
        ts = []
        for i in range(12):
            a = args()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fit_players(smpl, *a, tgt, variant=1, iters=50)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f'plan {plan:12s} B={B}: {sorted(ts[2:])[len(ts[2:]) // 2]:.1f} us (incl. launch)')
