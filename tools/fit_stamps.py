"""Per-phase cycles of one Adam iteration of the fit kernel (diagnostic twin csrc/libspsp_stamps.so, -DSPSP_FIT_STAMPS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SPSP_LIB_PATH'] = os.path.join(ROOT, 'soccerplayershapepose_amd', 'csrc', 'libspsp_stamps.so')
os.environ['SPSP_FIT_DEBUG'] = '1'
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import fit_players
dev = torch.device('cuda', 0)
B = 32
smpl = SMPL(S.smpl_model(seed=0), max_batch=B).to(dev)
R, b, cam = S.pose_params(B, seed=1)
tgt = torch.rand(B, 17, 2, device=dev) * 512
for _ in range(2):
    fit_players(smpl, torch.from_numpy(R).to(dev).contiguous(), torch.from_numpy(b).to(dev).contiguous(), torch.from_numpy(cam).to(dev).contiguous(), tgt, variant=1, iters=50)
    torch.cuda.synchronize()
