import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import fit_players
dev=torch.device('cuda',0)
smpl=SMPL(S.smpl_model(0),max_batch=32).to(dev)
R,b,cam=S.pose_params(32,seed=1)
out=smpl(full_pose=torch.from_numpy(R).to(dev),betas=torch.from_numpy(b).to(dev))
tgt=torch.from_numpy(S.fit_targets_from_joints(out.joints.cpu().numpy(), cam*np.array([1.08,1,1],np.float32),seed=2)[:,:,:2]).float().to(dev)
for _ in range(2):
    Rg,bg,cg=(torch.from_numpy(a).to(dev).contiguous() for a in (R,b,cam))
    fit_players(smpl,Rg,bg,cg,tgt,variant=1,iters=50)
torch.cuda.synchronize()
