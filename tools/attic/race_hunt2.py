"""Which co-running kernel corrupts the SMPL skin kernel?  Stream A: SMPL forwards; streams B..: a chosen aggressor."""
import os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import fit_players
dev = torch.device('cuda', 0)
B = 32
aggr = sys.argv[1]            # regressor | fit | smpl | none
rounds = int(sys.argv[2])
sd = S.regressor_state_dict(seed=0); model = S.smpl_model(seed=0)
regs = [SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(sd).to(dev) for _ in range(3)]
smpls = [SMPL(model, max_batch=B).to(dev) for _ in range(4)]
x, _ = S.proxy_crops(B, seed=5)
native = regs[0].pack(torch.from_numpy(x).to(dev))
R, betas, _ = S.pose_params(B, seed=3, body_deg=20, global_deg=30)
R = torch.from_numpy(R).to(dev).contiguous(); betas = torch.from_numpy(betas).to(dev).contiguous()
cam = torch.tensor([[0.9, 0., 0.]], device=dev).repeat(B, 1).contiguous()
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
ref = smpls[0](full_pose=R, betas=betas).vertices.clone()
torch.cuda.synchronize()
sa = torch.cuda.Stream(); others = [torch.cuda.Stream() for _ in range(3)]
bad = 0; total = 0
for it in range(rounds):
    outs = []
    for rep in range(3):
        for i, st in enumerate(others):
            with torch.cuda.stream(st):
                if aggr == 'regressor': regs[i](native)
                elif aggr == 'fit':
                    fit_players(smpls[1 + i], R.clone(), betas.clone(), cam.clone(), tgt, variant=1, iters=50)
                elif aggr == 'smpl': smpls[1 + i](full_pose=R, betas=betas)
        with torch.cuda.stream(sa):
            for _ in range(4):
                outs.append(smpls[0](full_pose=R, betas=betas).vertices)
    torch.cuda.synchronize()
    for o in outs:
        total += 1
        if not torch.equal(o, ref):
            bad += 1
            if bad == 1:
                np.save('gpurun_out/skin_bad.npy', o.cpu().numpy()); np.save('gpurun_out/skin_ref.npy', ref.cpu().numpy())
                np.save('gpurun_out/skin_R.npy', R.cpu().numpy()); np.save('gpurun_out/skin_betas.npy', betas.cpu().numpy())
            if bad <= 3:
                d = (o - ref).abs(); w = (d > 0)
                pl = w.reshape(B, -1).any(1).nonzero().flatten().tolist()
                vi = w.any(2).nonzero()
                print(f'  mismatch: max {d.max().item():.3e}, players {pl}, lanes {sorted(set((vi[:,1] % 64).tolist()))}', flush=True)
print(f'aggressor={aggr} env={ {k: v for k, v in os.environ.items() if k.startswith("SPSP_")} }: {bad}/{total} SMPL forwards corrupted', flush=True)
