"""Stress: the same batch through `depth` regressor / fit / SMPL handles on concurrent streams, every result compared
bit-for-bit with a serial reference.  Reports the first backbone stage that differs."""
import os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import PlayerPipeline, reconstruct_players
dev = torch.device('cuda', 0)
B = int(sys.argv[1]); depth = int(sys.argv[2]); iters = int(sys.argv[3])
sd = S.regressor_state_dict(seed=0); model = S.smpl_model(seed=0)
mk_r = lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(sd)
mk_s = lambda: SMPL(model, max_batch=B)
pipe = PlayerPipeline(mk_r, mk_s, dev, depth=depth)
x, _ = S.proxy_crops(B, seed=5)
native = pipe.slots[0].regressor.pack(torch.from_numpy(x).to(dev))
tgt = (torch.rand(B, 17, 2, generator=torch.Generator().manual_seed(3)) * 300 + 100).to(dev)
ref = reconstruct_players(pipe.slots[0].regressor, pipe.slots[0].smpl, native, tgt, fit='pose_shape', iters=50)
refj = ref.joints.clone()
ref = {k: getattr(ref, k).clone() for k in ('cam', 'rotmats', 'betas', 'vertices')}
rc, rp, rs, rf = [t.clone() for t in pipe.slots[0].regressor(native, return_feats=True)]
stages = [pipe.slots[0].regressor.read_stage(i, B).clone() for i in range(6)]
torch.cuda.synchronize()
bad = 0
for it in range(iters):
    outs = []
    for s in pipe.slots:
        with torch.cuda.stream(s.stream):
            outs.append(s.regressor(native, return_feats=True))
    torch.cuda.synchronize()
    for si, (c, p, sh, f) in enumerate(outs):
        if not (torch.equal(f, rf) and torch.equal(c, rc) and torch.equal(p, rp)):
            bad += 1
            msg = f'iter {it} slot {si}: feats diff {(f-rf).abs().max().item():.3e} n={(f!=rf).sum().item()}'
            for st in range(6):
                got = pipe.slots[si].regressor.read_stage(st, B)
                ne = (got != stages[st])
                if ne.any():
                    idx = ne.nonzero()
                    msg += f' | first bad stage {st}: {ne.sum().item()} elems, imgs {sorted(set(idx[:,0].tolist()))}, ch [{idx[:,1].min().item()},{idx[:,1].max().item()}], y [{idx[:,2].min().item()},{idx[:,2].max().item()}], x [{idx[:,3].min().item()},{idx[:,3].max().item()}], max {(got-stages[st]).abs().max().item():.3e}'
                    break
            print(msg, flush=True)
            if bad > 12: break
    if bad > 12: break
print(f'regressor stress: {bad} mismatching results in {iters} x {depth}')
bad = 0
for it in range(max(1, iters // 4)):
    outs = [pipe.submit(native, tgt, fit='pose_shape', iters=50) for _ in range(depth)]
    pipe.synchronize()
    for si, o in enumerate(outs):
        for k in ref:
            if not torch.equal(getattr(o, k), ref[k]):
                bad += 1
                d = (getattr(o, k) - ref[k]).abs()
                redo = pipe.slots[(si + 1) % depth].smpl(full_pose=o.rotmats, betas=o.betas).vertices
                pl = (d.reshape(B, -1).max(1).values > 0).nonzero().flatten().tolist()
                print(f'e2e iter {it} slot {si}: {k} differs by {d.max().item():.3e}; players {pl}; '
                      f'{(d > 0).sum().item()} elems; serial SMPL redo from its own (R, betas) vs its vertices: '
                      f'{(redo - o.vertices).abs().max().item():.3e}, redo vs ref {(redo - ref["vertices"]).abs().max().item():.3e}; '
                      f'joints diff chain[0:24] {(o.joints[:, :24] - refj[:, :24]).abs().max().item():.3e} picked[24:45] {(o.joints[:, 24:45] - refj[:, 24:45]).abs().max().item():.3e} regressed {(o.joints[:, 45:] - refj[:, 45:]).abs().max().item():.3e}', flush=True)
                if k == 'vertices':
                    from soccerplayershapepose_amd.utils.rigid_transform_utils import rot6d_to_rotmat
                    R0 = rot6d_to_rotmat(rp).view(B, 24, 3, 3)
                    pre = pipe.slots[(si + 1) % depth].smpl(full_pose=R0, betas=rs).vertices
                    wrong = (d > 0)
                    p0 = pl[0]
                    vi = wrong[p0].any(1).nonzero().flatten()
                    print(f'   player {p0}: {vi.numel()} wrong vertices, idx min {vi.min().item()} max {vi.max().item()}, idx%64 set {sorted(set((vi % 64).tolist()))[:70]}, blocks {sorted(set((vi // 64).tolist()))[:40]}')
                    print(f'   wrong-vs-prefit max diff on wrong elems: {(o.vertices - pre)[wrong].abs().max().item():.3e}; ref-vs-prefit there: {(ref["vertices"] - pre)[wrong].abs().max().item():.3e}')
                    v0 = vi[0].item()
                    print('   sample got', o.vertices[p0, v0].tolist(), 'ref', ref['vertices'][p0, v0].tolist(), 'prefit', pre[p0, v0].tolist(), 'other player same vertex (got, p0+1)', o.vertices[p0 + 1, v0].tolist())
                break
    if bad > 3: break
print(f'end-to-end stress: {bad} mismatching results')
