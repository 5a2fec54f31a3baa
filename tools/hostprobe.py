import os, time, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from soccerplayershapepose_amd import synthetic as S
from soccerplayershapepose_amd.models.regressor import SingleInputRegressor
from soccerplayershapepose_amd.models.smpl_official import SMPL
from soccerplayershapepose_amd.player_recon import PlayerPipeline
dev = torch.device('cuda', 0)
B = int(sys.argv[1]); depth = int(sys.argv[2])
sd = S.regressor_state_dict(seed=0); model = S.smpl_model(seed=0)
pipe = PlayerPipeline(lambda: SingleInputRegressor(18, 18, 3, mean_params=S.mean_params(), max_batch=B).load_state_dict(sd),
                      lambda: SMPL(model, max_batch=B), dev, depth=depth)
reg = pipe.slots[0].regressor
x, _ = S.proxy_crops(B, seed=1234)
native = reg.pack(torch.from_numpy(x).to(dev))
tgt = torch.rand(B, 17, 2, device=dev) * 512
torch.cuda.synchronize()
for _ in range(8): pipe.submit(native, tgt, fit='pose_shape', iters=50)
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for _ in range(K): pipe.submit(native, tgt, fit='pose_shape', iters=50)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'B={B} depth={depth}: host enqueue {1e6*(t1-t0)/K:.0f} us/step, total {1e6*(t2-t0)/K:.0f} us/step -> {B*K/(t2-t0):.0f} players/s')

# ---- same steps replayed from a HIP graph per slot ----
if len(sys.argv) > 3 and sys.argv[3] == 'graph':
    from soccerplayershapepose_amd.player_recon import reconstruct_players
    graphs = []
    for slot in pipe.slots:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=slot.stream):
            res = reconstruct_players(slot.regressor, slot.smpl, native, tgt, fit='pose_shape', iters=50)
        graphs.append((g, res, slot.stream))
    torch.cuda.synchronize()
    def replay(i):
        g, res, st = graphs[i % depth]
        with torch.cuda.stream(st):
            g.replay()
    for i in range(8): replay(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): replay(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'GRAPH B={B} depth={depth}: host enqueue {1e6*(t1-t0)/K:.0f} us/step, total {1e6*(t2-t0)/K:.0f} us/step -> {B*K/(t2-t0):.0f} players/s')
    ref = pipe.submit(native, tgt, fit='pose_shape', iters=50)
    torch.cuda.synchronize()
    print('graph vs eager max |dv|:', (ref.vertices - graphs[0][1].vertices).abs().max().item())
