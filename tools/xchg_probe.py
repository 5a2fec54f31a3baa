"""World-size-1 rehearsal of the two record exchanges: per-call cost with nothing else on the GPU, and whether the call
blocks the host.  (At world size 1 RCCL's all-gather is a device copy either way; the point is the per-call overhead.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
import torch, torch.distributed as dist
from soccerplayershapepose_amd.sharding import RecordExchange
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1)
rec = torch.randn(32, 229, device=dev)
for mode in ('abi', 'torch'):
    ex = RecordExchange(dev, mode=mode)
    for _ in range(20): ex.all_gather(rec, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): out = ex.all_gather(rec, 32)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{mode}: host {1e6*(t1-t0)/500:.1f} us per call, device-complete {1e6*(t2-t0)/500:.1f} us per call, ok={torch.equal(out, rec)}', flush=True)
    lat = []
    for _ in range(10):
        time.sleep(0.01)
        torch.cuda.synchronize(); t0 = time.perf_counter(); ex.all_gather(rec, 32); torch.cuda.synchronize(); lat.append(1e6 * (time.perf_counter() - t0))
    print(f'{mode}: ONE call on an idle GPU, issue -> device complete: ' + ' '.join(f'{v:.0f}' for v in lat) + ' us', flush=True)
    big = torch.randn(2048, 2048, device=dev)
    for variant in ('work then gather',):
        lat = []
        for _ in range(10):
            time.sleep(0.01)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            y = big @ big
            torch.cuda.synchronize(); t1 = time.perf_counter()
            y = big @ big; ex.all_gather(rec, 32)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            lat.append((1e6 * (t1 - t0), 1e6 * (t2 - t1)))
        print(f'{mode}: matmul alone / matmul + all_gather (us): ' + ' '.join(f'{a:.0f}/{b:.0f}' for a, b in lat), flush=True)
    ex.close()
dist.destroy_process_group()
