import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
import torch, torch.distributed as dist
from soccerplayershapepose_amd.sharding import RecordExchange
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1)
for mode in ('abi', 'torch'):
    ex = RecordExchange(dev, mode=mode)
    rec = torch.randn(32, 229, device=dev)
    big = torch.randn(8192, 8192, device=dev)
    for trial in range(3):
        torch.cuda.synchronize()
        # queue ~5 ms of GPU work, then call the exchange: a blocking call returns only after the work
        t0 = time.perf_counter()
        for _ in range(4): big @ big
        t1 = time.perf_counter()
        out = ex.all_gather(rec, 32)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f'{mode}: queueing {1e3*(t1-t0):.2f} ms, all_gather call {1e3*(t2-t1):.2f} ms, drain {1e3*(t3-t2):.2f} ms, ok={torch.equal(out, rec)}', flush=True)
    ex.close()
dist.destroy_process_group()
