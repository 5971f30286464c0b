"""RCCL smoke check on a one-GPU box: backend "nccl" with world size 1, one all_reduce of the flat gradient (59 x 200k floats).
The N > 1 exchange itself can only be rehearsed with gloo here (bench.py --backend gloo); the driver runs the real one."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(59 * 200000, device="cuda")
for _ in range(3): dist.all_reduce(x)
torch.cuda.synchronize(); t = time.time()
for _ in range(20): dist.all_reduce(x)
torch.cuda.synchronize(); print("nccl world-1 all_reduce of %.1f MB: %.3f ms" % (x.numel() * 4 / 1e6, (time.time() - t) / 20 * 1e3), float(x[0]))
dist.barrier(); dist.destroy_process_group()
