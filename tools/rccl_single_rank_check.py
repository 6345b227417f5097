"""One-rank RCCL sanity run on a GPU box (the pool has no multi-GPU node): backend "nccl" initialises, and the two collectives bench.py
uses with world > 1 (gather of the final latents onto rank 0, all_reduce of a counter / of the step time) execute on the device."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
x = torch.randn(4, 30, 256, 32, device=dev)
out = [torch.empty_like(x)]
dist.gather(x, out, dst=0)
ones = torch.ones(1, device=dev, dtype=torch.int32)
dist.all_reduce(ones, op=dist.ReduceOp.SUM)
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
mine = torch.tensor([0.5, 0.0], device=dev, dtype=torch.float64)   # (bench.py: every rank's own step time, gathered by all ranks)
every = [torch.empty_like(mine)]
dist.all_gather(every, mine)
assert torch.equal(every[0], mine)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(out[0], x) and int(ones.item()) == 1 and float(t) == 1.25
print("rccl single-rank: backend", dist.get_backend(), "gather / all_reduce / all_gather / barrier ok")
dist.destroy_process_group()
