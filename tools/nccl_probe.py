"""RCCL sanity check on one GPU (the multi-GPU bench itself is run by the driver): process group
init, barrier and the all_gather_into_tensor used by distributed.gather_poses."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
rows = torch.arange(18 * 4, dtype=torch.float32, device="cuda").reshape(4, 18)
out = torch.empty_like(rows)
dist.all_gather_into_tensor(out, rows)
dist.barrier()
t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(out, rows) and float(t) == 1.5
dist.destroy_process_group()
print("nccl ok")
