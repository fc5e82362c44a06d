"""Single-rank RCCL smoke (GPU box): the process-group calls bench.py makes at N > 1 (init with device_id, barrier,
all_reduce MAX, gather to rank 0) on world_size 1 - checks that RCCL loads and the calls are valid on this stack."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
x = torch.randn(2, 64, 128, 4, device=dev)
buf = [torch.empty_like(x)]
dist.gather(x, gather_list=buf, dst=0)
torch.cuda.synchronize()
assert float(t.item()) == 1.5 and torch.equal(buf[0], x)
dist.destroy_process_group()
print("rccl single-rank smoke: ok")
