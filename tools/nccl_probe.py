"""world_size-1 probe of the torch.distributed calls bench.py makes on the nccl (RCCL) backend:
all_gather of a 36-element int64 tensor on the GPU, all_reduce MAX of a float64, barrier."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t = torch.arange(36, dtype=torch.int64).to(dev)
parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
dist.all_gather(parts, t)
assert (parts[0].cpu() == torch.arange(36)).all()
x = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
dist.barrier()
print("nccl probe ok", float(x.item()))
dist.destroy_process_group()
