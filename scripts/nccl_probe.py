import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
n_local = torch.tensor([5], dtype=torch.int64, device=dev)
sizes = [torch.zeros_like(n_local)]
dist.all_gather(sizes, n_local)
x = torch.arange(5, dtype=torch.int32, device=dev)
out = [torch.empty_like(x)]
dist.all_gather(out, x)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("nccl ok", sizes[0].item(), out[0].tolist(), float(t))
dist.destroy_process_group()
