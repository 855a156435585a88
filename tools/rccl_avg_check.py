import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(8, dtype=torch.float32, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.AVG)
torch.cuda.synchronize()
print("AVG ok", x.tolist())
dist.destroy_process_group()
