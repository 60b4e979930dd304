import os, torch, torch.distributed as dist, numpy as np
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dev=torch.device("cuda",0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
name=["/dev/shm/x", 123]
dist.broadcast_object_list(name, src=0)
dist.barrier()
head=torch.arange(5, dtype=torch.int64).to(dev)
heads=[torch.zeros_like(head)]
dist.all_gather(heads, head)
print("nccl world-1 collectives ok", name, torch.stack(heads).cpu().numpy())
t=torch.tensor([1.5],dtype=torch.float64,device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); print(t.item())
dist.destroy_process_group()
