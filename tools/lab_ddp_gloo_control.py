"""Control for the 2-rank gloo rehearsal on ONE GPU: plain torch DDP (no kernel of this library), conv + bf16
autocast, a few steps.  Launch: python -m torch.distributed.run --nproc-per-node 2 tools/lab_ddp_gloo_control.py"""
import faulthandler, os, sys, time
import torch
import torch.distributed as dist
faulthandler.dump_traceback_later(100, exit=True)
dist.init_process_group("gloo")
rank = dist.get_rank()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Conv2d(128, 128, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(128, 128, 3, padding=1),
                        torch.nn.ReLU(), torch.nn.Conv2d(128, 2, 3, padding=1)).to(dev)
big = os.environ.get("LAB_BIG") == "1"
if big:   # ~the update block's parameter volume: several 25 MB buckets
    m = torch.nn.Sequential(*[torch.nn.Conv2d(128 if i else 128, 128, 3, padding=1) for i in range(60)],
                            torch.nn.Conv2d(128, 2, 3, padding=1)).to(dev)
ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], gradient_as_bucket_view=False)
x = torch.randn(8, 128, 80, 180, device=dev)
for it in range(4):
    t0 = time.time()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ddp(x)
    loss = y.float().abs().mean()
    loss.backward()
    torch.cuda.synchronize()
    print(f"rank {rank} step {it} ok {time.time() - t0:.2f}s", flush=True)
dist.barrier()
print(f"rank {rank} done", flush=True)
