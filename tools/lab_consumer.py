"""lab: what a store cache policy on the lookup's / fused lookup's OUTPUT costs the kernel that reads it next.
Run with RAFTCORR_LIB=tools/lab_libs/<policy>/libraftcorr_hip.so (tools/lab_store_policy_build.sh).
Prints producer-alone and producer+consumer times (HIP events, current stream) at 8 and 64 KITTI pairs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def timed(fn, n=40):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "?"
    dev = torch.device("cuda:0")
    for batch in (8, 64):
        cfg = dict(bench.WORKLOADS["kitti"])
        cfg["batch"] = batch
        step = bench.Step(cfg, dev, 1)
        dt = step.f1.dtype
        g = torch.Generator(device=dev).manual_seed(5)
        w1 = (torch.randn((64, 36, 1, 1), generator=g, device=dev) * 0.2)
        b1 = torch.randn((64,), generator=g, device=dev)
        w1q, b1q = w1.to(dt), b1.to(dt)
        w3 = (torch.randn((64, 64, 3, 3), generator=g, device=dev) * 0.05).to(dt)
        b3 = torch.randn((64,), generator=g, device=dev).to(dt)
        with torch.no_grad():
            blk = step.cls(step.f1, step.f2, num_levels=bench.LEVELS, radius=bench.RADIUS)
            c = step.coords
            res = dict(
                lookup=timed(lambda: blk(c)),
                lookup_cast=timed(lambda: blk(c).to(dt)),
                lookup_convc1=timed(lambda: torch.nn.functional.conv2d(blk(c).to(dt), w1q, b1q)),
                fused=timed(lambda: blk.lookup_convc1(c, w1, b1)),
                fused_sum=timed(lambda: blk.lookup_convc1(c, w1, b1).sum()),
                fused_convc2=timed(lambda: torch.nn.functional.conv2d(blk.lookup_convc1(c, w1, b1), w3, b3, padding=1)),
            )
            out = blk(c)
            fo = blk.lookup_convc1(c, w1, b1)
            res["cast_alone"] = timed(lambda: out.to(dt))
            res["sum_alone"] = timed(lambda: fo.sum())
            res["convc2_alone"] = timed(lambda: torch.nn.functional.conv2d(fo, w3, b3, padding=1))
        print(f"policy {tag:8s} pairs {batch:3d} " + " ".join(f"{k} {v:8.2f}" for k, v in res.items()), flush=True)
        del step, blk, out, fo
        torch.cuda.empty_cache()


main()
