"""lab: lookup output stores always non-temporal (tools/lab_libs/lk_nt) against always agent-scope (lk_sc1) WITH the
kernel that reads the taps next, by batch: fp32 taps + cast to 16 bits, 16-bit taps + a reduction that reads them once.
Run with RAFTCORR_LIB=tools/lab_libs/<lk_nt|lk_sc1>/libraftcorr_hip.so python tools/lab_nt_threshold_consumer.py <tag>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def timed(fn, n=30):
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
    tag = sys.argv[1]
    dev = torch.device("cuda:0")
    for batch in (24, 32, 48, 64):
        cfg = dict(bench.WORKLOADS["kitti"])
        cfg["batch"] = batch
        step = bench.Step(cfg, dev, 1)
        dt = step.f1.dtype
        with torch.no_grad():
            blk = step.cls(step.f1, step.f2, num_levels=bench.LEVELS, radius=bench.RADIUS)
            blk16 = step.cls(step.f1, step.f2, num_levels=bench.LEVELS, radius=bench.RADIUS, out_dtype=dt)
            cl = step.coords_list
            it = iter(range(10 ** 9))
            nxt = lambda: cl[next(it) % len(cl)]
            res = dict(f32=timed(lambda: blk(nxt())), f32_cast=timed(lambda: blk(nxt()).to(dt)),
                       h16=timed(lambda: blk16(nxt())), h16_sum=timed(lambda: blk16(nxt()).sum()))
            o32, o16 = blk(cl[0]), blk16(cl[0])
            res["cast_alone"] = timed(lambda: o32.to(dt))
            res["sum_alone"] = timed(lambda: o16.sum())
        print(f"{tag:7s} pairs {batch:3d} " + " ".join(f"{k} {v:8.2f}" for k, v in res.items()), flush=True)
        del step, blk, blk16, o32, o16
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
