"""Run ON THE GPU BOX: host time of one eager lookup call (no synchronisation inside the loop), beside the
kernel's own time.  Tells whether the eager step is bound by the host's launch cadence."""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raft_stereo_amd as R

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
f1 = torch.randn((8, 256, 96, 312), generator=g, device=dev).half()
f2 = torch.randn((8, 256, 96, 312), generator=g, device=dev).half()
coords = torch.zeros((8, 2, 96, 312), device=dev)
coords[:, 0] = torch.rand((8, 96, 312), generator=g, device=dev) * 300
with torch.no_grad():
    blk = R.CorrBlock1DHip(f1, f2, num_levels=4, radius=4)
    for _ in range(50):
        blk(coords)
    torch.cuda.synchronize()
    for n in (32, 2000):
        t0 = time.perf_counter()
        for _ in range(n):
            out = blk(coords)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{n} calls: host {1e6 * (t1 - t0) / n:.2f} us per call, with the final sync {1e6 * (t2 - t0) / n:.2f} us per call")
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        out = blk(coords)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
