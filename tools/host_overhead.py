"""Where the host time of one lookup goes (run on the GPU box): times the pieces of CorrBlock1DHip.__call__
on the realtime shape, where the kernel (~3 us) is shorter than the host path."""
import sys
import time

import torch

sys.path.insert(0, ".")
import raft_stereo_amd as R
from raft_stereo_amd import corr as C
from raft_stereo_amd._lib import lib


def bench(fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


dev = torch.device("cuda:0")
f1 = torch.randn(1, 256, 60, 80, device=dev).half()
f2 = torch.randn(1, 256, 60, 80, device=dev).half()
coords = torch.zeros(1, 2, 60, 80, device=dev)
coords[:, 0] = torch.arange(80, device=dev).float() - 3.3
with torch.no_grad():
    blk = R.CorrBlock1DHip(f1, f2)
    out = blk(coords)
    print("full __call__            %.2f us" % bench(lambda: blk(coords)))
    print("_lookup                  %.2f us" % bench(lambda: blk._lookup(coords)))
    print("_check_coords            %.2f us" % bench(lambda: C._check_coords(coords, 1, 60, 80, dev)))
    print("torch.empty              %.2f us" % bench(lambda: torch.empty(blk._out_shape, dtype=torch.float32, device=dev)))
    print("_stream                  %.2f us" % bench(lambda: C._stream(dev)))
    print("raw stream               %.2f us" % bench(lambda: torch._C._cuda_getCurrentRawStream(0)))
    st = C._stream(dev)
    cp, op = coords.data_ptr(), out.data_ptr()
    print("data_ptr x2              %.2f us" % bench(lambda: (coords.data_ptr(), out.data_ptr())))
    print("ctypes call (launch)     %.2f us" % bench(lambda: lib.rc_lookup_fwd(blk._vol_ptr, blk._lay_ref, cp, 2, op, 4, 0, st)))
    with C._on_device(dev):
        pass
    print("_on_device ctx           %.2f us" % bench(lambda: C._on_device(dev).__enter__()))
    print("is_grad_enabled          %.2f us" % bench(torch.is_grad_enabled))

# the autograd path (training): Function.apply per lookup, then the deferred backward
g1 = torch.randn(8, 256, 80, 180, device=dev).bfloat16().requires_grad_(True)
g2 = torch.randn(8, 256, 80, 180, device=dev).bfloat16().requires_grad_(True)
c8 = torch.zeros(8, 2, 80, 180, device=dev)
c8[:, 0] = torch.arange(180, device=dev).float() - 3.3
blk = R.CorrBlock1DHip(g1, g2)
print("autograd __call__        %.2f us (kernel ~8.5 us)" % bench(lambda: blk(c8), n=2000))
blk._pending.clear()
gout = torch.randn(8, 36, 80, 180, device=dev)


def train_step():
    b = R.CorrBlock1DHip(g1, g2)
    outs = [b(c8) for _ in range(22)]
    g1.grad = g2.grad = None
    torch.autograd.backward(outs, [gout] * 22)


for _ in range(5):
    train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    train_step()
t1 = time.perf_counter()           # host time to ENQUEUE 50 steps (no sync in between)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("train step: host enqueue %.1f us, with GPU drain %.1f us" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    train_step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
