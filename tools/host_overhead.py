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
