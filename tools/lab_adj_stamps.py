"""Lab (run ON THE GPU BOX with RAFTCORR_LIB=tools/lab_adj_stamps.so, built by tools/lab_adj.sh with
-DRC_LAB_ADJ_STAMPS): where a wave of the fused adjoint spends its time in bench.py's sceneflow_train step.
Prints, per phase, the mean time since the wave started, in units of the kernel's whole duration and in us
(the scale comes from the kernel's own first and last stamp against the HIP-event time of the backward)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = dict(bench.WORKLOADS["sceneflow_train"])
dev = torch.device("cuda:0")
step = bench.Step(cfg, dev, seed=7)
for _ in range(6):
    step.run(record=False)
torch.cuda.synchronize()
import raft_stereo_amd._lib as L  # noqa: E402

lib = L.lib
NS, NW, NB = 16, 4, 8192
buf = np.zeros(NB * NW * NS, dtype=np.uint64)
fn = lib.rc_lab_adj_read_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.size) == 0
st = buf.reshape(NB, NW, NS).astype(np.int64)
used = st[:, :, 0] != 0
w = st[used]                      # [waves, NS]
t0 = int(np.quantile(w[:, 0], 0.001))
span = int(np.quantile(w[:, 14], 0.999)) - t0
print(f"waves {w.shape[0]}  kernel span {span} ticks (s_memtime)")
names = {0: "start", 1: "rows zeroed", 12: "lookups done", 13: "after barrier", 14: "end"}
for g in range(10):
    names[2 + g] = f"group {g} of lookups"
life = (w[:, 14] - w[:, 0])
print(f"mean wave lifetime {life.mean():.0f} ticks = {life.mean() / span:.3f} of the kernel's span")
for s in range(NS):
    ok = w[:, s] != 0
    if not ok.any() or s not in names:
        continue
    dt = (w[ok, s] - w[ok, 0])
    print(f"  {names[s]:22s} +{dt.mean():9.0f} ticks  ({dt.mean() / life.mean():.3f} of a wave's life)  waves {ok.sum()}")
starts = np.sort(w[:, 0] - t0)
print("wave start times, deciles (fraction of the kernel):", np.round(np.quantile(starts, np.linspace(0, 1, 11)) / span, 3))
