"""Generate golden vectors for the correlation hot path from the reference's own Python.

Run in the build container only (the reference tree is not shipped anywhere else):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [/root/reference]

It imports `core.corr` from the reference, runs CorrBlock1D (the `reg` path,
core/corr.py:110-156) and PytorchAlternateCorrBlock1D (core/corr.py:64-107) on CPU in fp32 on
small seeded inputs, and stores inputs + outputs (+ autograd gradients) as compressed .npz
files next to this script.  Only data is stored; no reference source is copied.
"""
import os
import sys

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from core.corr import CorrBlock1D, PytorchAlternateCorrBlock1D  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make_coords(B, H, W1, W2, rng, kind):
    xs = np.arange(W1, dtype=np.float32)[None, None, :].repeat(B, 0).repeat(H, 1)
    ys = np.arange(H, dtype=np.float32)[None, :, None].repeat(B, 0).repeat(W1, 2)
    if kind == "smooth":
        disp = rng.uniform(0, W2 / 3, size=(B, H, 1)).astype(np.float32) + \
            rng.uniform(-0.5, 0.5, size=(B, H, W1)).astype(np.float32)
        x = xs - disp
    elif kind == "random":
        x = rng.uniform(-6, W2 + 6, size=(B, H, W1)).astype(np.float32)
    elif kind == "edge":
        # integer-exact, negative, == W2-1, > W2-1, far out of range, half-integers
        pool = np.array([0.0, 1.0, -1.0, -0.5, -4.0, -5.0, -13.25, W2 - 1.0, W2 - 0.5, W2 + 0.0,
                         W2 + 3.0, W2 + 4.0, W2 + 5.5, 1e4, -1e4, 2.5, 7.0, (W2 - 1) / 2.0,
                         W2 / 2.0 + 0.25, 3.999999, 4.000001], dtype=np.float32)
        x = pool[rng.integers(0, len(pool), size=(B, H, W1))]
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(np.stack([x, ys], axis=1).astype(np.float32))


def run_case(name, B, D, H, W1, W2, L, r, seed, with_alt=True):
    rng = np.random.default_rng(seed)
    f1 = rng.standard_normal((B, D, H, W1)).astype(np.float32)
    f2 = rng.standard_normal((B, D, H, W2)).astype(np.float32)
    kinds = ["smooth", "random", "edge"]
    coords = [make_coords(B, H, W1, W2, rng, k) for k in kinds]
    gouts = [rng.standard_normal((B, L * (2 * r + 1), H, W1)).astype(np.float32) for _ in kinds]

    t1 = torch.from_numpy(f1).requires_grad_(True)
    t2 = torch.from_numpy(f2).requires_grad_(True)
    blk = CorrBlock1D(t1, t2, num_levels=L, radius=r)
    out = {"f1": f1, "f2": f2, "L": np.int64(L), "r": np.int64(r)}
    for i in range(L):
        lvl = blk.corr_pyramid[i].detach().numpy()
        out[f"level{i}"] = lvl.reshape(B, H, W1, lvl.shape[-1])
    looks = []
    for j, (c, k) in enumerate(zip(coords, kinds)):
        o = blk(torch.from_numpy(c))
        looks.append(o)
        out[f"coords_{k}"] = c
        out[f"lookup_{k}"] = o.detach().numpy()
        out[f"gout_{k}"] = gouts[j]
    # single lookup gradient
    g1, g2 = torch.autograd.grad((looks[0] * torch.from_numpy(gouts[0])).sum(), (t1, t2), retain_graph=True)
    out["df1_single"], out["df2_single"] = g1.numpy(), g2.numpy()
    # three accumulated lookups (what RAFT's GRU loop produces, core/raft_stereo.py:108-110)
    tot = sum((o * torch.from_numpy(g)).sum() for o, g in zip(looks, gouts))
    g1, g2 = torch.autograd.grad(tot, (t1, t2))
    out["df1_multi"], out["df2_multi"] = g1.numpy(), g2.numpy()

    if with_alt and H > 1:
        with torch.no_grad():
            alt = PytorchAlternateCorrBlock1D(torch.from_numpy(f1), torch.from_numpy(f2), num_levels=L, radius=r)
            for c, k in zip(coords, kinds):
                out[f"alt_{k}"] = alt(torch.from_numpy(c)).numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 0},
          os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.manual_seed(1234)
    torch.set_num_threads(4)
    run_case("tiny_b2_d32_h3_w24", B=2, D=32, H=3, W1=24, W2=24, L=4, r=4, seed=11)
    run_case("odd_b1_d64_h2_w40x45", B=1, D=64, H=2, W1=40, W2=45, L=4, r=4, seed=12)
    run_case("d256_b1_h4_w128", B=1, D=256, H=4, W1=128, W2=128, L=4, r=4, seed=13)
    run_case("wide_b1_d256_h2_w72x200", B=1, D=256, H=2, W1=72, W2=200, L=4, r=4, seed=14)
    run_case("lv2_r3_b1_d32_h2_w33", B=1, D=32, H=2, W1=33, W2=33, L=2, r=3, seed=15)
