"""CPU baseline port of the reference's `reg` correlation path, on the same torch ops.

TEST INFRASTRUCTURE ONLY (see oracle/corr_oracle.py).  This file exists so that
`bench.py`'s `cpu_baseline` leg times the operations the reference itself runs on a CPU
(torch.einsum + avg_pool2d + grid_sample; core/corr.py:110-156, core/utils/utils.py:59-68)
on the GPU box, where the reference tree is absent.  kind = "port".
"""
import math

import torch
import torch.nn.functional as F


class RegCorrCPU:
    """Same call contract as core/corr.py:110-146: RegCorrCPU(f1, f2, num_levels, radius)(coords)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        b, d, h, w1 = fmap1.shape
        w2 = fmap2.shape[-1]
        vol = torch.einsum("bdhi,bdhj->bhij", fmap1, fmap2) / math.sqrt(d)     # corr.py:154-156
        vol = vol.reshape(b * h * w1, 1, 1, w2)
        self.levels = [vol]
        for _ in range(1, num_levels):                                          # corr.py:123-125
            vol = F.avg_pool2d(vol, (1, 2), stride=(1, 2))
            self.levels.append(vol)
        self.radius = radius

    def __call__(self, coords):
        b, _, h, w = coords.shape
        r = self.radius
        x = coords[:, 0].reshape(b * h * w, 1, 1)
        taps = torch.arange(-r, r + 1, dtype=coords.dtype, device=coords.device).view(1, 1, -1)
        feats = []
        for i, lvl in enumerate(self.levels):                                   # corr.py:133-143
            wl = lvl.shape[-1]
            gx = 2.0 * (x / 2 ** i + taps) / (wl - 1) - 1.0                     # utils.py:63
            grid = torch.stack([gx, torch.zeros_like(gx)], dim=-1)              # [N,1,2r+1,2]
            s = F.grid_sample(lvl, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            feats.append(s.view(b, h, w, 2 * r + 1))
        return torch.cat(feats, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def time_reg_path(b, d, h, w, iters, num_levels=4, radius=4, threads=None, seed=1234):
    """Returns (build_seconds, lookup_seconds_total, threads_used) for one pass on CPU."""
    import time
    if threads:
        torch.set_num_threads(int(threads))
    g = torch.Generator().manual_seed(seed)
    f1 = torch.randn(b, d, h, w, generator=g)
    f2 = torch.randn(b, d, h, w, generator=g)
    xs = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w)
    ys = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1).expand(b, 1, h, w)
    disp = torch.rand(b, 1, h, w, generator=g) * (w / 3)
    coords = torch.cat([xs - disp, ys], dim=1).contiguous()
    RegCorrCPU(f1, f2, num_levels, radius)(coords)            # warm-up
    t0 = time.perf_counter()
    blk = RegCorrCPU(f1, f2, num_levels, radius)
    t1 = time.perf_counter()
    for _ in range(iters):
        blk(coords)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, torch.get_num_threads()
