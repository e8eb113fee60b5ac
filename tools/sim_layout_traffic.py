#!/usr/bin/env python3
"""Line-level traffic model of candidate pyramid layouts (CPU only; quoted in DESIGN.md section 10).

For the bench's own lookup coordinates it counts, per image row and level, the distinct 128-byte lines a
lookup touches when a line holds P pixels x E consecutive entries (P * E = 64 two-byte elements), for
  cur   the group-skewed layout of include/raftcorr.h   (block row = w2 / E - (w1 >> i) / E)
  disp  a disparity-major layout                         (block row = (w2 - (w1 >> i)) / E)
with one window per level (`separate`) or one run per level pair, levels (0,1) and (2,3) (`merged`: what the
library does since round 2).  Output: bytes per pixel and lookup.

    python tools/sim_layout_traffic.py [exact|gridsample]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_coords  # noqa: E402

B, H, W, R = 2, 96, 312, 4


def count(x, level, start, nw, layout, P, E):
    Wi = W >> level
    q = np.arange(W) >> level
    cnt = 0
    for b in range(x.shape[0]):
        for h in range(H):
            keys = []
            for m in range(nw):
                w2 = start[b, h] + m
                ok = (w2 >= 0) & (w2 < Wi)
                row = (w2 // E - q // E) if layout == "cur" else (w2 - q + 4096) // E
                keys.append((row * 1000 + np.arange(W) // P)[ok])
            cnt += len(np.unique(np.concatenate(keys)))
    return cnt * 128 / (x.shape[0] * H * W)


def run(c, name, ext):
    nw = 2 * R + 2 + 2 * ext
    x = c[:, 0]
    s = [np.floor(x / 2 ** i).astype(int) - R - ext for i in range(4)]
    for lay, P, E in [("cur", 8, 8), ("disp", 8, 8), ("disp", 16, 4), ("disp", 32, 2)]:
        sep = [count(x, i, s[i], nw, lay, P, E) for i in range(4)]
        m01 = count(x, 0, 2 * s[1], 2 * nw, lay, P, E)
        m23 = count(x, 2, 2 * s[3], 2 * nw, lay, P, E)
        print(f"{name:6s} {lay:4s} {P:2d} px x {E} entries per line: separate {[round(v, 1) for v in sep]} = "
              f"{sum(sep):6.1f} B/px   merged (0,1) {m01:5.1f} + (2,3) {m23:5.1f} = {m01 + m23:6.1f} B/px")


def kernel_lines(x, predicated):
    """The lookup kernel's own addressing (csrc/window.h: locate_window, load_blocks), exact mode, 16-bit pyramid:
    distinct 128-byte lines per pixel and lookup for the runs of level pairs (0,1) and (2,3).
    predicated=False: round 2's loads -- every lane loads 4 blocks of its run, lanes whose run ends early re-read
    their last block, lanes whose run misses the image load blocks 0.. of their column, groups left or right of the
    image wrap around to valid rows; predicated=True: round 3's -- blocks without an image entry are not fetched
    (buffer loads, offset out of range)."""
    Bx, Hx, Wx = x.shape
    w1 = np.arange(Wx)
    out = []
    for i in (0, 2):
        Wi = Wx >> i
        nb = (Wi + 7) // 8 * 8 // 8
        s_hi = np.floor(x * 0.5 / 2 ** i).astype(int) - R
        wb_raw = 2 * s_hi
        any_ok = (wb_raw + 4 * R + 3 >= 0) & (wb_raw < Wi)
        wb = np.where(any_ok, wb_raw, 0)
        g0, o = wb >> 3, wb & 7
        nbk = (o + 4 * R + 3) >> 3
        tot = 0
        for b in range(Bx):
            keys = []
            for k in range(4):
                g = g0[b] + (k if predicated else np.minimum(k, nbk[b]))
                sb = (g - (w1 >> i) // 8 + nb - 4) % nb
                key = (np.arange(Hx)[:, None] * 64 + sb) * 64 + w1 // 8
                if predicated:
                    keys.append(key[any_ok[b] & (k <= nbk[b]) & (g >= 0) & (g * 8 < Wi)])
                else:
                    keys.append(key.ravel())
            tot += len(np.unique(np.concatenate(keys)))
        out.append(tot * 128 / (Bx * Hx * Wx))
    return out


if __name__ == "__main__":
    ext = 1 if len(sys.argv) > 1 and sys.argv[1] == "gridsample" else 0
    c = synthetic_coords(B, H, W, seed=1234)
    run(c, "bench", ext)
    if not ext:
        rng = np.random.default_rng(0)
        moved = c[:, 0] + rng.uniform(-1, 1, size=c[:, 0].shape).astype(np.float32)     # bench.py moving_coords
        for pred in (False, True):
            a, b = kernel_lines(moved, pred)
            print(f"kernel addressing, {'round 3 (blocks without image entries not fetched)' if pred else 'round 2 (every lane loads 4 blocks)'}: "
                  f"(0,1) {a:5.1f} + (2,3) {b:5.1f} = {a + b:6.1f} B/px  (+ 4 B coordinate)")
        print(f"pixels whose level-0 run misses the image: {((2 * (np.floor(moved / 2).astype(int) - R) + 4 * R + 3) < 0).mean():.3f}")
    c0 = c.copy()
    c0[:, 0] = np.arange(W)[None, None, :] - 20.3
    run(c0, "const", ext)
