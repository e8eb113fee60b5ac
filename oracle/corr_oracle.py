"""CPU oracle for the RAFT-Stereo 1-D correlation hot path (numpy, float32).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may.

Every function restates one piece of the reference and cites it (paths are relative to
the upstream tree, which is NOT shipped):

  corr_volume        core/corr.py:148-156   (einsum 'aijk,aijh->ajkh' then / sqrt(D))
  pool_pyramid       core/corr.py:119-125   (avg_pool2d([1,2]) chain, floor widths)
  lookup(EXACT)      sampler/sampler_kernel.cu:19-60   (floor + linear interpolation)
  lookup(GRIDSAMPLE) core/corr.py:127-146 + core/utils/utils.py:59-68
                     (the reg path: pixel -> normalised -> pixel round trip of grid_sample)
  lookup_bwd         sampler/sampler_kernel.cu:63-105 / autograd of the reg path
  pool_pyramid_bwd   adjoint of pool_pyramid
  corr_volume_bwd    adjoint of corr_volume
  alt_lookup         core/corr.py:64-107 (on-the-fly variant; equal to reg by linearity)
  convc1_relu(_bwd)  core/update.py:71,78 (first layer of the motion encoder on a lookup result)
  sampler_fwd/bwd_rounded  sampler/sampler_kernel.cu:39-58, :83-103 with a 16-bit scalar_t (every product
                     and sum rounded to the storage type)

Parity pin: the functions are checked in tests/test_oracle_golden.py against vectors
captured from the reference's own Python (tests/golden/make_golden.py ran
`core.corr.CorrBlock1D` / `PytorchAlternateCorrBlock1D` on CPU in the build container).
The reference has no tests or fixtures of its own for this path.
"""
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- build
def corr_volume(f1, f2):
    """C[b,h,w1,w2] = sum_d f1[b,d,h,w1] * f2[b,d,h,w2] / sqrt(D)   (core/corr.py:148-156)."""
    f1 = np.asarray(f1, dtype=F32)
    f2 = np.asarray(f2, dtype=F32)
    D = f1.shape[1]
    c = np.einsum("bdhi,bdhj->bhij", f1, f2, optimize=True).astype(F32)
    return (c / F32(np.sqrt(F32(D)))).astype(F32)


def level_widths(w2, num_levels):
    """Widths of the levels that are ever read: W2 >> i (core/corr.py:123-125, :133-134)."""
    return [w2 >> i for i in range(num_levels)]


def pool_pyramid(vol, num_levels, round_to=None):
    """Levels 0..L-1; level i+1[j] = (level i[2j] + level i[2j+1]) / 2, odd tail dropped.

    core/corr.py:119-125 keeps one more level that is never read (core/corr.py:133-134).
    `round_to` (np.float16 or a callable) re-rounds every level to the storage type, which is
    what the mixed-precision CorrBlockFast1D does (core/corr.py:37-42: pooled levels are means
    of already-rounded values).
    """
    def rnd(x):
        if round_to is None:
            return x.astype(F32)
        if callable(round_to) and not isinstance(round_to, type):
            return round_to(x).astype(F32)
        return x.astype(round_to).astype(F32)

    cur = rnd(np.asarray(vol, dtype=F32))
    out = [cur]
    for _ in range(1, num_levels):
        w = cur.shape[-1] // 2
        if w < 1:
            raise ValueError("W2 too small for the requested number of levels")
        cur = rnd((cur[..., 0:2 * w:2] + cur[..., 1:2 * w:2]) * F32(0.5))
        out.append(cur)
    return out


# ----------------------------------------------------------------------------- lookup
EXACT = "exact"            # sampler_kernel.cu:39-58 coordinates
GRIDSAMPLE = "gridsample"  # core/corr.py:135-141 + core/utils/utils.py:63 + grid_sample coordinates


def tap_coords(x, level, width, radius, mode):
    """For every pixel and tap k in 0..2r: (i0, w0, w1) such that the tap's value is
    w0 * V[i0] + w1 * V[i0 + 1] (zero outside [0, width)).

    EXACT      : x_i = x / 2**level, f = floor(x_i), a = x_i - f; i0 = f - r + k, w0 = 1-a, w1 = a
                 (sampler_kernel.cu:42-57).
    GRIDSAMPLE : p = x_i + (k - r) (core/corr.py:137), g = 2p/(W-1) - 1 (utils.py:63),
                 ix = ((g+1)/2)*(W-1) (grid_sample, align_corners=True), i0 = floor(ix),
                 w1 = ix - i0, w0 = (i0+1) - ix; every step rounds in fp32.
    """
    x = np.asarray(x, dtype=F32)
    xi = (x / F32(2 ** level)).astype(F32)
    ks = np.arange(2 * radius + 1)
    if mode == EXACT:
        xf = np.floor(xi)
        a = (xi - xf).astype(F32)
        f = np.where(np.abs(xf) > 1e9, -10 ** 9, xf).astype(np.int64)
        i0 = f[..., None] - radius + ks
        w1 = np.broadcast_to(a[..., None], i0.shape).astype(F32)
        w0 = (F32(1) - w1).astype(F32)
        return i0, w0, w1
    if mode != GRIDSAMPLE:
        raise ValueError(mode)
    p = (xi[..., None] + (ks - radius).astype(F32)).astype(F32)
    g = ((F32(2) * p) / F32(width - 1) - F32(1)).astype(F32)
    ix = (((g + F32(1)) / F32(2)) * F32(width - 1)).astype(F32)
    x0 = np.floor(ix)
    w1 = (ix - x0).astype(F32)
    w0 = ((x0 + F32(1)) - ix).astype(F32)
    bad = ~(np.abs(x0) < 1e9)          # also catches nan (W == 1)
    i0 = np.where(bad, -10 ** 9, x0).astype(np.int64)
    return i0, np.where(bad, 0, w0).astype(F32), np.where(bad, 0, w1).astype(F32)


def _fetch(level, idx):
    W = level.shape[-1]
    ok = (idx >= 0) & (idx < W)
    v = np.take_along_axis(level, np.clip(idx, 0, W - 1), axis=-1)
    return np.where(ok, v, F32(0)).astype(F32)


def lookup(pyr, coords, radius, mode=EXACT):
    """out[b, i*(2r+1)+k, h, w] = w0 * V_i[i0] + w1 * V_i[i0+1] with (i0,w0,w1) = tap_coords."""
    x = np.asarray(coords, dtype=F32)[:, 0]
    outs = []
    for i, lvl in enumerate(pyr):
        lvl = np.asarray(lvl, dtype=F32)
        i0, w0, w1 = tap_coords(x, i, lvl.shape[-1], radius, mode)
        outs.append((_fetch(lvl, i0) * w0 + _fetch(lvl, i0 + 1) * w1).astype(F32))
    out = np.concatenate(outs, axis=-1)              # [B,H,W1,L*(2r+1)]
    return np.ascontiguousarray(out.transpose(0, 3, 1, 2))


def lookup_exact(pyr, coords, radius):
    """floor + lerp lookup of sampler_kernel.cu:39-58, all levels, output [B, L*(2r+1), H, W1]."""
    return lookup(pyr, coords, radius, EXACT)


def lookup_gridsample(pyr, coords, radius):
    """The reg path (core/corr.py:127-146) with bilinear_sampler's coordinate round trip."""
    return lookup(pyr, coords, radius, GRIDSAMPLE)


def lookup_bwd(coords, grad_out, widths, radius, mode=EXACT):
    """Adjoint of `lookup` w.r.t. each level (sampler_kernel.cu:83-103 for EXACT; autograd
    through grid_sample for GRIDSAMPLE).  Returns a list of dense [B,H,W1,W_i] gradients."""
    x = np.asarray(coords, dtype=F32)[:, 0]
    g = np.asarray(grad_out, dtype=F32)
    B, C, H, W1 = g.shape
    nt = 2 * radius + 1
    bb, hh, ww = np.meshgrid(np.arange(B), np.arange(H), np.arange(W1), indexing="ij")
    outs = []
    for i, W in enumerate(widths):
        gi = g[:, i * nt:(i + 1) * nt].transpose(0, 2, 3, 1)      # [B,H,W1,nt]
        i0, w0, w1 = tap_coords(x, i, W, radius, mode)
        dv = np.zeros((B, H, W1, W), dtype=F32)
        for k in range(nt):
            for idx, wt in ((i0[..., k], w0[..., k]), (i0[..., k] + 1, w1[..., k])):
                m = (idx >= 0) & (idx < W)
                np.add.at(dv, (bb[m], hh[m], ww[m], idx[m]), (gi[..., k] * wt)[m])
        outs.append(dv)
    return outs


def lookup_exact_bwd(coords, grad_out, widths, radius):
    return lookup_bwd(coords, grad_out, widths, radius, EXACT)


def pool_pyramid_bwd(level_grads):
    """Fold the gradients of levels 1..L-1 into level 0 (adjoint of pool_pyramid)."""
    acc = np.asarray(level_grads[-1], dtype=F32).copy()
    for lg in reversed(level_grads[:-1]):
        lg = np.asarray(lg, dtype=F32)
        up = np.zeros_like(lg)
        w = acc.shape[-1]
        up[..., 0:2 * w:2] = acc * F32(0.5)
        up[..., 1:2 * w:2] = acc * F32(0.5)
        acc = (lg + up).astype(F32)
    return acc


def corr_volume_bwd(dvol, f1, f2):
    """df1[b,d,h,w1] = sum_w2 dC[b,h,w1,w2] f2[b,d,h,w2] / sqrt(D); df2 likewise."""
    f1 = np.asarray(f1, dtype=F32)
    f2 = np.asarray(f2, dtype=F32)
    D = f1.shape[1]
    s = F32(1) / F32(np.sqrt(F32(D)))
    df1 = np.einsum("bhij,bdhj->bdhi", dvol, f2, optimize=True).astype(F32) * s
    df2 = np.einsum("bhij,bdhi->bdhj", dvol, f1, optimize=True).astype(F32) * s
    return df1.astype(F32), df2.astype(F32)


# ----------------------------------------------------------------------------- alt
def pool_fmap(f2, num_levels):
    """fmap2 <- avg_pool2d(fmap2, [1,2]) per level (core/corr.py:104)."""
    cur = np.asarray(f2, dtype=F32)
    out = [cur]
    for _ in range(1, num_levels):
        w = cur.shape[-1] // 2
        cur = ((cur[..., 0:2 * w:2] + cur[..., 1:2 * w:2]) * F32(0.5)).astype(F32)
        out.append(cur)
    return out


def alt_lookup(f1, f2, coords, num_levels, radius, mode=EXACT):
    """On-the-fly correlation (core/corr.py:89-107): never materialises the volume.
    The fmap2 row is the pixel's own row (coords channel 1 is the row index in RAFT-Stereo,
    core/raft_stereo.py:120); x taps follow `tap_coords` on the pooled fmap2."""
    f1 = np.asarray(f1, dtype=F32)
    B, D, H, W1 = f1.shape
    x = np.asarray(coords, dtype=F32)[:, 0]
    s = F32(1) / F32(np.sqrt(F32(D)))
    f1t = f1.transpose(0, 2, 3, 1)                      # [B,H,W1,D]
    outs = []
    for i, g in enumerate(pool_fmap(f2, num_levels)):
        W = g.shape[-1]
        gt = g.transpose(0, 2, 3, 1)                    # [B,H,W,D]
        i0, w0, w1 = tap_coords(x, i, W, radius, mode)

        def dots(idx):
            ok = (idx >= 0) & (idx < W)
            idc = np.clip(idx, 0, W - 1)
            v = np.zeros(idx.shape, dtype=F32)
            for k in range(idx.shape[-1]):
                rows = np.take_along_axis(gt, idc[..., k][..., None], axis=2)     # [B,H,W1,D]
                v[..., k] = (rows * f1t).sum(-1)
            return np.where(ok, v * s, F32(0)).astype(F32)
        outs.append((dots(i0) * w0 + dots(i0 + 1) * w1).astype(F32))
    out = np.concatenate(outs, axis=-1)
    return np.ascontiguousarray(out.transpose(0, 3, 1, 2))


def convc1_relu(corr, weight, bias, relu=True, round_to=None):
    """First layer of the motion encoder on a lookup result (core/update.py:71,78):
    `F.relu(convc1(corr))`, convc1 = nn.Conv2d(L*(2r+1), 64, 1).  corr [B,K,H,W],
    weight [64,K] (or [64,K,1,1]), bias [64] -> [B,64,H,W].  `round_to` (a function) rounds corr
    and the weight to the 16-bit type first, as autocast does before the convolution."""
    corr = np.asarray(corr, dtype=F32)
    w = np.asarray(weight, dtype=F32).reshape(np.asarray(weight).shape[0], -1)
    if round_to is not None:
        corr, w = round_to(corr), round_to(w)
    out = np.einsum("ok,bkhw->bohw", w.astype(np.float64), corr.astype(np.float64))
    out = out + np.asarray(bias, dtype=np.float64)[None, :, None, None]
    if relu:
        out = np.maximum(out, 0.0)
    return out.astype(F32)


def convc1_relu_bwd(corr, weight, bias, grad_act, relu=True):
    """Adjoint of convc1_relu (autograd of core/update.py:71,78): (d corr, d weight [64,K], d bias [64])."""
    corr = np.asarray(corr, dtype=np.float64)
    w = np.asarray(weight, dtype=np.float64).reshape(np.asarray(weight).shape[0], -1)
    pre = np.einsum("ok,bkhw->bohw", w, corr) + np.asarray(bias, dtype=np.float64)[None, :, None, None]
    gm = np.asarray(grad_act, dtype=np.float64) * ((pre > 0) if relu else 1.0)
    return (np.einsum("ok,bohw->bkhw", w, gm).astype(F32), np.einsum("bohw,bkhw->ok", gm, corr).astype(F32),
            gm.sum(axis=(0, 2, 3)).astype(F32))


# ----------------------------------------------------------------------------- helpers
def sampler_fwd_rounded(volume, coords, radius, rnd):
    """sampler_forward_kernel with `scalar_t` a 16-bit type (sampler/sampler_kernel.cu:39-58): the lerp
    weights are converted to scalar_t (:53, :56), every product is rounded to scalar_t and `+=` rounds the
    sum again.  `rnd` rounds a float32 array to the storage type and back (np.float16 round trip, or
    round_bf16).  volume [B,H,W1,W2] (values already representable), coords [B,>=1,H,W1] -> [B,2r+1,H,W1]."""
    vol = np.asarray(volume, dtype=F32)
    x0 = np.asarray(coords, dtype=F32)[:, 0]
    B, H, W1, W2 = vol.shape
    fx = np.floor(x0)
    dx = (x0 - fx).astype(F32)
    wa, wb = rnd((F32(1.0) - dx).astype(F32)), rnd(dx)
    base = np.where(np.abs(fx) < 1e8, fx, -2.0 ** 28).astype(np.int64) - radius
    rd = 2 * radius + 1
    s = np.zeros((B, H, W1, rd + 1), dtype=F32)
    for i in range(rd + 1):
        x1 = base + i
        ok = (x1 >= 0) & (x1 < W2)
        s[..., i] = np.where(ok, np.take_along_axis(vol, np.clip(x1, 0, W2 - 1)[..., None], axis=3)[..., 0], 0)
    out = np.zeros((B, rd, H, W1), dtype=F32)
    for k in range(rd):
        out[:, k] = rnd(rnd(s[..., k] * wa) + rnd(s[..., k + 1] * wb))
    return out


def sampler_bwd_rounded(coords, corr_grad, W2, radius, rnd):
    """sampler_backward_kernel with a 16-bit scalar_t (sampler/sampler_kernel.cu:83-103): g = 0;
    g += corr_grad[i-1] * scalar_t(dx); g += corr_grad[i] * scalar_t(1 - dx), each product and each sum rounded
    to scalar_t; volume_grad is zero elsewhere.  corr_grad [B,2r+1,H,W1] -> [B,H,W1,W2]."""
    g = np.asarray(corr_grad, dtype=F32)
    x0 = np.asarray(coords, dtype=F32)[:, 0]
    B, rd, H, W1 = g.shape
    fx = np.floor(x0)
    dx = (x0 - fx).astype(F32)
    wa, wb = rnd((F32(1.0) - dx).astype(F32)), rnd(dx)
    base = np.where(np.abs(fx) < 1e8, fx, -2.0 ** 28).astype(np.int64) - radius
    out = np.zeros((B, H, W1, W2), dtype=F32)
    for i in range(rd + 1):
        acc = np.zeros((B, H, W1), dtype=F32)
        if i > 0:
            acc = rnd(acc + rnd(g[:, i - 1] * wb))
        if i < rd:
            acc = rnd(acc + rnd(g[:, i] * wa))
        x1 = base + i
        ok = (x1 >= 0) & (x1 < W2)
        bi, hi, wi = np.nonzero(ok)
        out[bi, hi, wi, x1[ok]] = acc[ok]
    return out


def rel_to_max(a, b):
    """The parity metric of SURVEY.md 8(d): max|a-b| / max|b|."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / den


def round_bf16(x):
    """float32 -> bfloat16 (round to nearest even) -> float32, in numpy."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(F32).reshape(x.shape)


def synthetic_coords(B, H, W, seed=1234, adversarial=False, max_disp=None):
    """SURVEY.md 8(d) synthetic coords: x = w - disp, disp piecewise smooth in [0, W/2]."""
    rng = np.random.default_rng(seed)
    xs = np.arange(W, dtype=F32)[None, None, :].repeat(B, 0).repeat(H, 1)
    ys = np.arange(H, dtype=F32)[None, :, None].repeat(B, 0).repeat(W, 2)
    if adversarial:
        disp = rng.uniform(0, W, size=(B, H, W)).astype(F32)
    else:
        md = W / 3 if max_disp is None else max_disp
        coarse = rng.uniform(0, md, size=(B, 8, 8)).astype(F32)
        gy = np.linspace(0, 7, H, dtype=F32)
        gx = np.linspace(0, 7, W, dtype=F32)
        y0 = np.clip(np.floor(gy).astype(int), 0, 6)
        x0 = np.clip(np.floor(gx).astype(int), 0, 6)
        fy = (gy - y0)[None, :, None]
        fx = (gx - x0)[None, None, :]
        c00 = coarse[:, y0][:, :, x0]
        c01 = coarse[:, y0][:, :, x0 + 1]
        c10 = coarse[:, y0 + 1][:, :, x0]
        c11 = coarse[:, y0 + 1][:, :, x0 + 1]
        smooth = (c00 * (1 - fy) * (1 - fx) + c01 * (1 - fy) * fx + c10 * fy * (1 - fx) + c11 * fy * fx)
        disp = np.clip(smooth + rng.uniform(-0.5, 0.5, size=(B, H, W)), 0, W / 2).astype(F32)
    return np.ascontiguousarray(np.stack([xs - disp, ys], axis=1).astype(F32))
