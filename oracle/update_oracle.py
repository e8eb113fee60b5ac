"""CPU oracle for the pointwise work of a GRU iteration (numpy, float32): ConvGRU gate arithmetic and
the convex upsampling of the disparity field (SURVEY.md section 8 f, ranks 3 and 4).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may.

Every function restates one piece of the reference and cites it (paths are relative to the upstream
tree, which is NOT shipped):

  gru_gate_r        core/update.py:28-29      r = sigmoid(convr(hx) + cr);  r * h
  gru_update        core/update.py:27,29,31   z, q, h = (1-z)*h + z*q
  gru_update_bwd,
  gru_gate_r_bwd    autograd of the same lines
  upsample_flow     core/raft_stereo.py:55-67 softmax over 9, unfold(factor*flow, 3x3, padding=1), sum, permute
  upsample_flow_bwd autograd of the same lines
  pool2x, interp    core/update.py:87-95      3x3 stride-2 mean; bilinear resize, align_corners=True
  sequence_loss     train_stereo.py:35-70     masked L1 over all predictions, EPE metrics, and its gradient

Parity pin: tests/test_oracle_golden.py checks these against vectors captured from the reference's own
Python (tests/golden/make_golden_update.py ran `core.update.ConvGRU` and `RAFTStereo.upsample_flow` on
CPU in the build container and recorded the convolution outputs next to the results; make_golden_loss.py
ran the reference's `sequence_loss` function).  The reference
has no tests or fixtures of its own for these lines.
"""
import numpy as np

F32 = np.float32


def _rounder(round_to):
    """Every torch op on a 16-bit tensor evaluates in fp32 and rounds its result to the tensor type;
    `round_to` (None, np.float16, or a callable for bf16) re-applies that rounding after each step."""
    if round_to is None:
        return lambda x: np.asarray(x, dtype=F32)
    if callable(round_to) and not isinstance(round_to, type):
        return lambda x: np.asarray(round_to(np.asarray(x, dtype=F32)), dtype=F32)
    return lambda x: np.asarray(x, dtype=F32).astype(round_to).astype(F32)


def _sigmoid(x):
    x = np.asarray(x, dtype=np.float64)
    return (1.0 / (1.0 + np.exp(-x))).astype(F32)


def _tanh(x):
    return np.tanh(np.asarray(x, dtype=np.float64)).astype(F32)


def gru_gate_r(convr_out, cr, h, round_to=None):
    """r * h with r = sigmoid(convr_out + cr)   (core/update.py:28-29)."""
    rd = _rounder(round_to)
    r = rd(_sigmoid(rd(rd(convr_out) + rd(cr))))
    return rd(r * rd(h))


def gru_update(convz_out, cz, convq_out, cq, h, round_to=None):
    """h' = (1-z)*h + z*q,  z = sigmoid(convz_out + cz),  q = tanh(convq_out + cq)   (core/update.py:27,29,31)."""
    rd = _rounder(round_to)
    z = rd(_sigmoid(rd(rd(convz_out) + rd(cz))))
    q = rd(_tanh(rd(rd(convq_out) + rd(cq))))
    return rd(rd(rd(F32(1) - z) * rd(h)) + rd(z * q))


def gru_update_bwd(convz_out, cz, convq_out, cq, h, grad_out):
    """Adjoint of gru_update: (d z_pre, d q_pre, d h)."""
    z = _sigmoid(np.asarray(convz_out, F32) + np.asarray(cz, F32)).astype(np.float64)
    q = _tanh(np.asarray(convq_out, F32) + np.asarray(cq, F32)).astype(np.float64)
    g, h = np.asarray(grad_out, np.float64), np.asarray(h, np.float64)
    return ((g * (q - h) * z * (1 - z)).astype(F32), (g * z * (1 - q * q)).astype(F32), (g * (1 - z)).astype(F32))


def gru_gate_r_bwd(convr_out, cr, h, grad_rh):
    """Adjoint of gru_gate_r: (d r_pre, d h)."""
    r = _sigmoid(np.asarray(convr_out, F32) + np.asarray(cr, F32)).astype(np.float64)
    g, h = np.asarray(grad_rh, np.float64), np.asarray(h, np.float64)
    return (g * h * r * (1 - r)).astype(F32), (g * r).astype(F32)


def upsample_flow(flow, mask, factor, round_to=None):
    """[N,D,H,W] flow + [N,9*f*f,H,W] mask logits -> [N,D,f*H,f*W]   (core/raft_stereo.py:55-67).

    mask.view(N,1,9,f,f,H,W) -> softmax over the 9 -> sum_k mask_k * unfold(f*flow)_k -> permute(0,1,4,2,5,3):
    out[n,d,f*h+fy,f*w+fx].  `round_to` is the mask's storage type: torch.softmax returns that type."""
    rd = _rounder(round_to)
    flow = np.asarray(flow, dtype=F32)
    N, D, H, W = flow.shape
    f = factor
    m = rd(mask).reshape(N, 9, f, f, H, W).astype(np.float64)
    m = m - m.max(axis=1, keepdims=True)
    e = np.exp(m)
    p = rd((e / e.sum(axis=1, keepdims=True)).astype(F32))                  # [N,9,f,f,H,W]
    pad = np.zeros((N, D, H + 2, W + 2), dtype=F32)
    pad[:, :, 1:-1, 1:-1] = F32(f) * flow
    out = np.zeros((N, D, f, f, H, W), dtype=F32)
    for k in range(9):                                                      # unfold order: k = ky*3 + kx
        ky, kx = divmod(k, 3)
        nb = pad[:, :, ky:ky + H, kx:kx + W]                                # [N,D,H,W]
        out += p[:, None, k] * nb[:, :, None, None]
    return np.ascontiguousarray(out.transpose(0, 1, 4, 2, 5, 3)).reshape(N, D, f * H, f * W)


def upsample_flow_bwd(flow, mask, factor, grad_out, round_to=None):
    """Adjoint of upsample_flow (autograd of core/raft_stereo.py:59-67): (d flow, d mask).

    The gradient of the probabilities (sum over D of grad * unfold(f*flow)) and the softmax backward
    p * (gp - sum p gp) are rounded to the mask's storage type where torch's 16-bit tensors round."""
    rd = _rounder(round_to)
    flow = np.asarray(flow, dtype=F32)
    N, D, H, W = flow.shape
    f = factor
    m = rd(mask).reshape(N, 9, f, f, H, W).astype(np.float64)
    m = m - m.max(axis=1, keepdims=True)
    e = np.exp(m)
    p = rd((e / e.sum(axis=1, keepdims=True)).astype(F32)).astype(np.float64)       # [N,9,f,f,H,W]
    g = np.asarray(grad_out, np.float64).reshape(N, D, H, f, W, f).transpose(0, 1, 3, 5, 2, 4)  # [N,D,f,f,H,W]
    pad = np.zeros((N, D, H + 2, W + 2), dtype=np.float64)
    pad[:, :, 1:-1, 1:-1] = F32(f) * flow
    gp = np.zeros((N, 9, f, f, H, W), dtype=np.float64)
    dpad = np.zeros_like(pad)
    for k in range(9):
        ky, kx = divmod(k, 3)
        nb = pad[:, :, ky:ky + H, kx:kx + W]                                          # [N,D,H,W]
        gp[:, k] = (g * nb[:, :, None, None]).sum(axis=1)
        dpad[:, :, ky:ky + H, kx:kx + W] += (p[:, None, k] * g).sum(axis=(2, 3))
    gp = rd(gp.astype(F32)).astype(np.float64)
    dmask = rd((p * (gp - (p * gp).sum(axis=1, keepdims=True))).astype(F32)).reshape(N, 9 * f * f, H, W)
    return (f * dpad[:, :, 1:-1, 1:-1]).astype(F32), dmask


def sequence_loss(flow_preds, flow_gt, valid, loss_gamma=0.9, max_flow=700):
    """(loss, metrics, grads) of train_stereo.py:35-70 for one flow channel.

    valid' = (valid >= 0.5) & (|gt| < max_flow); loss = sum_i gamma'^(n-i-1) * mean_valid' |pred_i - gt| with
    gamma' = gamma^(15/(n-1)); epe = |pred_last - gt| on valid' -> mean, fraction < 1, < 3, < 5; the gradient
    of the loss w.r.t. pred_i is w_i * sign(pred_i - gt) / #valid' on valid'."""
    n = len(flow_preds)
    gt = np.asarray(flow_gt, dtype=F32)
    mag = np.sqrt(np.sum(gt.astype(F32) ** 2, axis=1))
    ok = ((np.asarray(valid, F32) >= 0.5) & (mag < max_flow))[:, None]
    cnt = int(ok.sum())
    g = loss_gamma ** (15 / (n - 1))
    loss, grads = 0.0, []
    for i, p in enumerate(flow_preds):
        w = g ** (n - i - 1)
        d = np.asarray(p, F32) - gt
        loss += w * float(np.abs(d)[ok].astype(np.float64).mean())
        grads.append(np.where(ok, w * np.sign(d) / cnt, 0.0).astype(F32))
    epe = np.sqrt(np.sum((np.asarray(flow_preds[-1], F32) - gt) ** 2, axis=1))[ok[:, 0]].astype(np.float64)
    metrics = {"epe": float(epe.mean()), "1px": float((epe < 1).mean()), "3px": float((epe < 3).mean()),
               "5px": float((epe < 5).mean())}
    return loss, metrics, grads


def pool2x(x):
    """avg_pool2d(x, 3, stride=2, padding=1), padding counted in the mean   (core/update.py:87-88)."""
    x = np.asarray(x, dtype=np.float64)
    N, C, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    pad = np.zeros((N, C, 2 * Ho + 1, 2 * Wo + 1))
    pad[:, :, 1:H + 1, 1:W + 1] = x
    out = np.zeros((N, C, Ho, Wo))
    for dy in range(3):
        for dx in range(3):
            out += pad[:, :, dy:dy + 2 * Ho:2, dx:dx + 2 * Wo:2]
    return (out / 9.0).astype(F32)


def interp(x, Ho, Wo):
    """F.interpolate(x, (Ho, Wo), mode='bilinear', align_corners=True)   (core/update.py:93-95)."""
    x = np.asarray(x, dtype=np.float64)
    H, W = x.shape[2:]
    # torch computes the source positions in fp32: scale = (in-1)/(out-1), position = scale * index
    sy = F32(H - 1) / F32(Ho - 1) if Ho > 1 else F32(0)
    sx = F32(W - 1) / F32(Wo - 1) if Wo > 1 else F32(0)
    fy, fx = (sy * np.arange(Ho, dtype=F32)).astype(F32), (sx * np.arange(Wo, dtype=F32)).astype(F32)
    y0, x0 = np.minimum(fy.astype(int), H - 1), np.minimum(fx.astype(int), W - 1)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    ly, lx = (fy - y0.astype(F32)).astype(np.float64)[:, None], (fx - x0.astype(F32)).astype(np.float64)[None, :]
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return (top * (1 - ly) + bot * ly).astype(F32)


def pool2x_bwd(grad_out, H, W):
    """Adjoint of pool2x for an [N,C,H,W] input."""
    g = np.asarray(grad_out, dtype=np.float64)
    N, C, Ho, Wo = g.shape
    dpad = np.zeros((N, C, 2 * Ho + 1, 2 * Wo + 1))
    for dy in range(3):
        for dx in range(3):
            dpad[:, :, dy:dy + 2 * Ho:2, dx:dx + 2 * Wo:2] += g / 9.0
    return dpad[:, :, 1:H + 1, 1:W + 1].astype(F32)


def interp_bwd(grad_out, H, W):
    """Adjoint of interp for an [N,C,H,W] input (same fp32 source positions as the forward)."""
    g = np.asarray(grad_out, dtype=np.float64)
    N, C, Ho, Wo = g.shape

    def axis(n_in, n_out):
        s = F32(n_in - 1) / F32(n_out - 1) if n_out > 1 else F32(0)
        f = (s * np.arange(n_out, dtype=F32)).astype(F32)
        i0 = np.minimum(f.astype(int), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l = (f - i0.astype(F32)).astype(np.float64)
        m = np.zeros((n_out, n_in))
        np.add.at(m, (np.arange(n_out), i0), 1 - l)
        np.add.at(m, (np.arange(n_out), i1), l)
        return m
    return np.einsum("ncop,oy,px->ncyx", g, axis(H, Ho), axis(W, Wo)).astype(F32)


def rel_to_max(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
