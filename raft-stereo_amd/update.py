"""The pointwise work of a GRU iteration around the correlation lookup, backed by libraftcorr_hip.so
(SURVEY.md section 8 f, ranks 3 and 4).

    ConvGRUHip(hidden_dim, input_dim, kernel_size=3)      core/update.py:16-32  ConvGRU
        .forward(h, cz, cr, cq, *x_list)                  same parameters (convz/convr/convq), same call
    upsample_flow_hip(flow, mask, n_downsample)           core/raft_stereo.py:55-67  RAFTStereo.upsample_flow
    sequence_loss_hip(flow_preds, flow_gt, valid, ...)    train_stereo.py:35-70      sequence_loss

The three convolutions of the GRU are torch's (dense convolutions are outside this package's path); the
sigmoid / tanh / blend arithmetic around them and the whole convex upsampling run in the HIP library.
There is no CPU or eager fallback: tensors must live on a GPU.
"""
import ctypes

import torch
import torch.nn as nn

from raft_stereo_amd import _lib
from raft_stereo_amd._lib import lib, check
from raft_stereo_amd.corr import _DT, _stream, _on_device


def _batch_view(t, like, what):
    """`t` as (pointer, batch stride) of an [N,C,H,W] operand whose images are dense: either contiguous or
    a channel slice of a wider contiguous tensor (the `.split(dim=1)` views of core/raft_stereo.py:88).
    Anything else is copied."""
    if t.shape != like.shape or t.dtype != like.dtype or t.device != like.device:
        raise ValueError(f"{what} must match h: {tuple(like.shape)} {like.dtype} on {like.device}, "
                         f"got {tuple(t.shape)} {t.dtype} on {t.device}")
    N, C, H, W = t.shape
    if t.stride()[1:] != (H * W, W, 1) or (N > 1 and t.stride(0) < C * H * W):
        t = t.contiguous()
    return t, (t.stride(0) if N > 1 else C * H * W)


class _GateRFn(torch.autograd.Function):
    """rh = sigmoid(convr_out + cr) * h (out of place; the differentiable route)."""

    @staticmethod
    def forward(ctx, convr_out, cr, h):
        cr_t, cr_bs = _batch_view(cr, h, "cr")
        out = torch.empty_like(h)
        _gate_r_raw(convr_out, cr_t, cr_bs, h, out)
        ctx.save_for_backward(convr_out, cr_t, h)
        return out

    @staticmethod
    def backward(ctx, grad_rh):
        convr_out, cr_t, h = ctx.saved_tensors
        g, g_bs = _batch_view(grad_rh, h, "grad_rh")
        N, per = h.shape[0], h[0].numel()
        d_pre, d_h = torch.empty_like(h), torch.empty_like(h)
        with _on_device(h.device):
            check(lib.rc_gru_gate_r_bwd(convr_out.data_ptr(), cr_t.data_ptr(), cr_t.stride(0) if N > 1 else per,
                                        h.data_ptr(), g.data_ptr(), g_bs, d_pre.data_ptr(), d_h.data_ptr(), 0,
                                        N, per, _DT[h.dtype], _stream(h.device)), "rc_gru_gate_r_bwd")
        return d_pre, d_pre, d_h


def _gate_r_raw(convr_out, cr_t, cr_bs, h, dst):
    N, per = h.shape[0], h[0].numel()
    convr_out, conv_bs = _batch_view(convr_out, h, "convr_out")
    with _on_device(h.device):
        check(lib.rc_gru_gate_r(convr_out.data_ptr(), conv_bs, cr_t.data_ptr(), cr_bs, h.data_ptr(), dst.data_ptr(),
                                dst.stride(0) if N > 1 else per, N, per, _DT[h.dtype], _stream(h.device)),
              "rc_gru_gate_r")


class _UpdateFn(torch.autograd.Function):
    """h_out = (1-z)*h + z*q with z = sigmoid(convz_out + cz), q = tanh(convq_out + cq)."""

    @staticmethod
    def forward(ctx, convz_out, cz, convq_out, cq, h):
        cz_t, cz_bs = _batch_view(cz, h, "cz")
        cq_t, cq_bs = _batch_view(cq, h, "cq")
        N, per = h.shape[0], h[0].numel()
        out = torch.empty_like(h)
        with _on_device(h.device):
            check(lib.rc_gru_update(convz_out.data_ptr(), per, cz_t.data_ptr(), cz_bs, convq_out.data_ptr(), per,
                                    cq_t.data_ptr(), cq_bs, h.data_ptr(), out.data_ptr(), N, per, _DT[h.dtype],
                                    _stream(h.device)), "rc_gru_update")
        ctx.save_for_backward(convz_out, cz_t, convq_out, cq_t, h)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        convz_out, cz_t, convq_out, cq_t, h = ctx.saved_tensors
        N, per = h.shape[0], h[0].numel()
        g = grad_out.contiguous()
        dz, dq, dh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        with _on_device(h.device):
            check(lib.rc_gru_update_bwd(convz_out.data_ptr(), cz_t.data_ptr(), cz_t.stride(0) if N > 1 else per,
                                        convq_out.data_ptr(), cq_t.data_ptr(), cq_t.stride(0) if N > 1 else per,
                                        h.data_ptr(), g.data_ptr(), dz.data_ptr(), dq.data_ptr(), dh.data_ptr(),
                                        N, per, _DT[h.dtype], _stream(h.device)), "rc_gru_update_bwd")
        return dz, dz, dq, dq, dh


def _check_gru_operand(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"raft_stereo_amd GRU kernels need GPU tensors (there is no CPU path): {what}")
    if t.dtype not in _DT:
        raise TypeError(f"unsupported dtype {t.dtype} for {what}")
    if t.dim() != 4:
        raise ValueError(f"{what} must be [N, C, H, W]")


def gru_gate_r(convr_out, cr, h, out=None):
    """`sigmoid(convr_out + cr) * h` (core/update.py:28-29).  `out`, if given (inference only), is an
    [N,C,H,W] channel slice of a wider buffer: the first hidden_dim channels of `torch.cat([r*h, x], dim=1)`."""
    for t, w in ((convr_out, "convr_out"), (cr, "cr"), (h, "h")):
        _check_gru_operand(t, w)
    h = h.contiguous()
    if convr_out.shape != h.shape or convr_out.dtype != h.dtype:
        raise ValueError("convr_out must match h")
    if out is None:
        return _GateRFn.apply(convr_out.contiguous(), cr, h)
    if torch.is_grad_enabled() and (convr_out.requires_grad or cr.requires_grad or h.requires_grad):
        raise RuntimeError("gru_gate_r(out=...) writes in place and is forward-only")
    if out.shape != h.shape or out.dtype != h.dtype or out.stride()[1:] != h.stride()[1:] \
            or (h.shape[0] > 1 and out.stride(0) < h[0].numel()):
        raise ValueError("out must be an [N,C,H,W] tensor of h's type whose images are dense")
    cr_t, cr_bs = _batch_view(cr.detach(), h, "cr")
    _gate_r_raw(convr_out.detach(), cr_t, cr_bs, h.detach(), out)
    return out


def gru_update(convz_out, cz, convq_out, cq, h):
    """`(1-z)*h + z*q` with `z = sigmoid(convz_out + cz)`, `q = tanh(convq_out + cq)` (core/update.py:27-31)."""
    for t, w in ((convz_out, "convz_out"), (cz, "cz"), (convq_out, "convq_out"), (cq, "cq"), (h, "h")):
        _check_gru_operand(t, w)
    h = h.contiguous()
    if convz_out.shape != h.shape or convq_out.shape != h.shape or convz_out.dtype != h.dtype \
            or convq_out.dtype != h.dtype:
        raise ValueError("convz_out and convq_out must match h")
    if torch.is_grad_enabled() and any(t.requires_grad for t in (convz_out, cz, convq_out, cq, h)):
        return _UpdateFn.apply(convz_out.contiguous(), cz, convq_out.contiguous(), cq, h)
    # inference: the convolution outputs may be channel slices of a stacked convolution's result
    zc, z_bs = _batch_view(convz_out.detach(), h, "convz_out")
    qc, q_bs = _batch_view(convq_out.detach(), h, "convq_out")
    cz_t, cz_bs = _batch_view(cz.detach(), h, "cz")
    cq_t, cq_bs = _batch_view(cq.detach(), h, "cq")
    N, per = h.shape[0], h[0].numel()
    out = torch.empty_like(h)
    with _on_device(h.device):
        check(lib.rc_gru_update(zc.data_ptr(), z_bs, cz_t.data_ptr(), cz_bs, qc.data_ptr(), q_bs, cq_t.data_ptr(),
                                cq_bs, h.data_ptr(), out.data_ptr(), N, per, _DT[h.dtype], _stream(h.device)),
              "rc_gru_update")
    return out


def bias_act_(y, bias, relu=False):
    """In place `y = act(y + bias[c])` on a bias-free convolution's output [N,C,H,W] (rc_bias_act): one pass
    instead of torch's broadcast add_ and ReLU, same roundings.  Inference only."""
    if not y.is_cuda or y.dtype not in _DT or y.dim() != 4 or not y.is_contiguous():
        raise RuntimeError("bias_act_ needs a contiguous GPU [N,C,H,W] tensor (there is no CPU path)")
    if bias.dtype not in (y.dtype, torch.float32) or bias.numel() != y.shape[1] or not bias.is_contiguous():
        raise ValueError("bias must be a contiguous [C] tensor of y's type or fp32")
    if torch.is_grad_enabled() and (y.requires_grad or bias.requires_grad):
        raise RuntimeError("bias_act_ writes in place and is forward-only")
    N, C, H, W = y.shape
    if y.numel():
        with _on_device(y.device):
            check(lib.rc_bias_act(y.data_ptr(), bias.data_ptr(), _DT[bias.dtype], N, C, H * W, _DT[y.dtype],
                                  int(bool(relu)), _stream(y.device)), "rc_bias_act")
    return y


def conv_bias_act(conv, x, relu=False):
    """`conv(x)` / `F.relu(conv(x))` of an nn.Conv2d.  In inference on the GPU the convolution runs without its
    bias and the library adds bias and ReLU in one in-place pass; with gradients (or anything the kernel does not
    take) it is the torch expression itself."""
    plain = (conv.bias is None or not x.is_cuda or conv.padding_mode != "zeros"
             or (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or conv.bias.requires_grad)))
    if not plain:
        y = nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        bias = conv.bias.detach()
        if y.dtype in _DT and y.is_contiguous() and bias.dtype in (y.dtype, torch.float32):
            return bias_act_(y, bias, relu)
        y = y + bias.to(y.dtype).view(1, -1, 1, 1)
        return torch.relu(y) if relu else y
    y = conv(x)
    return torch.relu(y) if relu else y


class ConvGRUHip(nn.Module):
    """Drop-in for the reference's ConvGRU (core/update.py:16-32): same constructor, same parameter names
    (a reference state_dict loads as is), same `forward(h, cz, cr, cq, *x_list)`.  convz and convr read the
    same input, so they run as ONE convolution with stacked weights; everything pointwise is two library
    kernels instead of eleven torch ops."""

    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)

    def _stacked(self, dt):
        """convz and convr as one convolution: stacked weight and bias, cached while the parameters stand still.
        When they change (load_state_dict, an optimizer step) the cached buffers are refreshed IN PLACE, so a HIP
        graph captured over the inference path keeps reading current values on replay; tensors whose version
        counter cannot be read (inference-mode tensors) are re-stacked on every call."""
        params = (self.convz.weight, self.convr.weight, self.convz.bias, self.convr.bias)
        try:
            key = (dt,) + tuple(p._version for p in params) + tuple(p.data_ptr() for p in params)
        except RuntimeError:
            key = None
        if key is None or getattr(self, "_zr_key", None) != key:
            with torch.no_grad():
                w = torch.cat([self.convz.weight, self.convr.weight], dim=0).to(dt)
                b = torch.cat([self.convz.bias, self.convr.bias], dim=0).to(dt)
                old = getattr(self, "_zr_w", None)
                # (an in-place update of a tensor made under inference_mode is only legal under inference_mode)
                if old is not None and old.shape == w.shape and old.dtype == w.dtype and old.device == w.device \
                        and not (old.is_inference() and not torch.is_inference_mode_enabled()):
                    self._zr_w.copy_(w)
                    self._zr_b.copy_(b)
                else:
                    self._zr_w, self._zr_b = w, b
            self._zr_key = key
        return self._zr_w, self._zr_b

    def forward(self, h, cz, cr, cq, *x_list):
        C = self.hidden_dim
        if torch.is_autocast_enabled():           # what autocast would hand the convolutions
            dt = torch.get_autocast_dtype("cuda")
        else:
            dt = h.dtype
            for t in x_list:
                dt = torch.promote_types(dt, t.dtype)
        # update.py:24-25; operands are cast BEFORE the concatenation (same values, and a fp32 member -- the motion
        # features carry the fp32 flow, update.py:85 -- no longer drags the whole buffer through fp32)
        hx = torch.cat([t.to(dt) for t in (h, *x_list)], dim=1)
        h, cz, cr, cq = h.to(dt).contiguous(), cz.to(dt), cr.to(dt), cq.to(dt)
        # any trainable parameter (not only convz: a fine-tune may freeze some of them) or input takes the
        # autograd route
        train = torch.is_grad_enabled() and (hx.requires_grad or cr.requires_grad or cz.requires_grad
                                             or cq.requires_grad or any(p.requires_grad for p in self.parameters()))
        if train:
            zr = nn.functional.conv2d(hx, torch.cat([self.convz.weight, self.convr.weight], dim=0).to(dt),
                                      torch.cat([self.convz.bias, self.convr.bias], dim=0).to(dt),
                                      padding=self.convz.padding)
            rhx = torch.cat([gru_gate_r(zr[:, C:], cr, h), hx[:, C:]], dim=1)    # [r*h, x]  update.py:28-29
            qc = self.convq(rhx)
            return gru_update(zr[:, :C], cz, qc.to(dt), cq, h)                    # update.py:27,29,31
        with torch.no_grad():
            w, b = self._stacked(dt)
            zr = bias_act_(nn.functional.conv2d(hx, w, None, padding=self.convz.padding), b)
            gru_gate_r(zr[:, C:], cr, h, out=hx[:, :C])   # hx is this call's own copy; both convolutions have read it
            qc = conv_bias_act(self.convq, hx)
            return gru_update(zr[:, :C], cz, qc.to(dt), cq, h)


def _resample_ok(x):
    return x.is_cuda and x.dim() == 4 and x.dtype in _DT


class _Pool2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        out = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device)
        with _on_device(x.device):
            check(lib.rc_pool2x(x.data_ptr(), out.data_ptr(), N, C, H, W, out[0].numel(), _DT[x.dtype],
                                _stream(x.device)), "rc_pool2x")
        ctx.shape = (N, C, H, W)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        N, C, H, W = ctx.shape
        g = grad_out.contiguous()
        dx = torch.empty((N, C, H, W), dtype=g.dtype, device=g.device)
        with _on_device(g.device):
            check(lib.rc_pool2x_bwd(g.data_ptr(), dx.data_ptr(), N, C, H, W, _DT[g.dtype], _stream(g.device)),
                  "rc_pool2x_bwd")
        return dx


class _InterpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        N, C, H, W = x.shape
        out = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device)
        with _on_device(x.device):
            check(lib.rc_interp_bilinear(x.data_ptr(), out.data_ptr(), N, C, H, W, Ho, Wo, out[0].numel(),
                                         _DT[x.dtype], _stream(x.device)), "rc_interp_bilinear")
        ctx.shape = (N, C, H, W, Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        N, C, H, W, Ho, Wo = ctx.shape
        g = grad_out.contiguous()
        dx = torch.empty((N, C, H, W), dtype=g.dtype, device=g.device)
        with _on_device(g.device):
            check(lib.rc_interp_bilinear_bwd(g.data_ptr(), dx.data_ptr(), N, C, H, W, Ho, Wo, _DT[g.dtype],
                                             _stream(g.device)), "rc_interp_bilinear_bwd")
        return dx, None, None


def pool2x_hip(x):
    """`F.avg_pool2d(x, 3, stride=2, padding=1)` (core/update.py:87-88) in the library, differentiable."""
    if not _resample_ok(x):
        raise RuntimeError("pool2x_hip needs a GPU [N,C,H,W] tensor (there is no CPU path)")
    return _Pool2xFn.apply(x.contiguous())


def interp_hip(x, dest):
    """`F.interpolate(x, dest.shape[2:], mode='bilinear', align_corners=True)` (core/update.py:93-95) in the
    library, in x's own type (fp32 arithmetic), differentiable."""
    if not _resample_ok(x):
        raise RuntimeError("interp_hip needs a GPU [N,C,H,W] tensor (there is no CPU path)")
    return _InterpFn.apply(x.contiguous(), int(dest.shape[2]), int(dest.shape[3]))


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, mask, f, C):
        N, D, H, W = flow.shape
        with _on_device(flow.device):
            out = torch.empty((N, C, f * H, f * W), dtype=torch.float32, device=flow.device)
            check(lib.rc_upsample_flow_fwd(flow.data_ptr(), D * H * W, mask.data_ptr(), out.data_ptr(), N, C, H, W,
                                           f, _DT[mask.dtype], _stream(flow.device)), "rc_upsample_flow_fwd")
        ctx.save_for_backward(flow, mask)
        ctx.f, ctx.C = f, C
        return out

    @staticmethod
    def backward(ctx, grad_out):
        flow, mask = ctx.saved_tensors
        f, C = ctx.f, ctx.C
        N, D, H, W = flow.shape
        g = grad_out.float().contiguous()
        with _on_device(flow.device):
            d_mask = torch.empty_like(mask)
            d_flow = torch.empty_like(flow) if C == D else torch.zeros_like(flow)
            nbytes = lib.rc_upsample_flow_bwd_work_bytes(N, C, H, W)
            work = torch.empty(int(nbytes), dtype=torch.uint8, device=flow.device)
            check(lib.rc_upsample_flow_bwd(flow.data_ptr(), D * H * W, mask.data_ptr(), g.data_ptr(),
                                           d_mask.data_ptr(), d_flow.data_ptr(), D * H * W, work.data_ptr(), N, C,
                                           H, W, f, _DT[mask.dtype], _stream(flow.device)), "rc_upsample_flow_bwd")
        return d_flow, d_mask, None, None


def upsample_flow_hip(flow, mask, n_downsample, channels=None):
    """RAFTStereo.upsample_flow (core/raft_stereo.py:55-67): `[N, D, H, W]` flow and `[N, 9*f*f, H, W]` mask
    logits -> `[N, D, f*H, f*W]` fp32, f = 2**n_downsample.  `channels` = 1 computes only the x component (all
    that core/raft_stereo.py:134 keeps).  Differentiable w.r.t. flow and mask."""
    if not (flow.is_cuda and mask.is_cuda) or flow.device != mask.device:
        raise RuntimeError("upsample_flow_hip needs GPU tensors on one device (there is no CPU path)")
    if flow.dim() != 4 or mask.dim() != 4:
        raise ValueError("flow must be [N, D, H, W] and mask [N, 9*f*f, H, W]")
    f = 2 ** n_downsample
    N, D, H, W = flow.shape
    C = D if channels is None else channels
    if C not in (1, 2) or C > D:
        raise ValueError(f"channels must be 1 or 2 and at most D={D}")
    if tuple(mask.shape) != (N, 9 * f * f, H, W):
        raise ValueError(f"mask must be [{N}, {9 * f * f}, {H}, {W}], got {tuple(mask.shape)}")
    if mask.dtype not in _DT:
        raise TypeError(f"unsupported mask dtype {mask.dtype}")
    return _UpsampleFn.apply(flow.float().contiguous(), mask.contiguous(), f, C)


def _pred_view(p, N, H, W):
    """A prediction as (tensor, batch stride): float [N,1,H,W] whose images are dense (a `[:, :1]` slice of a
    2-channel tensor qualifies); anything else is copied."""
    if p.dtype is not torch.float32 or p.stride()[2:] != (W, 1) or (N > 1 and p.stride(0) < H * W):
        p = p.float().contiguous()
    return p, (p.stride(0) if N > 1 else H * W)


class _SeqLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow_gt, valid, max_flow, weights, *preds):
        n = len(preds)
        N, _, H, W = flow_gt.shape
        dev = flow_gt.device
        views = [_pred_view(p, N, H, W) for p in preds]
        bs = views[0][1]
        if any(v[1] != bs for v in views):
            views = [(v[0].contiguous(), H * W) for v in views]
            bs = H * W
        ptrs = (ctypes.c_void_p * n)(*[v[0].data_ptr() for v in views])
        wts = (ctypes.c_float * n)(*weights)
        with _on_device(dev):
            results = torch.empty(n + 7, dtype=torch.float64, device=dev)
            work = torch.empty(int(lib.rc_sequence_loss_work_bytes(n)), dtype=torch.uint8, device=dev)
            check(lib.rc_sequence_loss_fwd(ptrs, wts, n, bs, flow_gt.data_ptr(), valid.data_ptr(), float(max_flow),
                                           results.data_ptr(), work.data_ptr(), N, H * W, _stream(dev)),
                  "rc_sequence_loss_fwd")
        ctx.views, ctx.bs, ctx.weights, ctx.max_flow = [v[0] for v in views], bs, weights, float(max_flow)
        ctx.save_for_backward(flow_gt, valid, results)
        ctx.mark_non_differentiable(results)
        return results[n + 5].float(), results

    @staticmethod
    def backward(ctx, grad_loss, _grad_results):
        flow_gt, valid, results = ctx.saved_tensors
        n = len(ctx.views)
        N, _, H, W = flow_gt.shape
        dev = flow_gt.device
        grads = [torch.empty((N, 1, H, W), dtype=torch.float32, device=dev) for _ in range(n)]
        ptrs = (ctypes.c_void_p * n)(*[v.data_ptr() for v in ctx.views])
        gptrs = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
        wts = (ctypes.c_float * n)(*ctx.weights)
        gl = grad_loss.float().contiguous()
        with _on_device(dev):
            check(lib.rc_sequence_loss_bwd(ptrs, gptrs, wts, n, ctx.bs, H * W, flow_gt.data_ptr(), valid.data_ptr(),
                                           ctx.max_flow, results.data_ptr(), gl.data_ptr(), N, H * W, _stream(dev)),
                  "rc_sequence_loss_bwd")
        return (None, None, None, None, *grads)


def sequence_loss_hip(flow_preds, flow_gt, valid, loss_gamma=0.9, max_flow=700):
    """sequence_loss of the reference (train_stereo.py:35-70): `(flow_loss, metrics)` from the list of
    per-iteration predictions `[N,1,H,W]`, the ground truth `[N,1,H,W]` and the validity map `[N,H,W]`.
    One pass over all predictions on the GPU and ONE host synchronisation (for the metrics' `.item()`s, which
    the reference also pays, on top of one per prediction for its boolean-mask gathers).  The reference's
    asserts (no nan/inf in the predictions, train_stereo.py:51-52) are checked on the same synchronisation."""
    n_predictions = len(flow_preds)
    assert n_predictions >= 1
    if n_predictions > _lib.RC_MAX_FUSED_LOOKUPS:
        raise ValueError(f"at most {_lib.RC_MAX_FUSED_LOOKUPS} predictions per call")
    if not (flow_gt.is_cuda and valid.is_cuda and all(p.is_cuda for p in flow_preds)):
        raise RuntimeError("sequence_loss_hip needs GPU tensors (there is no CPU path)")
    if flow_gt.dim() != 4 or flow_gt.shape[1] != 1:
        raise ValueError("flow_gt must be [N,1,H,W] (stereo: one flow channel)")
    N, _, H, W = flow_gt.shape
    assert valid.unsqueeze(1).shape == flow_gt.shape, [valid.shape, flow_gt.shape]       # train_stereo.py:47
    for p in flow_preds:
        assert p.shape == flow_gt.shape, [p.shape, flow_gt.shape]                        # train_stereo.py:57
    # We adjust the loss_gamma so it is consistent for any number of iterations (train_stereo.py:53-55)
    adjusted_loss_gamma = loss_gamma ** (15 / (n_predictions - 1))
    weights = [adjusted_loss_gamma ** (n_predictions - i - 1) for i in range(n_predictions)]
    flow_loss, results = _SeqLossFn.apply(flow_gt.detach().float().contiguous(), valid.detach().float().contiguous(),
                                          max_flow, weights, *flow_preds)
    r = results.tolist()                       # the one synchronisation
    n = n_predictions
    assert r[n + 6] == 0, "nan or inf in the flow predictions"                           # train_stereo.py:51-52
    cnt = r[n]
    metrics = {
        'epe': r[n + 1] / cnt if cnt else float("nan"),
        '1px': r[n + 2] / cnt if cnt else float("nan"),
        '3px': r[n + 3] / cnt if cnt else float("nan"),
        '5px': r[n + 4] / cnt if cnt else float("nan"),
    }
    return flow_loss, metrics
