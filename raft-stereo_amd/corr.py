"""Correlation blocks with the reference's interface, backed by libraftcorr_hip.so.

    block = CorrBlock1DHip(fmap1, fmap2, num_levels=4, radius=4)     # core/corr.py:111
    corr  = block(coords)            # [B, num_levels*(2*radius+1), H, W] fp32   core/corr.py:127

CorrBlock1DHip    -> `reg_hip`  (replaces CorrBlock1D, core/corr.py:110-156, and
                                  CorrBlockFast1D + CorrSampler, core/corr.py:17-61)
AltCorrBlock1DHip -> `alt_hip`  (replaces PytorchAlternateCorrBlock1D, core/corr.py:64-107)
CorrSamplerHip mirrors the autograd wrapper of core/corr.py:17-29 on top of the
`corr_sampler` twin (row-major single-level volumes) for callers of that lower-level interface.

Torch only owns memory, streams and the autograd graph here; all arithmetic is in the HIP
library.  There is no CPU or eager fallback: tensors must live on a GPU.
"""
import ctypes
import weakref

import torch

from raft_stereo_amd import _lib, sampler
from raft_stereo_amd._lib import lib, check, Layout, AltLayout

_DT = {torch.float32: _lib.RC_F32, torch.float16: _lib.RC_F16, torch.bfloat16: _lib.RC_BF16}
COORD_MODES = {"exact": _lib.RC_COORD_EXACT, "gridsample": _lib.RC_COORD_GRIDSAMPLE}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # what compiled torch code calls: ~0.1 us


def _stream(dev):
    """hipStream_t of torch's current stream on `dev` (the library launches where torch would)."""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` without its cost when `dev` is already current (the per-iteration
    lookup of a small image is bound by host time, not by the kernel)."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        self.idx = dev.index if dev.index is not None else torch.cuda.current_device()

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.prev != self.idx:
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch.cuda.set_device(self.prev)
        return False


def _check_fmaps(fmap1, fmap2, num_levels, radius):
    if fmap1.dim() != 4 or fmap2.dim() != 4:
        raise ValueError("fmap1/fmap2 must be [B, D, H, W]")
    if not (fmap1.is_cuda and fmap2.is_cuda):
        raise RuntimeError("raft_stereo_amd correlation blocks need GPU tensors (there is no CPU path)")
    if fmap1.device != fmap2.device:
        raise RuntimeError("fmap1 and fmap2 live on different devices")
    if fmap1.dtype != fmap2.dtype or fmap1.dtype not in _DT:
        raise TypeError(f"unsupported feature dtypes {fmap1.dtype}/{fmap2.dtype}")
    B, D, H, W1 = fmap1.shape
    B2, D2, H2, W2 = fmap2.shape
    if (B, D, H) != (B2, D2, H2):
        raise ValueError(f"fmap shapes disagree: {tuple(fmap1.shape)} vs {tuple(fmap2.shape)}")
    if not (1 <= num_levels <= _lib.RC_MAX_LEVELS):
        raise ValueError(f"num_levels must be in 1..{_lib.RC_MAX_LEVELS}")
    if not (1 <= radius <= _lib.RC_MAX_RADIUS):
        raise ValueError(f"radius must be in 1..{_lib.RC_MAX_RADIUS}")
    if (W2 >> num_levels) < 1:
        # the reference pools num_levels times (core/corr.py:123-125) and avg_pool2d raises
        raise RuntimeError(f"Output size is too small: W2={W2} cannot be pooled {num_levels} times")
    return B, D, H, W1, W2


def _check_coords(coords, B, H, W1, device):
    if coords.dim() != 4 or coords.shape[0] != B or coords.shape[2] != H or coords.shape[3] != W1 \
            or coords.shape[1] < 1:
        raise ValueError(f"coords must be [B={B}, >=1, H={H}, W={W1}], got {tuple(coords.shape)}")
    if coords.device != device:
        raise RuntimeError("coords live on a different device than the feature maps")
    if coords.dtype is torch.float32 and coords.is_contiguous():
        return coords.detach() if coords.requires_grad else coords
    return coords.detach().float().contiguous()


class _BuildFn(torch.autograd.Function):
    """Builds the pyramid into block.volume and returns a 1-element token that ties every
    lookup to the build in the autograd graph.  Its backward runs once, after all lookups'
    backwards have queued their (coords, grad_out) in block._pending."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, block):
        block._build(fmap1, fmap2)
        # a weak reference: block -> token -> this node -> ctx -> block would be a cycle, and the pyramid would then
        # wait for Python's cyclic collector instead of being freed with the block.  Every lookup node holds the
        # block strongly (it needs the pyramid), so the block is alive here whenever a lookup took part in the loss.
        ctx.block = weakref.ref(block)
        ctx.save_for_backward(fmap1, fmap2)
        ctx.set_materialize_grads(False)    # the token's gradient carries nothing: lookups return None for it
        return torch.zeros(1, dtype=torch.float32, device=fmap1.device)

    @staticmethod
    def backward(ctx, _grad_token):
        fmap1, fmap2 = ctx.saved_tensors
        block = ctx.block()
        pending = []
        if block is not None:
            pending, block._pending = block._pending, []
        if not pending:           # no lookup took part in the loss
            return torch.zeros_like(fmap1), torch.zeros_like(fmap2), None
        df1, df2 = torch.empty_like(fmap1), torch.empty_like(fmap2)
        lay = block.grad_layout
        dev = fmap1.device
        nmax = _lib.RC_MAX_FUSED_LOOKUPS
        with torch.cuda.device(dev):
            st = _stream(dev)
            if len(pending) <= nmax or block.volume is None:
                # one fused adjoint of all lookups -> dense gradient of level 0 -> two contractions.
                # (alt_hip has no pyramid to mirror, so more than nmax lookups go in chunks whose
                # feature gradients are summed in fp32.)
                ld = int(lib.rc_dense_grad_ld(lay))
                dense = torch.empty(block.B * block.H * block.W1 * ld, dtype=fmap1.dtype, device=dev)
                chunks = [pending[k:k + nmax] for k in range(0, len(pending), nmax)]
                sum1 = sum2 = None
                for part in chunks:
                    n = len(part)
                    if len({c.shape[1] for c, _ in part}) != 1:     # one channel count per fused call
                        part = [(c[:, :1].contiguous(), g) for c, g in part]
                    cl = (ctypes.c_void_p * n)(*[c.data_ptr() for c, _ in part])
                    gl = (ctypes.c_void_p * n)(*[g.data_ptr() for _, g in part])
                    check(lib.rc_lookup_bwd_dense(cl, gl, n, part[0][0].shape[1], dense.data_ptr(), lay, block.D,
                                                  block.in_dtype, block.radius, block.coord_mode, st),
                          "rc_lookup_bwd_dense")
                    check(lib.rc_build_bwd_dense(dense.data_ptr(), lay, fmap1.data_ptr(), fmap2.data_ptr(),
                                                 df1.data_ptr(), df2.data_ptr(), block.D, block.in_dtype, st),
                          "rc_build_bwd_dense")
                    if len(chunks) > 1:
                        sum1 = df1.to(torch.float32, copy=True) if sum1 is None else sum1.add_(df1)
                        sum2 = df2.to(torch.float32, copy=True) if sum2 is None else sum2.add_(df2)
                if len(chunks) > 1:
                    df1, df2 = sum1.to(fmap1.dtype), sum2.to(fmap2.dtype)
            else:
                # more lookups than one fused call takes: accumulate into a float gradient pyramid
                gvol = torch.zeros(lay.total, dtype=torch.float32, device=dev)
                for coords, grad_out in pending:
                    check(lib.rc_lookup_bwd(coords.data_ptr(), coords.shape[1], grad_out.data_ptr(), gvol.data_ptr(),
                                            lay, block.radius, block.coord_mode, st), "rc_lookup_bwd")
                nbytes = lib.rc_build_bwd_work_bytes(lay, block.in_dtype)
                if nbytes < 0:
                    raise _lib.RaftCorrError("rc_build_bwd_work_bytes rejected the layout")
                work = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
                check(lib.rc_build_bwd(gvol.data_ptr(), lay, fmap1.data_ptr(), fmap2.data_ptr(), df1.data_ptr(),
                                       df2.data_ptr(), block.D, block.in_dtype, work.data_ptr(), st),
                      "rc_build_bwd")
        return df1, df2, None


class _LookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, coords, block):
        ctx.block = block
        ctx.save_for_backward(coords)    # saved, not stashed: an in-place edit of coords before backward raises
        return block._lookup(coords)

    @staticmethod
    def backward(ctx, grad_out):
        # nothing is computed here: the build's backward, which autograd runs after every lookup's,
        # takes all (coords, grad_out) pairs in one fused call
        if grad_out.dtype is not torch.float32 or not grad_out.is_contiguous():
            grad_out = grad_out.float().contiguous()                              # core/corr.py:27
        coords, = ctx.saved_tensors
        ctx.block._pending.append((coords, grad_out))
        return None, None, None     # autograd still runs the build's backward once all lookups have


class _LookupConvc1Fn(torch.autograd.Function):
    """relu(convc1(lookup(coords))) as one differentiable node.  Nothing but the 64-channel result is kept
    for the backward: it recomputes the lookup, runs rc_convc1_bwd (tap, weight and bias gradients) and
    queues the tap gradients for the build's fused adjoint like a plain lookup does."""

    @staticmethod
    def forward(ctx, token, coords, weight, bias, block, relu):
        act = block._lookup_convc1(coords, weight, bias, relu)
        ctx.block, ctx.relu, ctx.has_token = block, relu, token is not None
        ctx.save_for_backward(act, weight, bias, coords)
        return act

    @staticmethod
    def backward(ctx, grad_act):
        act, weight, bias, coords = ctx.saved_tensors
        block = ctx.block
        K = block.num_levels * (2 * block.radius + 1)
        dev = block.device
        corr = block._lookup(coords, fp32=True)               # recomputed: the forward never stored it
        g = grad_act.to(act.dtype).contiguous()
        w32 = weight.detach().reshape(_lib.RC_CONVC1_OUT, K).float().contiguous()
        with torch.cuda.device(dev):
            grad_corr = torch.empty_like(corr)
            gw = torch.empty((_lib.RC_CONVC1_OUT, K), dtype=torch.float32, device=dev)
            gb = torch.empty((_lib.RC_CONVC1_OUT,), dtype=torch.float32, device=dev)
            work = torch.empty(int(lib.rc_convc1_bwd_work_bytes()), dtype=torch.uint8, device=dev)
            check(lib.rc_convc1_bwd(corr.data_ptr(), act.data_ptr(), g.data_ptr(), w32.data_ptr(),
                                    grad_corr.data_ptr(), gw.data_ptr(), gb.data_ptr(), work.data_ptr(), block.B, K,
                                    block.H * block.W1, block.vol_dtype, int(ctx.relu), _stream(dev)),
                  "rc_convc1_bwd")
        if ctx.has_token:
            block._pending.append((coords, grad_corr))        # taken by the build's backward
        return None, None, gw.view(weight.shape).to(weight.dtype), gb.view(bias.shape).to(bias.dtype), None, None


class CorrBlock1DHip:
    """`reg_hip`: materialised all-pairs 1-D correlation pyramid + per-iteration lookup.

    fp32 feature maps give an fp32 volume built with the exact-f32 MFMA and grid_sample-
    compatible tap coordinates (the reg path, core/corr.py:110-156); fp16/bf16 feature maps (the
    autocast case of reg_cuda, core/raft_stereo.py:96-97) give a 16-bit volume and the
    floor+lerp coordinates of sampler_kernel.cu.  `coord_mode` overrides that choice.
    Differentiable w.r.t. fmap1/fmap2 only (coords are detached, core/raft_stereo.py:109,
    core/corr.py:29).
    """

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, coord_mode=None, volume_dtype=None, out_dtype=None):
        B, D, H, W1, W2 = _check_fmaps(fmap1, fmap2, num_levels, radius)
        self.num_levels, self.radius = num_levels, radius
        self.B, self.D, self.H, self.W1, self.W2 = B, D, H, W1, W2
        self.device = fmap1.device
        self.in_dtype = _DT[fmap1.dtype]
        self.volume_torch_dtype = volume_dtype or fmap1.dtype
        # out_dtype: fp32 (default: what CorrBlock1D returns, core/corr.py:145-146) or the volume's own 16-bit
        # type (what CorrBlockFast1D returns under mixed precision: CorrSampler allocates volume.options(),
        # core/corr.py:44-51, sampler_kernel.cu:122-124) -- half the bytes the lookup writes
        if out_dtype not in (None, torch.float32, self.volume_torch_dtype):
            raise TypeError("out_dtype must be float32 or the volume's dtype")
        self.out_torch_dtype = out_dtype or torch.float32
        if self.volume_torch_dtype not in (fmap1.dtype, torch.float32):
            raise TypeError("volume_dtype must be the feature dtype or float32")
        self.vol_dtype = _DT[self.volume_torch_dtype]
        if coord_mode is None:
            coord_mode = "gridsample" if fmap1.dtype == torch.float32 else "exact"
        self.coord_mode = COORD_MODES[coord_mode]
        self.layout = Layout()          # of the volume
        self.grad_layout = Layout()     # of the float gradient pyramid
        check(lib.rc_pyramid_layout(B, H, W1, W2, num_levels, self.vol_dtype, self.layout), "rc_pyramid_layout")
        check(lib.rc_grad_pyramid_layout(B, H, W1, W2, num_levels, self.grad_layout), "rc_grad_pyramid_layout")
        self.volume = None
        self._pending = []              # (coords, grad_out) of lookups whose backward has run
        fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
        if torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad):
            self._token = _BuildFn.apply(fmap1, fmap2, self)
        else:
            self._token = None
            self._build(fmap1.detach(), fmap2.detach())

    # -- raw calls into the library
    def _build(self, fmap1, fmap2):
        with torch.cuda.device(self.device):
            self.volume = torch.empty(self.layout.total, dtype=self.volume_torch_dtype, device=self.device)
            check(lib.rc_build_pyramid(fmap1.data_ptr(), fmap2.data_ptr(), self.volume.data_ptr(), self.layout,
                                       self.D, self.in_dtype, _stream(self.device)),
                  "rc_build_pyramid")
        # per-call constants of the lookup, converted once
        self._vol_ptr = self.volume.data_ptr()
        self._lay_ref = ctypes.byref(self.layout)
        self._out_shape = (self.B, self.num_levels * (2 * self.radius + 1), self.H, self.W1)

    def _lookup(self, coords, fp32=False):
        """fp32=True: the fp32 taps whatever out_dtype says (the fused layer's backward recomputes them)."""
        vt = self.out_torch_dtype is not torch.float32 and not fp32
        with _on_device(self.device):
            out = torch.empty(self._out_shape, dtype=self.out_torch_dtype if vt else torch.float32, device=self.device)
            fn = lib.rc_lookup_fwd_vt if vt else lib.rc_lookup_fwd
            rc = fn(self._vol_ptr, self._lay_ref, coords.data_ptr(), coords.shape[1],
                    out.data_ptr(), self.radius, self.coord_mode, _stream(self.device))
            if rc:
                check(rc, "rc_lookup_fwd_vt" if vt else "rc_lookup_fwd")
        return out

    def __call__(self, coords):
        coords = _check_coords(coords, self.B, self.H, self.W1, self.device)
        if self._token is not None and torch.is_grad_enabled():
            return _LookupFn.apply(self._token, coords, self)
        return self._lookup(coords)

    def update_and_lookup(self, coords1, delta_flow):
        """One GRU iteration's coordinate glue + lookup in a single kernel (inference):
        `delta_flow[:,1] = 0; coords1 = coords1 + delta_flow` (core/raft_stereo.py:120-123) of the
        previous iteration, then `corr = corr_fn(coords1); flow = coords1 - coords0` (:110-111).
        Returns `(coords1_new, flow, corr)`; coords1 `[B,2,H,W]` fp32, delta_flow `[B,2,H,W]`."""
        if torch.is_grad_enabled() and (self._token is not None or delta_flow.requires_grad):
            raise RuntimeError("update_and_lookup is forward-only: call it under torch.no_grad()")
        coords1 = _check_coords(coords1, self.B, self.H, self.W1, self.device)
        if tuple(delta_flow.shape) != (self.B, 2, self.H, self.W1) or delta_flow.device != self.device:
            raise ValueError(f"delta_flow must be [B={self.B}, 2, H={self.H}, W={self.W1}] on {self.device}")
        delta = delta_flow.detach().float().contiguous()
        with _on_device(self.device):
            new = torch.empty((self.B, 2, self.H, self.W1), dtype=torch.float32, device=self.device)
            flow = torch.empty_like(new)
            out = torch.empty(self._out_shape, dtype=torch.float32, device=self.device)
            check(lib.rc_lookup_fwd_update(self._vol_ptr, self._lay_ref, coords1.data_ptr(), coords1.shape[1],
                                           delta.data_ptr(), new.data_ptr(), flow.data_ptr(), out.data_ptr(),
                                           self.radius, self.coord_mode, _stream(self.device)),
                  "rc_lookup_fwd_update")
        return new, flow, out

    def lookup_convc1(self, coords, weight, bias, relu=True):
        """`F.relu(convc1(block(coords)))` (core/update.py:71,78 on the result of
        core/raft_stereo.py:110) in one kernel: the L*(2r+1)-channel tensor is never written.

        weight: convc1.weight `[64, L*(2r+1), 1, 1]` (or `[64, L*(2r+1)]`), bias `[64]`.  Returns
        `[B, 64, H, W]` in the pyramid's dtype: fp16/bf16 under mixed precision (what autocast makes
        of the convolution), fp32 otherwise.  Differentiable w.r.t. the feature maps, weight and bias
        (SURVEY.md section 8 f, rank 1): the backward recomputes the lookup instead of keeping it."""
        K = self.num_levels * (2 * self.radius + 1)
        if weight.numel() != _lib.RC_CONVC1_OUT * K or weight.shape[0] != _lib.RC_CONVC1_OUT \
                or bias.numel() != _lib.RC_CONVC1_OUT:
            raise ValueError(f"convc1 weight must be [{_lib.RC_CONVC1_OUT}, {K}(,1,1)] and bias "
                             f"[{_lib.RC_CONVC1_OUT}], got {tuple(weight.shape)} and {tuple(bias.shape)}")
        if self.radius > _lib.RC_CONVC1_MAX_RADIUS:
            raise ValueError(f"lookup_convc1 supports radius <= {_lib.RC_CONVC1_MAX_RADIUS}")
        if weight.device != self.device or bias.device != self.device:
            raise RuntimeError("convc1 parameters live on a different device than the feature maps")
        coords = _check_coords(coords, self.B, self.H, self.W1, self.device)
        if torch.is_grad_enabled() and (self._token is not None or weight.requires_grad or bias.requires_grad):
            return _LookupConvc1Fn.apply(self._token, coords, weight, bias, self, bool(relu))
        return self._lookup_convc1(coords, weight, bias, relu)

    def update_lookup_convc1(self, coords1, delta_flow, weight, bias, relu=True):
        """A whole GRU iteration's glue and the fused lookup in ONE kernel (inference):
        `delta_flow[:,1] = 0; coords1 = coords1 + delta_flow` of the previous iteration
        (core/raft_stereo.py:120-123; `delta_flow=None` on the first iteration: nothing is added), then
        `flow = coords1 - coords0` (:111) and `relu(convc1(corr_fn(coords1)))` (:110, core/update.py:78).
        delta_flow `[B,2,H,W]` in fp32 / fp16 / bf16 as the flow head returns it (promoted to fp32 like torch's
        `+`).  Returns `(coords1_new, flow, activation)`."""
        if torch.is_grad_enabled() and (self._token is not None or weight.requires_grad or bias.requires_grad
                                        or (delta_flow is not None and delta_flow.requires_grad)):
            raise RuntimeError("update_lookup_convc1 is forward-only: call it under torch.no_grad()")
        if self.radius > _lib.RC_CONVC1_MAX_RADIUS:
            raise ValueError(f"lookup_convc1 supports radius <= {_lib.RC_CONVC1_MAX_RADIUS}")
        coords1 = _check_coords(coords1, self.B, self.H, self.W1, self.device)
        dptr, ddt = None, _lib.RC_F32
        if delta_flow is not None:
            if tuple(delta_flow.shape) != (self.B, 2, self.H, self.W1) or delta_flow.device != self.device:
                raise ValueError(f"delta_flow must be [B={self.B}, 2, H={self.H}, W={self.W1}] on {self.device}")
            delta = delta_flow.detach()
            if delta.dtype not in _DT:
                delta = delta.float()
            delta = delta.contiguous()
            dptr, ddt = delta.data_ptr(), _DT[delta.dtype]
        with _on_device(self.device):
            st = _stream(self.device)
            self._pack_convc1(weight, bias, st)
            new = torch.empty((self.B, 2, self.H, self.W1), dtype=torch.float32, device=self.device)
            flow = torch.empty_like(new)
            out = torch.empty((self.B, _lib.RC_CONVC1_OUT, self.H, self.W1), dtype=self.volume_torch_dtype,
                              device=self.device)
            check(lib.rc_lookup_convc1_fwd_update(self._vol_ptr, self._lay_ref, coords1.data_ptr(), coords1.shape[1],
                                                  dptr, ddt, new.data_ptr(), flow.data_ptr(),
                                                  self._convc1_packed.data_ptr(), self._convc1_bias.data_ptr(),
                                                  out.data_ptr(), self.radius, self.coord_mode, int(bool(relu)), st),
                  "rc_lookup_convc1_fwd_update")
        return new, flow, out

    def _pack_convc1(self, weight, bias, st):
        """convc1's weight in MFMA fragment order, cached while the parameters stand still."""
        K = self.num_levels * (2 * self.radius + 1)
        if weight.numel() != _lib.RC_CONVC1_OUT * K or weight.shape[0] != _lib.RC_CONVC1_OUT \
                or bias.numel() != _lib.RC_CONVC1_OUT:
            raise ValueError(f"convc1 weight must be [{_lib.RC_CONVC1_OUT}, {K}(,1,1)] and bias "
                             f"[{_lib.RC_CONVC1_OUT}], got {tuple(weight.shape)} and {tuple(bias.shape)}")
        if weight.device != self.device or bias.device != self.device:
            raise RuntimeError("convc1 parameters live on a different device than the feature maps")
        try:
            key = (weight.data_ptr(), weight._version, bias.data_ptr(), bias._version)
        except RuntimeError:      # inference-mode tensors have no version counter: re-pack on every call
            key = None
        if key is None or getattr(self, "_convc1_key", None) != key:
            w32 = weight.detach().reshape(_lib.RC_CONVC1_OUT, K).float().contiguous()
            self._convc1_packed = torch.empty(_lib.RC_CONVC1_OUT * 64, dtype=self.volume_torch_dtype,
                                              device=self.device)
            check(lib.rc_convc1_pack(w32.data_ptr(), self.num_levels, self.radius, self.vol_dtype,
                                     self._convc1_packed.data_ptr(), st), "rc_convc1_pack")
            self._convc1_bias = bias.detach().float().contiguous().clone()
            self._convc1_key = key

    def _lookup_convc1(self, coords, weight, bias, relu):
        K = self.num_levels * (2 * self.radius + 1)
        with torch.cuda.device(self.device):
            st = _stream(self.device)
            self._pack_convc1(weight, bias, st)
            out = torch.empty((self.B, _lib.RC_CONVC1_OUT, self.H, self.W1), dtype=self.volume_torch_dtype,
                              device=self.device)
            check(lib.rc_lookup_convc1_fwd(self._vol_ptr, self._lay_ref, coords.data_ptr(), coords.shape[1],
                                           self._convc1_packed.data_ptr(), self._convc1_bias.data_ptr(),
                                           out.data_ptr(), self.radius, self.coord_mode, int(bool(relu)), st),
                  "rc_lookup_convc1_fwd")
        return out

    # -- test / debug helper: the pyramid in the reference's dense layout
    def dense_level(self, i, volume=None, layout=None):
        """Level i as a dense [B,H,W1,W2>>i] tensor (un-skews the private layout with torch
        indexing; for tests only).  A level the layout does not store is the pairwise mean of the
        level below (fp32 mean of the stored values, rounded to the storage type) -- what the lookup
        kernels derive in registers."""
        lay = layout or self.layout
        vol = self.volume if volume is None else volume
        if not lay.stored[i]:
            below = self.dense_level(i - 1, volume=vol, layout=lay)
            n = lay.width[i]
            return ((below[..., 0:2 * n:2].float() + below[..., 1:2 * n:2].float()) * 0.5).to(below.dtype)
        Wi, WiP, pitch, rb = lay.width[i], lay.widthp[i], lay.pitch, lay.rb
        nb = WiP // rb
        n = self.B * self.H * WiP * pitch
        lvl = vol[lay.offset[i]:lay.offset[i] + n].view(self.B, self.H, nb, pitch, rb)
        w1 = torch.arange(self.W1, device=self.device).view(-1, 1)
        w2 = torch.arange(Wi, device=self.device).view(1, -1)
        sb = (w2 // rb - (w1 >> i) // rb + nb - _lib.RC_WRAP_ROWS // rb) % nb
        return lvl[:, :, sb, w1.expand_as(sb), (w2 % rb).expand_as(sb)]


class AltCorrBlock1DHip:
    """`alt_hip`: same outputs as CorrBlock1DHip without ever building the volume
    (PytorchAlternateCorrBlock1D, core/corr.py:64-107).  Like the reg path it samples along x
    only: coords channel 1 is the pixel's own row in RAFT-Stereo (core/raft_stereo.py:120).
    Differentiable w.r.t. fmap1/fmap2 like the reference's (which relies on autograd through
    grid_sample): the backward is the fused adjoint + two contractions of the reg path."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, coord_mode=None):
        B, D, H, W1, W2 = _check_fmaps(fmap1, fmap2, num_levels, radius)
        if radius > _lib.RC_ALT_MAX_RADIUS:
            raise ValueError(f"alt_hip supports radius <= {_lib.RC_ALT_MAX_RADIUS}")
        self.num_levels, self.radius = num_levels, radius
        self.B, self.D, self.H, self.W1, self.W2 = B, D, H, W1, W2
        self.device = fmap1.device
        self.in_dtype = _DT[fmap1.dtype]
        if coord_mode is None:
            coord_mode = "gridsample" if fmap1.dtype == torch.float32 else "exact"
        self.coord_mode = COORD_MODES[coord_mode]
        self.layout = AltLayout()
        check(lib.rc_alt_layout_init(B, D, H, W1, W2, num_levels, self.layout), "rc_alt_layout_init")
        # the adjoint of the alt lookups is the adjoint of the reg lookups (same function of the
        # feature maps): the backward needs only the geometry of a pyramid, never its memory
        self.grad_layout = Layout()
        check(lib.rc_grad_pyramid_layout(B, H, W1, W2, num_levels, self.grad_layout), "rc_grad_pyramid_layout")
        self.volume = None
        self._pending = []
        self._out_shape = (B, num_levels * (2 * radius + 1), H, W1)
        fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
        if torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad):
            self._token = _BuildFn.apply(fmap1, fmap2, self)
        else:
            self._token = None
            self._build(fmap1.detach(), fmap2.detach())

    def _build(self, fmap1, fmap2):
        with torch.cuda.device(self.device):
            self.f1t = torch.empty(self.layout.f1_elems, dtype=fmap1.dtype, device=self.device)
            self.f2t = torch.empty(self.layout.f2_elems, dtype=fmap1.dtype, device=self.device)
            check(lib.rc_alt_prepare(fmap1.data_ptr(), fmap2.data_ptr(), self.f1t.data_ptr(), self.f2t.data_ptr(),
                                     self.layout, self.in_dtype, _stream(self.device)), "rc_alt_prepare")

    def _lookup(self, coords):
        with _on_device(self.device):
            out = torch.empty(self._out_shape, dtype=torch.float32, device=self.device)
            check(lib.rc_alt_fwd(self.f1t.data_ptr(), self.f2t.data_ptr(), self.layout, self.in_dtype,
                                 coords.data_ptr(), coords.shape[1], out.data_ptr(), self.radius,
                                 self.coord_mode, _stream(self.device)), "rc_alt_fwd")
        return out

    def __call__(self, coords):
        coords = _check_coords(coords, self.B, self.H, self.W1, self.device)
        if self._token is not None and torch.is_grad_enabled():
            return _LookupFn.apply(self._token, coords, self)
        return self._lookup(coords)


class CorrSamplerHip(torch.autograd.Function):
    """core/corr.py:17-29 on the HIP `corr_sampler` twin."""

    @staticmethod
    def forward(ctx, volume, coords, radius):
        ctx.save_for_backward(volume, coords)
        ctx.radius = radius
        corr, = sampler.forward(volume, coords, radius)
        return corr

    @staticmethod
    def backward(ctx, grad_output):
        volume, coords = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_volume, = sampler.backward(volume, coords, grad_output, ctx.radius)
        return grad_volume, None, None
