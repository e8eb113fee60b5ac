"""GPU parity tests for the pointwise work of a GRU iteration (SURVEY.md section 8 f, ranks 3 and 4):
ConvGRU gate arithmetic and convex upsampling, through the C-ABI library, against the golden vectors
captured from the reference's ConvGRU / upsample_flow and against the CPU oracle.

Bars (metric max|a-b| / max|ref|): fp32 1e-5 for the library's own arithmetic (checked on the
reference's recorded convolution outputs); 1e-4 where torch's GPU convolutions are part of the
comparison (ConvGRUHip end to end; the convolutions are not this package's); fp16 2e-3, bf16 1.6e-2
against the oracle evaluated with the same per-operation rounding.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import update_oracle as U
from conftest import GRU_CASES, LOSS_CASES, UPSAMPLE_CASES

pytestmark = pytest.mark.gpu

TOL32 = 1e-5
TOL_CONV = 1e-4
TOL16 = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}


@pytest.fixture(scope="module")
def R():
    import raft_stereo_amd
    assert torch.cuda.is_available(), "these tests need a GPU"
    return raft_stereo_amd


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def rel(a, b):
    return U.rel_to_max(a.detach().float().cpu().numpy() if torch.is_tensor(a) else a, b)


def _round(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).float().numpy()


def _rounder(dtype):
    return lambda x: _round(np.asarray(x, dtype=np.float32), dtype)


def _ctx_views(g, dtype=None):
    """cz, cr, cq as the reference hands them over: split views of one tensor (core/raft_stereo.py:88)."""
    ctx = dev(g["ctx"], dtype)
    return ctx, ctx.split(split_size=int(g["C"]), dim=1)


# ------------------------------------------------------------------ GRU kernels on the reference's conv outputs
@pytest.mark.parametrize("case", GRU_CASES)
def test_gru_kernels_fp32_match_reference(R, golden, case):
    g = golden(case)
    C = int(g["C"])
    _, (cz, cr, cq) = _ctx_views(g)
    assert not cz.is_contiguous() or g["ctx"].shape[0] == 1
    h = dev(g["h"])
    out = R.gru_update(dev(g["zc"]), cz, dev(g["qc"]), cq, h)
    assert rel(out, g["h_out"]) < TOL32
    rh = R.gru_gate_r(dev(g["rc"]), cr, h)
    assert rel(rh, U.gru_gate_r(g["rc"], g["ctx"][:, C:2 * C], g["h"])) < TOL32
    # the same product written into the first C channels of the next convolution's input buffer
    xs = [dev(g[f"x{i}"]) for i in range(int(g["n_x"]))]
    buf = torch.cat([torch.full_like(h, 7.0)] + xs, dim=1)
    keep = buf[:, C:].clone()
    R.gru_gate_r(dev(g["rc"]), cr, h, out=buf[:, :C])
    assert torch.equal(buf[:, :C], rh) and torch.equal(buf[:, C:], keep)
    qc = torch.nn.functional.conv2d(buf, dev(g["w_convq.weight"]), dev(g["w_convq.bias"]), padding=1)
    assert rel(qc, g["qc"]) < TOL_CONV


@pytest.mark.parametrize("case", GRU_CASES)
def test_gru_kernels_backward_fp32_match_reference_autograd(R, golden, case):
    g = golden(case)
    C = int(g["C"])
    ctx = dev(g["ctx"]).requires_grad_(True)
    cz, cr, cq = ctx.split(split_size=C, dim=1)
    zc, rc, qc, h = (dev(g[k]).requires_grad_(True) for k in ("zc", "rc", "qc", "h"))
    R.gru_update(zc, cz, qc, cq, h).backward(dev(g["grad_out"]))
    assert rel(zc.grad, g["d_zc"]) < TOL32
    assert rel(qc.grad, g["d_qc"]) < TOL32
    assert rel(ctx.grad[:, :C], g["d_ctx"][:, :C]) < TOL32
    assert rel(ctx.grad[:, 2 * C:], g["d_ctx"][:, 2 * C:]) < TOL32
    _, _, dh_u = U.gru_update_bwd(g["zc"], g["ctx"][:, :C], g["qc"], g["ctx"][:, 2 * C:], g["h"], g["grad_out"])
    assert rel(h.grad, dh_u) < TOL32
    # gate adjoint: d(r*h) is the input gradient of the reference's convq at its first C channels
    d_rh = torch.nn.grad.conv2d_input((h.shape[0], g["w_convq.weight"].shape[1]) + tuple(h.shape[2:]),
                                      torch.from_numpy(g["w_convq.weight"]), torch.from_numpy(g["d_qc"]),
                                      padding=1)[:, :C].contiguous()
    ctx.grad = None
    h.grad = None
    R.gru_gate_r(rc, cr, h).backward(d_rh.cuda())
    assert rel(rc.grad, g["d_rc"]) < TOL32
    assert rel(ctx.grad[:, C:2 * C], g["d_ctx"][:, C:2 * C]) < TOL32
    _, dh_r = U.gru_gate_r_bwd(g["rc"], g["ctx"][:, C:2 * C], g["h"], d_rh.numpy())
    assert rel(h.grad, dh_r) < TOL32


def test_gate_r_bwd_accumulates_into_grad_h(R, golden):
    """rc_gru_gate_r_bwd(accumulate_grad_h=1) adds to what grad_h holds (the C-ABI route for callers that
    chain the two adjoints without autograd)."""
    from raft_stereo_amd._lib import lib, check, RC_F32
    g = golden(GRU_CASES[0])
    C = int(g["C"])
    rc, h = dev(g["rc"]), dev(g["h"])
    cr = dev(g["ctx"][:, C:2 * C])
    grh = torch.randn_like(h)
    N, per = h.shape[0], h[0].numel()
    d_pre, base = torch.empty_like(h), torch.randn_like(h)
    acc = base.clone()
    check(lib.rc_gru_gate_r_bwd(rc.data_ptr(), cr.data_ptr(), per, h.data_ptr(), grh.data_ptr(), per,
                                d_pre.data_ptr(), acc.data_ptr(), 1, N, per, RC_F32,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "rc_gru_gate_r_bwd")
    dr, dh = U.gru_gate_r_bwd(g["rc"], g["ctx"][:, C:2 * C], g["h"], grh.cpu().numpy())
    assert rel(d_pre, dr) < TOL32
    assert rel(acc - base, dh) < 1e-5


# ------------------------------------------------------------------ ConvGRUHip end to end (torch convolutions)
@pytest.mark.parametrize("case", GRU_CASES)
def test_convgru_module_matches_reference(R, golden, case):
    g = golden(case)
    C, nx = int(g["C"]), int(g["n_x"])
    xdim = sum(g[f"x{i}"].shape[1] for i in range(nx))
    gru = R.ConvGRUHip(C, xdim).cuda()
    gru.load_state_dict({k[2:]: dev(v) for k, v in g.items() if k.startswith("w_")})   # reference names
    ctx = dev(g["ctx"]).requires_grad_(True)
    h = dev(g["h"]).requires_grad_(True)
    xs = [dev(g[f"x{i}"]).requires_grad_(True) for i in range(nx)]
    out = gru(h, *ctx.split(split_size=C, dim=1), *xs)
    assert rel(out, g["h_out"]) < TOL_CONV
    out.backward(dev(g["grad_out"]))
    assert rel(h.grad, g["d_h"]) < TOL_CONV
    assert rel(ctx.grad, g["d_ctx"]) < TOL_CONV
    for i, x in enumerate(xs):
        assert rel(x.grad, g[f"d_x{i}"]) < TOL_CONV
    for k, p in gru.named_parameters():
        assert rel(p.grad, g["dw_" + k]) < TOL_CONV, k
    with torch.no_grad():          # the inference route writes r*h in place into its own [h, x] copy
        out2 = gru(h.detach(), *ctx.detach().split(split_size=C, dim=1), *[x.detach() for x in xs])
    assert rel(out2, g["h_out"]) < TOL_CONV
    assert rel(h, g["h"]) == 0.0


def _torch_convgru(gru, h, cz, cr, cq, *x_list):
    """The reference's ConvGRU.forward (core/update.py:23-32) spelled with torch ops on the same parameters."""
    x = torch.cat(x_list, dim=1)
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(gru.convz(hx) + cz)
    r = torch.sigmoid(gru.convr(hx) + cr)
    q = torch.tanh(gru.convq(torch.cat([r * h, x], dim=1)) + cq)
    return (1 - z) * h + z * q


@pytest.mark.parametrize("amp", [None, torch.float16, torch.bfloat16])
def test_convgru_module_under_autocast_at_model_size(R, amp):
    """gru08 of the default model at the KITTI 1/4 resolution (hidden 128, inputs 128 + 128), inference and
    training, with and without autocast, against the same torch ops.  Under autocast the motion features are
    fp32 (core/update.py:85 concatenates the fp32 flow) while h and the context are 16-bit."""
    torch.manual_seed(31)
    N, C, H, W = 2, 128, 96, 312
    gru = R.ConvGRUHip(C, 256).cuda()
    cdt = amp or torch.float32
    h = torch.tanh(torch.randn(N, C, H, W, device="cuda")).to(cdt)
    ctx = torch.randn(N, 3 * C, H, W, device="cuda").to(cdt)
    motion = torch.randn(N, 128, H, W, device="cuda")
    up = torch.randn(N, 128, H, W, device="cuda").to(cdt)
    tol = 2e-4 if amp is None else 2 * TOL16[amp]
    with torch.autocast("cuda", dtype=cdt, enabled=amp is not None):
        with torch.no_grad():
            mine = gru(h, *ctx.split(C, dim=1), motion, up)
            ref = _torch_convgru(gru, h, *ctx.split(C, dim=1), motion, up)
        assert mine.dtype == ref.dtype
        assert rel(mine, ref.float().cpu().numpy()) < tol
    # training: gradients of the parameters and of h (a fresh autocast region: the weight casts cached under
    # no_grad above carry no autograd history)
    with torch.autocast("cuda", dtype=cdt, enabled=amp is not None):
        grads = []
        gout = torch.randn(N, C, H, W, device="cuda")       # O(1) gradients: nothing underflows in fp16
        for fn in (gru, lambda *a: _torch_convgru(gru, *a)):
            gru.zero_grad()
            hh = h.clone().requires_grad_(True)
            out = fn(hh, *ctx.split(C, dim=1), motion, up)
            (out.float() * gout).sum().backward()
            grads.append([hh.grad.float()] + [p.grad.float().clone() for p in gru.parameters()])
        for a, b in zip(*grads):
            assert rel(a, b.cpu().numpy()) < (1e-3 if amp is None else 5e-2)


# ------------------------------------------------------------------ 16-bit: per-operation rounding
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", GRU_CASES)
def test_gru_kernels_16bit(R, golden, case, dtype):
    g = golden(case)
    C = int(g["C"])
    rd = _rounder(dtype)
    _, (cz, cr, cq) = _ctx_views(g, dtype)
    h = dev(g["h"], dtype)
    out = R.gru_update(dev(g["zc"], dtype), cz, dev(g["qc"], dtype), cq, h)
    assert out.dtype == dtype
    ref = U.gru_update(g["zc"], g["ctx"][:, :C], g["qc"], g["ctx"][:, 2 * C:], g["h"], round_to=rd)
    assert rel(out, ref) < TOL16[dtype]
    rh = R.gru_gate_r(dev(g["rc"], dtype), cr, h)
    assert rel(rh, U.gru_gate_r(g["rc"], g["ctx"][:, C:2 * C], g["h"], round_to=rd)) < TOL16[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_gru_kernels_equal_torch_ops_at_full_size(R, dtype):
    """KITTI share of one GPU: net[0] is [8,128,96,312].  The chain of torch ops of core/update.py:27-31 on the
    same device is the second reference here (same per-operation rounding), forward and backward."""
    torch.manual_seed(5)
    shp = (8, 128, 96, 312)
    ctx = torch.randn(shp[0], 3 * shp[1], *shp[2:], device="cuda", dtype=dtype)
    cz, cr, cq = ctx.split(split_size=shp[1], dim=1)
    zc, rc, qc = (torch.randn(shp, device="cuda", dtype=dtype) * 2 for _ in range(3))
    h = torch.tanh(torch.randn(shp, device="cuda", dtype=dtype))
    tol = TOL32 if dtype == torch.float32 else TOL16[dtype]
    z = torch.sigmoid(zc + cz)
    r = torch.sigmoid(rc + cr)
    q = torch.tanh(qc + cq)
    ref = (1 - z) * h + z * q
    assert rel(R.gru_update(zc, cz, qc, cq, h), ref.float().cpu().numpy()) < tol
    assert rel(R.gru_gate_r(rc, cr, h), (r * h).float().cpu().numpy()) < tol
    if dtype == torch.float32:
        leaves = [t.clone().requires_grad_(True) for t in (zc, qc, h)]
        gout = torch.randn(shp, device="cuda")
        R.gru_update(leaves[0], cz, leaves[1], cq, leaves[2]).backward(gout)
        mine = [t.grad.clone() for t in leaves]
        ref_leaves = [t.clone().requires_grad_(True) for t in (zc, qc, h)]
        z2, q2 = torch.sigmoid(ref_leaves[0] + cz), torch.tanh(ref_leaves[1] + cq)
        ((1 - z2) * ref_leaves[2] + z2 * q2).backward(gout)
        for a, b in zip(mine, ref_leaves):
            assert rel(a, b.grad.cpu().numpy()) < TOL32


# ------------------------------------------------------------------ convex upsampling
@pytest.mark.parametrize("case", UPSAMPLE_CASES)
def test_upsample_fp32_matches_reference(R, golden, case):
    g = golden(case)
    nd = int(g["n_downsample"])
    with torch.no_grad():
        up = R.upsample_flow_hip(dev(g["flow"]), dev(g["mask"]), nd)
        assert up.dtype == torch.float32 and tuple(up.shape) == g["up"].shape
        assert rel(up, g["up"]) < TOL32
        up1 = R.upsample_flow_hip(dev(g["flow"]), dev(g["mask"]), nd, channels=1)   # raft_stereo.py:134
        assert torch.equal(up1, up[:, :1])


@pytest.mark.parametrize("case", UPSAMPLE_CASES)
def test_upsample_16bit_mask(R, golden, case):
    g = golden(case)
    nd = int(g["n_downsample"])
    with torch.no_grad():
        up = R.upsample_flow_hip(dev(g["flow"]), dev(g["mask"], torch.float16), nd)
        assert rel(up, g["up_mask16"]) < TOL16[torch.float16]        # the reference on the fp16 mask
        ref = U.upsample_flow(g["flow"], _round(g["mask"], torch.float16), 2 ** nd, round_to=np.float16)
        assert rel(up, ref) < 1e-3                                    # one fp16 ulp of a probability
        upb = R.upsample_flow_hip(dev(g["flow"]), dev(g["mask"], torch.bfloat16), nd)
        refb = U.upsample_flow(g["flow"], _round(g["mask"], torch.bfloat16), 2 ** nd,
                               round_to=_rounder(torch.bfloat16))
        assert rel(upb, refb) < 8e-3


@pytest.mark.parametrize("case", UPSAMPLE_CASES)
def test_upsample_backward_matches_reference_autograd(R, golden, case):
    g = golden(case)
    nd = int(g["n_downsample"])
    flow, mask = dev(g["flow"]).requires_grad_(True), dev(g["mask"]).requires_grad_(True)
    R.upsample_flow_hip(flow, mask, nd).backward(dev(g["grad_out"]))
    assert rel(flow.grad, g["d_flow"]) < TOL32
    assert rel(mask.grad, g["d_mask"]) < TOL32
    # only the x component (what the model keeps): the y plane gets no gradient
    flow1, mask1 = dev(g["flow"]).requires_grad_(True), dev(g["mask"]).requires_grad_(True)
    R.upsample_flow_hip(flow1, mask1, nd, channels=1).backward(dev(g["grad_out"][:, :1]))
    df, dm = U.upsample_flow_bwd(g["flow"][:, :1], g["mask"], 2 ** nd, g["grad_out"][:, :1])
    assert rel(flow1.grad[:, :1], df) < TOL32 and float(flow1.grad[:, 1].abs().max()) == 0.0
    assert rel(mask1.grad, dm) < TOL32
    # fp16 mask (autocast): gradients rounded where the reference's fp16 tensors round
    flow2, mask2 = dev(g["flow"]).requires_grad_(True), dev(g["mask"], torch.float16).requires_grad_(True)
    R.upsample_flow_hip(flow2, mask2, nd).backward(dev(g["grad_out"]))
    assert mask2.grad.dtype == torch.float16
    assert rel(flow2.grad, g["d_flow_mask16"]) < TOL16[torch.float16]
    assert rel(mask2.grad, g["d_mask_mask16"]) < TOL16[torch.float16]


def test_upsample_backward_adjoint_identity_at_full_size(R):
    """<upsample(flow + e), g> - <upsample(flow), g> = <e, d flow> exactly (the map is linear in flow), and the
    mask gradient sums to zero over the 9 logits of every output pixel (softmax), at the SceneFlow training size."""
    torch.manual_seed(11)
    N, H, W, nd = 8, 80, 180, 2
    f = 4
    flow = (torch.randn(N, 2, H, W, device="cuda") * 4).requires_grad_(True)
    mask = torch.randn(N, 9 * f * f, H, W, device="cuda").requires_grad_(True)
    gout = torch.randn(N, 2, f * H, f * W, device="cuda")
    up = R.upsample_flow_hip(flow, mask, nd)
    up.backward(gout)
    e = torch.randn_like(flow)
    with torch.no_grad():
        up2 = R.upsample_flow_hip(flow + e, mask, nd)
        lhs = ((up2 - up).double() * gout.double()).sum().item()
        rhs = (e.double() * flow.grad.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * abs(rhs)
    s = mask.grad.view(N, 9, f * f, H, W).sum(dim=1).abs().max().item()
    assert s <= 1e-4 * mask.grad.abs().max().item()


@pytest.mark.parametrize("nd,shape", [(2, (8, 96, 312)), (3, (1, 60, 80)), (2, (1, 504, 752))])
def test_upsample_full_size_properties(R, nd, shape):
    """Size-independent properties at BASELINE sizes: (1) equal logits -> f x the zero-padded 3x3 box mean;
    (2) a constant field stays constant (x f) away from the border; (3) fx/fy placement: a mask that puts all
    weight on the centre tap replicates f*flow into f x f blocks."""
    torch.manual_seed(3)
    N, H, W = shape
    f = 2 ** nd
    flow = torch.randn(N, 2, H, W, device="cuda") * 4
    with torch.no_grad():
        zeros = torch.zeros(N, 9 * f * f, H, W, device="cuda", dtype=torch.float16)
        up = R.upsample_flow_hip(flow, zeros, nd)
        box = torch.nn.functional.avg_pool2d(f * flow, 3, stride=1, padding=1, count_include_pad=True)
        # 1/9 is rounded to fp16 (the mask's type) before the products
        w9 = float(torch.tensor(1.0 / 9).half())
        expect = (box * 9 * w9).repeat_interleave(f, dim=2).repeat_interleave(f, dim=3)
        assert rel(up, expect.cpu().numpy()) < 1e-5
        const = torch.full((N, 2, H, W), 1.5, device="cuda")
        mask = torch.randn(N, 9 * f * f, H, W, device="cuda")
        upc = R.upsample_flow_hip(const, mask, nd)
        assert (upc[:, :, f:-f, f:-f] - 1.5 * f).abs().max().item() < 1e-4
        centre = torch.full((N, 9, f * f, H, W), -1e4, device="cuda")
        centre[:, 4] = 0
        upk = R.upsample_flow_hip(flow, centre.view(N, 9 * f * f, H, W), nd)
        assert torch.equal(upk, (f * flow).repeat_interleave(f, dim=2).repeat_interleave(f, dim=3))


# ------------------------------------------------------------------ sequence loss
@pytest.mark.parametrize("case", LOSS_CASES)
def test_sequence_loss_matches_reference(R, golden, case):
    g = golden(case)
    n = int(g["n"])
    preds = [dev(g[f"pred_{i}"]).requires_grad_(True) for i in range(n)]
    loss, metrics = R.sequence_loss_hip(preds, dev(g["gt"]), dev(g["valid"]), loss_gamma=float(g["gamma"]),
                                        max_flow=float(g["max_flow"]))
    assert loss.dtype == torch.float32 and loss.dim() == 0
    assert abs(loss.item() - float(g["loss"])) <= TOL32 * abs(float(g["loss"]))
    for k, m in enumerate(("epe", "1px", "3px", "5px")):
        assert abs(metrics[m] - float(g["metrics"][k])) <= 1e-5
    (2.0 * loss).backward()                      # an upstream gradient other than 1
    for i in range(n):
        assert rel(preds[i].grad, 2.0 * g[f"d_pred_{i}"]) < TOL32, i


def test_sequence_loss_on_channel_slices_and_full_size(R):
    """Predictions handed over as `flow_up[:, :1]` slices of 2-channel tensors (core/raft_stereo.py:134) at the
    SceneFlow training size (8 x 320 x 720, 22 predictions), against the oracle; deterministic run to run."""
    torch.manual_seed(21)
    N, H, W, n = 8, 320, 720, 22
    gt = -torch.rand(N, 1, H, W, device="cuda") * 100
    valid = (torch.rand(N, H, W, device="cuda") > 0.1).float()
    full = [(gt + torch.randn(N, 2, H, W, device="cuda") * 3).requires_grad_(True) for _ in range(n)]
    preds = [f[:, :1] for f in full]
    loss, metrics = R.sequence_loss_hip(preds, gt, valid)
    loss.backward()
    ref_loss, ref_metrics, ref_grads = U.sequence_loss([p.detach().cpu().numpy() for p in preds], gt.cpu().numpy(),
                                                       valid.cpu().numpy())
    assert abs(loss.item() - ref_loss) <= TOL32 * abs(ref_loss)
    for m in ref_metrics:
        assert abs(metrics[m] - ref_metrics[m]) <= 1e-5
    for i in (0, 7, n - 1):
        assert rel(full[i].grad[:, :1], ref_grads[i]) < TOL32
        assert float(full[i].grad[:, 1].abs().max()) == 0.0
    loss2, _ = R.sequence_loss_hip([p.detach() for p in preds], gt, valid)
    assert loss2.item() == loss.item()
    with pytest.raises(AssertionError):          # train_stereo.py:51-52
        bad = [p.detach().clone() for p in preds]
        bad[3][0, 0, 5, 5] = float("nan")
        R.sequence_loss_hip(bad, gt, valid)


# ------------------------------------------------------------------ between-scale glue
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_pool2x_and_interp_match_reference(R, golden, dtype):
    g = golden("update_resample")
    tol = TOL32 if dtype == torch.float32 else TOL16[dtype]
    rd = (lambda a: a) if dtype == torch.float32 else _rounder(dtype)
    with torch.no_grad():
        for k in [k for k in g if k.startswith("pool_in")]:
            ref = g["pool_out" + k[len("pool_in"):]] if dtype == torch.float32 else U.pool2x(rd(g[k]))
            out = R.pool2x_hip(dev(g[k], dtype))
            assert out.dtype == dtype and tuple(out.shape) == ref.shape
            assert rel(out, ref) < tol, k
        for k in [k for k in g if k.startswith("interp_in")]:
            gold = g["interp_out" + k[len("interp_in"):]]
            ref = gold if dtype == torch.float32 else U.interp(rd(g[k]), *gold.shape[2:])
            out = R.interp_hip(dev(g[k], dtype), torch.empty(1, 1, *gold.shape[2:]))
            assert out.dtype == dtype and tuple(out.shape) == ref.shape
            assert rel(out, ref) < tol, k


def test_pool2x_and_interp_backward_match_reference_autograd(R, golden):
    g = golden("update_resample")
    for k in [k for k in g if k.startswith("pool_in")]:
        i = k[len("pool_in"):]
        x = dev(g[k]).requires_grad_(True)
        R.pool2x_hip(x).backward(dev(g["pool_cot" + i]))
        assert rel(x.grad, g["pool_grad" + i]) < TOL32, k
    for k in [k for k in g if k.startswith("interp_in")]:
        i = k[len("interp_in"):]
        x = dev(g[k]).requires_grad_(True)
        R.interp_hip(x, torch.empty(1, 1, *g["interp_out" + i].shape[2:])).backward(dev(g["interp_cot" + i]))
        assert rel(x.grad, g["interp_grad" + i]) < TOL32, k
    # model size, 16-bit: against torch's own backward
    torch.manual_seed(14)
    lo = torch.randn(2, 128, 40, 90, device="cuda").half().requires_grad_(True)
    cot = torch.randn(2, 128, 80, 180, device="cuda").half()
    R.interp_hip(lo, cot).backward(cot)
    ref = lo.detach().float().requires_grad_(True)
    torch.nn.functional.interpolate(ref, (80, 180), mode="bilinear", align_corners=True).backward(cot.float())
    assert rel(lo.grad, ref.grad.cpu().numpy()) < TOL16[torch.float16]
    hi = torch.randn(2, 128, 80, 180, device="cuda").half().requires_grad_(True)
    cot = torch.randn(2, 128, 40, 90, device="cuda").half()
    R.pool2x_hip(hi).backward(cot)
    ref = hi.detach().float().requires_grad_(True)
    torch.nn.functional.avg_pool2d(ref, 3, stride=2, padding=1).backward(cot.float())
    assert rel(hi.grad, ref.grad.cpu().numpy()) < TOL16[torch.float16]


def test_pool2x_and_interp_equal_torch_at_model_size(R):
    """net[0] -> 1/8 resolution and net[1] -> 1/4 resolution of the KITTI share, against torch's own kernels."""
    torch.manual_seed(12)
    with torch.no_grad():
        for dtype in (torch.float32, torch.float16):
            hi = torch.randn(8, 128, 96, 312, device="cuda").to(dtype)
            lo = torch.randn(8, 128, 48, 156, device="cuda").to(dtype)
            tol = TOL32 if dtype == torch.float32 else TOL16[dtype]
            ref = torch.nn.functional.avg_pool2d(hi.float(), 3, stride=2, padding=1)
            assert rel(R.pool2x_hip(hi), ref.cpu().numpy()) < tol
            ref = torch.nn.functional.interpolate(lo.float(), (96, 312), mode="bilinear", align_corners=True)
            assert rel(R.interp_hip(lo, hi), ref.cpu().numpy()) < tol


def test_update_ops_replay_from_a_hip_graph(R):
    """No entry point allocates or synchronises, so the pointwise GRU kernels and the upsampling can sit inside a
    captured GRU loop: the replayed graph gives the eager results."""
    gen = torch.Generator(device="cuda").manual_seed(41)
    N, C, H, W = 1, 32, 12, 40
    ctx = torch.randn((N, 3 * C, H, W), generator=gen, device="cuda").half()
    cz, cr, cq = ctx.split(C, dim=1)
    zr = torch.randn((N, 2 * C, H, W), generator=gen, device="cuda").half()
    qc = torch.randn((N, C, H, W), generator=gen, device="cuda").half()
    h0 = torch.tanh(torch.randn((N, C, H, W), generator=gen, device="cuda")).half()
    flow = torch.randn((N, 2, H, W), generator=gen, device="cuda")
    mask = torch.randn((N, 9 * 16, H, W), generator=gen, device="cuda").half()

    def loop():
        h, outs = h0, []
        for _ in range(3):
            buf = torch.cat([h, h], dim=1)
            R.gru_gate_r(zr[:, C:], cr, h, out=buf[:, :C])
            h = R.gru_update(zr[:, :C], cz, qc, cq, h)
            outs += [buf, h]
        outs.append(R.upsample_flow_hip(flow, mask, 2, channels=1))
        return outs

    with torch.no_grad():
        eager = [o.clone() for o in loop()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            loop()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = loop()
        for o in captured:
            o.zero_()
        graph.replay()
        torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(captured, eager))


def test_update_argument_errors(R):
    h = torch.zeros(1, 4, 2, 2)
    with pytest.raises(RuntimeError):
        R.gru_update(h, h, h, h, h)                       # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        R.upsample_flow_hip(torch.zeros(1, 2, 2, 2), torch.zeros(1, 144, 2, 2), 2)
    hc = h.cuda()
    with pytest.raises(ValueError):
        R.gru_update(hc, hc, hc, hc[:, :2], hc)           # shape mismatch
    with pytest.raises(ValueError):
        with torch.no_grad():
            R.upsample_flow_hip(torch.zeros(1, 2, 2, 2).cuda(), torch.zeros(1, 100, 2, 2).cuda(), 2)
    from raft_stereo_amd._lib import lib
    assert lib.rc_upsample_flow_fwd(None, 0, None, None, 1, 1, 1, 1, 4, 0, None) != 0
    assert lib.rc_upsample_flow_fwd(1, 8, 1, 16, 1, 2, 2, 2, 3, 0, None) != 0      # factor 3
    assert lib.rc_gru_update(None, 0, None, 0, None, 0, None, 0, None, None, 1, 1, 0, None) != 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hw", [(12, 40), (7, 13), (5, 6), (3, 12)])      # 16-, 2-, 4- and 8-byte accesses at fp16
def test_bias_act_equals_torch_add_and_relu_bit_for_bit(R, dtype, hw):
    """rc_bias_act against the two torch passes it replaces (broadcast add_ of the bias cast to the tensor's
    type, then relu): same roundings, so the bar is equality; fp32 and same-type bias, NaN kept."""
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((3, 10, *hw), generator=g, device="cuda").to(dtype)
    x[0, 0, 0, 0] = float("nan")
    b32 = torch.randn(10, generator=g, device="cuda")
    for bias in (b32, b32.to(dtype)):
        for relu in (False, True):
            want = x + bias.to(dtype).view(1, -1, 1, 1)
            want = torch.relu(want) if relu else want
            got = R.bias_act_(x.clone(), bias, relu)
            assert torch.equal(torch.nan_to_num(got, nan=123.0), torch.nan_to_num(want, nan=123.0))
            assert torch.isnan(got[0, 0, 0, 0])


@pytest.mark.parametrize("amp", [False, True])
def test_conv_bias_act_equals_the_module_call(R, amp):
    """conv_bias_act(conv, x, relu) against F.relu(conv(x)) with torch's own bias handling, fp32 and autocast."""
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(24, 40, 3, padding=1).cuda()
    x = torch.randn(2, 24, 20, 36, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        for relu in (False, True):
            want = conv(x)
            want = torch.relu(want) if relu else want
            got = R.conv_bias_act(conv, x, relu)
            assert got.dtype == want.dtype
            # the bias-free convolution may pick another MIOpen solver than the biased call: value bar, not bits
            assert rel(got, want.float().cpu().numpy()) <= (2e-3 if amp else TOL_CONV)
    # with gradients it is the torch expression itself
    y = R.conv_bias_act(conv, x.requires_grad_(True), True)
    y.sum().backward()
    assert conv.bias.grad is not None and x.grad is not None
