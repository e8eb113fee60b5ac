"""Pin the CPU oracle (oracle/corr_oracle.py) to vectors captured from the reference's own
CorrBlock1D / PytorchAlternateCorrBlock1D (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import corr_oracle as O
from conftest import CONVC1_CASES, GOLDEN_CASES, MODEL_CASE

KINDS = ["smooth", "random", "edge"]
TOL = 1e-5          # max|a-b| / max|ref|, the fp32 bar of BASELINE.json


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_volume_and_pyramid(golden, case):
    g = golden(case)
    L = int(g["L"])
    pyr = O.pool_pyramid(O.corr_volume(g["f1"], g["f2"]), L)
    for i in range(L):
        assert pyr[i].shape == g[f"level{i}"].shape
        assert O.rel_to_max(pyr[i], g[f"level{i}"]) < 2e-6


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("kind", KINDS)
def test_lookup_gridsample_matches_reg(golden, case, kind):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    pyr = [g[f"level{i}"] for i in range(L)]
    out = O.lookup_gridsample(pyr, g[f"coords_{kind}"], r)
    assert out.shape == g[f"lookup_{kind}"].shape
    assert O.rel_to_max(out, g[f"lookup_{kind}"]) < 2e-6


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("kind", KINDS)
def test_lookup_exact_close_to_reg(golden, case, kind):
    """The floor+lerp form (sampler_kernel.cu, the reference's own native lookup) differs from
    the reg path only through the fp32 round trip of the tap coordinate in bilinear_sampler:
    a few ulp(x) times the local slope, i.e. up to ~1e-5..1e-4 of max at widths of hundreds.
    That is why the HIP lookup has a grid_sample-compatible coordinate mode for fp32 parity."""
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    pyr = [g[f"level{i}"] for i in range(L)]
    out = O.lookup_exact(pyr, g[f"coords_{kind}"], r)
    assert O.rel_to_max(out, g[f"lookup_{kind}"]) < 1e-4


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_far_out_of_range_is_zero(golden, case):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    c = g["coords_edge"]
    far = np.abs(c[:, 0]) >= 1e3
    assert far.any()
    ref = g["lookup_edge"].transpose(0, 2, 3, 1)[far]
    assert np.all(ref == 0)
    out = O.lookup_exact([g[f"level{i}"] for i in range(L)], c, r).transpose(0, 2, 3, 1)[far]
    assert np.all(out == 0)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_backward_single_and_multi(golden, case):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    widths = [g[f"level{i}"].shape[-1] for i in range(L)]
    acc = None
    for n, kind in enumerate(KINDS):
        lg = O.lookup_bwd(g[f"coords_{kind}"], g[f"gout_{kind}"], widths, r, O.GRIDSAMPLE)
        acc = lg if acc is None else [a + b for a, b in zip(acc, lg)]
        if n == 0:
            df1, df2 = O.corr_volume_bwd(O.pool_pyramid_bwd(acc), g["f1"], g["f2"])
            assert O.rel_to_max(df1, g["df1_single"]) < TOL
            assert O.rel_to_max(df2, g["df2_single"]) < TOL
    df1, df2 = O.corr_volume_bwd(O.pool_pyramid_bwd(acc), g["f1"], g["f2"])
    assert O.rel_to_max(df1, g["df1_multi"]) < TOL
    assert O.rel_to_max(df2, g["df2_multi"]) < TOL


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("kind", KINDS)
def test_alt_matches_reference_alt_and_reg(golden, case, kind):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    out = O.alt_lookup(g["f1"], g["f2"], g[f"coords_{kind}"], L, r, O.GRIDSAMPLE)
    # the bar is the reg path; the reference's own alt also round-trips y, so it is looser
    assert O.rel_to_max(out, g[f"lookup_{kind}"]) < TOL
    assert O.rel_to_max(out, g[f"alt_{kind}"]) < 5e-5
    out = O.alt_lookup(g["f1"], g["f2"], g[f"coords_{kind}"], L, r, O.EXACT)
    assert O.rel_to_max(out, g[f"lookup_{kind}"]) < 1e-4


@pytest.mark.parametrize("case", CONVC1_CASES)
@pytest.mark.parametrize("kind", ["smooth", "random", "edge"])
def test_convc1_relu_matches_reference_motion_encoder(golden, case, kind):
    """oracle lookup -> oracle convc1+relu against the reference's BasicMotionEncoder.convc1
    (core/update.py:71,78) applied to its own CorrBlock1D output."""
    g, c = golden(case), golden("convc1_" + case)
    L, r = int(g["L"]), int(g["r"])
    pyr = O.pool_pyramid(O.corr_volume(g["f1"], g["f2"]), L)
    corr = O.lookup_gridsample(pyr, g[f"coords_{kind}"], r)
    act = O.convc1_relu(corr, c["weight"], c["bias"])
    assert O.rel_to_max(act, c[f"act_{kind}"]) < 1e-5
    assert (c[f"act_{kind}"] == 0).any() and (c[f"act_{kind}"] > 0).any()   # the ReLU is exercised
    dc, dw, db = O.convc1_relu_bwd(g[f"lookup_{kind}"], c["weight"], c["bias"], c[f"gact_{kind}"])
    assert O.rel_to_max(dc, c[f"d_corr_{kind}"]) < 1e-5
    assert O.rel_to_max(dw, c[f"d_weight_{kind}"].reshape(dw.shape)) < 1e-5
    assert O.rel_to_max(db, c[f"d_bias_{kind}"]) < 1e-5


def test_oracle_replays_the_reference_model_run(golden):
    """Feature maps, per-iteration coords, block outputs and relu(convc1(corr)) recorded inside a
    forward pass of the reference's RAFTStereo (tests/golden/make_golden_model.py)."""
    g = golden(MODEL_CASE)
    L, r = int(g["L"]), int(g["r"])
    pyr = O.pool_pyramid(O.corr_volume(g["f1"], g["f2"]), L)
    for k in range(int(g["n_iters"])):
        corr = O.lookup_gridsample(pyr, g[f"coords_{k}"], r)
        assert O.rel_to_max(corr, g[f"corr_{k}"]) < 1e-5
        assert O.rel_to_max(O.convc1_relu(corr, g["convc1_weight"], g["convc1_bias"]), g[f"act_{k}"]) < 1e-5
        if k:
            assert not np.array_equal(g[f"coords_{k}"], g[f"coords_{k - 1}"])     # the GRU moved the coords
            assert np.array_equal(g[f"coords_{k}"][:, 1], g["coords_0"][:, 1])    # only along x


def test_pool_rounding_hook():
    v = np.linspace(-3, 3, 4 * 5 * 6 * 16, dtype=np.float32).reshape(4, 5, 6, 16)
    p16 = O.pool_pyramid(v, 3, round_to=np.float16)
    assert all(np.array_equal(x, x.astype(np.float16).astype(np.float32)) for x in p16)
    pb = O.pool_pyramid(v, 3, round_to=O.round_bf16)
    assert all(np.array_equal(x, O.round_bf16(x)) for x in pb)


def test_reg_torch_port_matches_golden(golden):
    """The torch-op port used as the CPU baseline reproduces the reference's outputs."""
    import torch
    from oracle.reg_torch import RegCorrCPU
    g = golden("tiny_b2_d32_h3_w24")
    blk = RegCorrCPU(torch.from_numpy(g["f1"]), torch.from_numpy(g["f2"]), 4, 4)
    for kind in KINDS:
        out = blk(torch.from_numpy(g[f"coords_{kind}"])).numpy()
        assert O.rel_to_max(out, g[f"lookup_{kind}"]) < 2e-6


# ----------------------------------------------------------------------------- GRU gates, upsampling
from conftest import GRU_CASES, UPSAMPLE_CASES  # noqa: E402
from oracle import update_oracle as U  # noqa: E402


def _gru_ctx(g):
    C = int(g["C"])
    return g["ctx"][:, :C], g["ctx"][:, C:2 * C], g["ctx"][:, 2 * C:]


@pytest.mark.parametrize("case", GRU_CASES)
def test_gru_pointwise_matches_reference_convgru(golden, case):
    """The oracle's gate arithmetic on the convolution outputs the reference's ConvGRU produced gives the
    reference's new hidden state (core/update.py:27-31), and its adjoints give the reference's autograd
    gradients at the convolution outputs."""
    g = golden(case)
    cz, cr, cq = _gru_ctx(g)
    assert U.rel_to_max(U.gru_update(g["zc"], cz, g["qc"], cq, g["h"]), g["h_out"]) <= 1e-6
    dz, dq, dh_u = U.gru_update_bwd(g["zc"], cz, g["qc"], cq, g["h"], g["grad_out"])
    assert U.rel_to_max(dz, g["d_zc"]) <= 1e-5
    assert U.rel_to_max(dq, g["d_qc"]) <= 1e-5
    # d ctx = the three pre-activation gradients side by side (cz/cr/cq are split views, raft_stereo.py:88)
    C = int(g["C"])
    assert U.rel_to_max(dz, g["d_ctx"][:, :C]) <= 1e-5
    assert U.rel_to_max(g["d_rc"], g["d_ctx"][:, C:2 * C]) <= 1e-6
    assert U.rel_to_max(dq, g["d_ctx"][:, 2 * C:]) <= 1e-5
    # r*h is what convq read in its first C input channels, so d(r*h) is the input gradient of that
    # convolution; with it the gate adjoint must give the reference's d_rc, and the three routes into h
    # (blend, r*h, the z/r convolutions' input) must add up to the reference's d_h
    import torch
    from torch.nn.grad import conv2d_input

    def conv_in_grad(d_out, w):
        shape = (d_out.shape[0], w.shape[1]) + d_out.shape[2:]
        return conv2d_input(shape, torch.from_numpy(w), torch.from_numpy(d_out), padding=1).numpy()
    d_rh = conv_in_grad(g["d_qc"], g["w_convq.weight"])[:, :C]
    dr, dh_r = U.gru_gate_r_bwd(g["rc"], cr, g["h"], d_rh)
    assert U.rel_to_max(dr, g["d_rc"]) <= 1e-5
    d_h = dh_u + dh_r + conv_in_grad(g["d_zc"], g["w_convz.weight"])[:, :C] \
        + conv_in_grad(g["d_rc"], g["w_convr.weight"])[:, :C]
    assert U.rel_to_max(d_h, g["d_h"]) <= 1e-5
    # forward of the gate: the reference's convq read [r*h, x] (update.py:29) and produced qc
    rhx = np.concatenate([U.gru_gate_r(g["rc"], cr, g["h"])] + [g[f"x{i}"] for i in range(int(g["n_x"]))], axis=1)
    qc = torch.nn.functional.conv2d(torch.from_numpy(rhx), torch.from_numpy(g["w_convq.weight"]),
                                    torch.from_numpy(g["w_convq.bias"]), padding=1).numpy()
    assert U.rel_to_max(qc, g["qc"]) <= 1e-5


@pytest.mark.parametrize("case", UPSAMPLE_CASES)
def test_upsample_matches_reference(golden, case):
    g = golden(case)
    f = 2 ** int(g["n_downsample"])
    assert U.rel_to_max(U.upsample_flow(g["flow"], g["mask"], f), g["up"]) <= 1e-6
    # fp16 mask (autocast): softmax returns fp16 probabilities
    assert U.rel_to_max(U.upsample_flow(g["flow"], g["mask"].astype(np.float16), f, round_to=np.float16),
                        g["up_mask16"]) <= 2e-3


@pytest.mark.parametrize("case", UPSAMPLE_CASES)
def test_upsample_backward_matches_reference_autograd(golden, case):
    g = golden(case)
    f = 2 ** int(g["n_downsample"])
    df, dm = U.upsample_flow_bwd(g["flow"], g["mask"], f, g["grad_out"])
    assert U.rel_to_max(df, g["d_flow"]) <= 1e-5
    assert U.rel_to_max(dm, g["d_mask"]) <= 1e-5
    df, dm = U.upsample_flow_bwd(g["flow"], g["mask"].astype(np.float16), f, g["grad_out"], round_to=np.float16)
    assert U.rel_to_max(df, g["d_flow_mask16"]) <= 2e-3
    assert U.rel_to_max(dm, g["d_mask_mask16"]) <= 2e-3


from conftest import LOSS_CASES  # noqa: E402


@pytest.mark.parametrize("case", LOSS_CASES)
def test_sequence_loss_matches_reference(golden, case):
    g = golden(case)
    n = int(g["n"])
    preds = [g[f"pred_{i}"] for i in range(n)]
    loss, metrics, grads = U.sequence_loss(preds, g["gt"], g["valid"], float(g["gamma"]), float(g["max_flow"]))
    assert abs(loss - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for k, m in enumerate(("epe", "1px", "3px", "5px")):
        assert abs(metrics[m] - float(g["metrics"][k])) <= 1e-5
    for i in range(n):
        assert U.rel_to_max(grads[i], g[f"d_pred_{i}"]) <= 1e-5


def test_pool2x_and_interp_match_reference(golden):
    g = golden("update_resample")
    for k in [k for k in g if k.startswith("pool_in")]:
        assert U.rel_to_max(U.pool2x(g[k]), g["pool_out" + k[len("pool_in"):]]) <= 1e-6, k
        i = k[len("pool_in"):]
        assert U.rel_to_max(U.pool2x_bwd(g["pool_cot" + i], *g[k].shape[2:]), g["pool_grad" + i]) <= 1e-6, k
    for k in [k for k in g if k.startswith("interp_in")]:
        ref = g["interp_out" + k[len("interp_in"):]]
        assert U.rel_to_max(U.interp(g[k], *ref.shape[2:]), ref) <= 1e-6, k
        i = k[len("interp_in"):]
        assert U.rel_to_max(U.interp_bwd(g["interp_cot" + i], *g[k].shape[2:]), g["interp_grad" + i]) <= 1e-5, k


def test_rounded_sampler_oracle_reduces_to_the_exact_lookup():
    """sampler_fwd/bwd_rounded (the scalar_t arithmetic of sampler_kernel.cu:50-56, :94-102) with the identity as
    rounding function are the plain floor + lerp lookup and its adjoint of one level."""
    import numpy as np
    from oracle import corr_oracle as O
    rng = np.random.default_rng(5)
    vol = rng.standard_normal((2, 3, 7, 19)).astype(np.float32)
    coords = np.stack([rng.uniform(-6, 25, (2, 3, 7)), np.zeros((2, 3, 7))], 1).astype(np.float32)
    coords[0, 0, 0, 0], coords[0, 0, 0, 1] = 4.0, 1e9
    ident = lambda a: np.asarray(a, np.float32)
    out = O.sampler_fwd_rounded(vol, coords, 3, ident)
    assert O.rel_to_max(out, O.lookup_exact([vol], coords, 3)) < 1e-6
    g = rng.standard_normal(out.shape).astype(np.float32)
    gv = O.sampler_bwd_rounded(coords, g, 19, 3, ident)
    assert O.rel_to_max(gv, O.lookup_bwd(coords, g, [19], 3, O.EXACT)[0]) < 1e-6
    assert abs(float((out * g).sum()) - float((vol * gv).sum())) < 1e-3          # adjoint identity
    half = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)
    assert np.abs(O.sampler_fwd_rounded(half(vol), coords, 3, half) - out).max() < 2e-2
