"""GPU parity tests: the HIP path (through the C-ABI library) against the CPU oracle and the
golden vectors captured from the reference.  Run on the GPU box with `-m gpu`.

Bars (metric max|a-b| / max|ref|):
  fp32  : 1e-5 against the reference's reg path (golden vectors)
  fp16  : 2e-3, bf16: 1.6e-2 against the fp32 oracle evaluated on the rounded inputs
"""
import numpy as np
import pytest
import torch

from oracle import corr_oracle as O
from conftest import CONVC1_CASES, GOLDEN_CASES, MODEL_CASE

pytestmark = pytest.mark.gpu

KINDS = ["smooth", "random", "edge"]
TOL32 = 1e-5
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
    return O.rel_to_max(a.detach().float().cpu().numpy() if torch.is_tensor(a) else a, b)


# ------------------------------------------------------------------ fp32: golden vectors
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_build_fp32_matches_reference_pyramid(R, golden, case):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    blk = R.CorrBlock1DHip(dev(g["f1"]), dev(g["f2"]), num_levels=L, radius=r)
    for i in range(L):
        got = blk.dense_level(i)
        assert tuple(got.shape) == g[f"level{i}"].shape
        assert rel(got, g[f"level{i}"]) < TOL32, f"level {i}"


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("kind", KINDS)
def test_lookup_fp32_matches_reference_reg(R, golden, case, kind):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    blk = R.CorrBlock1DHip(dev(g["f1"]), dev(g["f2"]), num_levels=L, radius=r)
    out = blk(dev(g[f"coords_{kind}"]))
    assert out.dtype == torch.float32 and out.is_contiguous()
    assert tuple(out.shape) == g[f"lookup_{kind}"].shape
    assert rel(out, g[f"lookup_{kind}"]) < TOL32
    # 1-channel coords (what CorrBlockFast1D passes, core/corr.py:47) give the same result
    out1 = blk(dev(g[f"coords_{kind}"][:, :1]))
    assert torch.equal(out, out1)


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("kind", KINDS)
def test_lookup_exact_mode_matches_sampler_semantics(R, golden, case, kind):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    blk = R.CorrBlock1DHip(dev(g["f1"]), dev(g["f2"]), num_levels=L, radius=r, coord_mode="exact")
    pyr = [blk.dense_level(i).cpu().numpy() for i in range(L)]
    out = blk(dev(g[f"coords_{kind}"]))
    assert rel(out, O.lookup_exact(pyr, g[f"coords_{kind}"], r)) < 2e-6
    far = np.abs(g[f"coords_{kind}"][:, 0]) >= 1e3
    if far.any():
        assert np.all(out.cpu().numpy().transpose(0, 2, 3, 1)[far] == 0)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_backward_fp32_matches_reference_autograd(R, golden, case):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    f1 = dev(g["f1"]).requires_grad_(True)
    f2 = dev(g["f2"]).requires_grad_(True)
    blk = R.CorrBlock1DHip(f1, f2, num_levels=L, radius=r)
    outs = [blk(dev(g[f"coords_{k}"])) for k in KINDS]
    assert all(o.requires_grad for o in outs)
    d1, d2 = torch.autograd.grad((outs[0] * dev(g["gout_smooth"])).sum(), (f1, f2), retain_graph=True)
    assert rel(d1, g["df1_single"]) < TOL32 and rel(d2, g["df2_single"]) < TOL32
    tot = sum((o * dev(g[f"gout_{k}"])).sum() for o, k in zip(outs, KINDS))
    tot.backward()
    assert rel(f1.grad, g["df1_multi"]) < TOL32
    assert rel(f2.grad, g["df2_multi"]) < TOL32


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("kind", KINDS)
def test_alt_fp32_matches_reference_reg(R, golden, case, kind):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    blk = R.AltCorrBlock1DHip(dev(g["f1"]), dev(g["f2"]), num_levels=L, radius=r)
    out = blk(dev(g[f"coords_{kind}"]))
    assert rel(out, g[f"lookup_{kind}"]) < TOL32
    assert rel(out, g[f"alt_{kind}"]) < 5e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("levels,radius,shape", [(1, 4, (2, 64, 3, 50, 50)), (2, 1, (1, 32, 2, 37, 61)),
                                                  (3, 6, (1, 96, 2, 90, 77)), (4, 4, (2, 256, 2, 75, 75)),
                                                  (3, 2, (1, 64, 1, 17, 40))])
def test_alt_equals_reg_every_level_count(R, dtype, levels, radius, shape):
    """alt_hip serves levels in pairs from the even levels' copies of f2 (an odd level's correlations are the
    means of two of the level below); reg_hip reads the materialised pyramid.  Same function of the inputs at
    every level count (a last level that stands alone included), odd widths, W1 != W2, coordinates far outside
    the image, both coordinate modes."""
    B, D, H, W1, W2 = shape
    gen = torch.Generator(device="cuda").manual_seed(levels * 100 + radius)
    f1 = torch.randn((B, D, H, W1), generator=gen, device="cuda").to(dtype)
    f2 = torch.randn((B, D, H, W2), generator=gen, device="cuda").to(dtype)
    coords = torch.zeros((B, 2, H, W1), device="cuda")
    coords[:, 0] = torch.rand((B, H, W1), generator=gen, device="cuda") * (W2 + 30) - 15
    coords[:, 0, 0, 0] = -1e4
    coords[:, 0, -1, -1] = 3e9
    coords[:, 0, 0, 1] = 7.0
    for mode in ("exact", "gridsample"):
        with torch.no_grad():
            reg = R.CorrBlock1DHip(f1, f2, num_levels=levels, radius=radius, coord_mode=mode,
                                   volume_dtype=torch.float32)(coords)
            alt = R.AltCorrBlock1DHip(f1, f2, num_levels=levels, radius=radius, coord_mode=mode)(coords)
        assert alt.shape == reg.shape == (B, levels * (2 * radius + 1), H, W1)
        assert rel(alt, reg.cpu().numpy()) < (2e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("radius", [1, 4, 6])
def test_alt_streamed_f1_is_the_same_lookup(R, dtype, radius, monkeypatch):
    """Operands too large for the Infinity Cache pick the kernel that fetches f1's fragments past the cache
    (rc_alt_fwd's size rule; full-depth features only).  A cache policy, not arithmetic: forced on a small
    frame it returns the plain kernel's bits, both coordinate modes.  (The size rule itself is what
    test_middlebury_full_size_alt_and_row_slabs runs: its full frame streams, its slabs do not.)"""
    B, D, H, W1, W2, L = 2, 256, 3, 75, 83, 4
    gen = torch.Generator(device="cuda").manual_seed(radius)
    f1 = torch.randn((B, D, H, W1), generator=gen, device="cuda").to(dtype)
    f2 = torch.randn((B, D, H, W2), generator=gen, device="cuda").to(dtype)
    coords = torch.zeros((B, 2, H, W1), device="cuda")
    coords[:, 0] = torch.rand((B, H, W1), generator=gen, device="cuda") * (W2 + 30) - 15
    for mode in ("exact", "gridsample"):
        alt = R.AltCorrBlock1DHip(f1, f2, num_levels=L, radius=radius, coord_mode=mode)
        got = {}
        for policy in ("plain", "stream"):
            monkeypatch.setenv("RC_ALT_F1_POLICY", policy)
            with torch.no_grad():
                got[policy] = alt(coords)
        monkeypatch.delenv("RC_ALT_F1_POLICY")
        assert float(got["plain"].abs().max()) > 0
        assert torch.equal(got["plain"], got["stream"]), mode


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("levels,shape", [(4, (2, 256, 3, 752, 752)), (4, (1, 96, 2, 72, 200)), (3, (1, 64, 2, 136, 40)),
                                          (2, (1, 32, 1, 8, 16)), (1, (1, 160, 2, 264, 120)), (4, (1, 256, 2, 184, 88))])
def test_alt_prepare_vector_kernel_writes_the_same_working_copies(R, dtype, levels, shape, monkeypatch):
    """rc_alt_prepare has two kernels for 16-bit features: 2-byte loads and an fp32 tile (any width), and 16-byte
    loads with the transposing LDS read (rows of whole 16-byte segments).  Same working copies, every element of
    them (padding positions and channels included), so every lookup is the same whichever ran."""
    B, D, H, W1, W2 = shape
    gen = torch.Generator(device="cuda").manual_seed(levels)
    f1 = torch.randn((B, D, H, W1), generator=gen, device="cuda").to(dtype)
    f2 = torch.randn((B, D, H, W2), generator=gen, device="cuda").to(dtype)
    got = {}
    for how in ("elements", "vector"):
        monkeypatch.setenv("RC_ALT_PREPARE", how)
        blk = R.AltCorrBlock1DHip(f1, f2, num_levels=levels, radius=2)
        # poison first: an element neither kernel writes would show
        blk.f1t.view(torch.int16).fill_(0x7bff)
        blk.f2t.view(torch.int16).fill_(0x7bff)
        from raft_stereo_amd._lib import lib, check
        check(lib.rc_alt_prepare(f1.data_ptr(), f2.data_ptr(), blk.f1t.data_ptr(), blk.f2t.data_ptr(), blk.layout,
                                 blk.in_dtype, torch.cuda.current_stream().cuda_stream), "rc_alt_prepare")
        got[how] = (blk.f1t.view(torch.int16).clone(), blk.f2t.view(torch.int16).clone())
    monkeypatch.delenv("RC_ALT_PREPARE")
    assert int((got["vector"][0] != 0).sum()) > 0
    assert torch.equal(got["elements"][0], got["vector"][0]), "f1t"
    assert torch.equal(got["elements"][1], got["vector"][1]), "f2t"


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_alt_backward_fp32_matches_reference_autograd(R, golden, case):
    """alt_hip is the same function of the feature maps as reg, so the reference's autograd
    gradients of the reg path pin its backward too (core/corr.py:64-107 relies on autograd)."""
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    f1 = dev(g["f1"]).requires_grad_(True)
    f2 = dev(g["f2"]).requires_grad_(True)
    blk = R.AltCorrBlock1DHip(f1, f2, num_levels=L, radius=r)
    outs = [blk(dev(g[f"coords_{k}"])) for k in KINDS]
    assert all(o.requires_grad for o in outs)
    sum((o * dev(g[f"gout_{k}"])).sum() for o, k in zip(outs, KINDS)).backward()
    assert rel(f1.grad, g["df1_multi"]) < TOL32
    assert rel(f2.grad, g["df2_multi"]) < TOL32


def test_replay_of_the_reference_model_run(R, golden):
    """The block inside a forward pass of the reference's RAFTStereo: its recorded feature maps in,
    its recorded per-iteration outputs back -- reg_hip, alt_hip, the fused convc1 and the fused
    coordinate update, all through the C-ABI."""
    g = golden(MODEL_CASE)
    L, r, n = int(g["L"]), int(g["r"]), int(g["n_iters"])
    w, bias = dev(g["convc1_weight"]), dev(g["convc1_bias"])
    with torch.no_grad():
        reg = R.make_corr_fn("reg_hip", dev(g["f1"]), dev(g["f2"]), radius=r, num_levels=L)
        alt = R.make_corr_fn("alt_hip", dev(g["f1"]), dev(g["f2"]), radius=r, num_levels=L)
        coords0 = dev(g["coords_0"])              # initialize_flow: the pixel grid (core/raft_stereo.py:50-51)
        coords1 = coords0
        for k in range(n):
            c = dev(g[f"coords_{k}"])
            assert rel(reg(c), g[f"corr_{k}"]) < TOL32
            assert rel(alt(c), g[f"corr_{k}"]) < TOL32
            assert rel(reg.lookup_convc1(c, w, bias), g[f"act_{k}"]) < TOL32
            if k:
                # the step the GRU took, replayed through the fused update: same coords, flow and corr
                delta = torch.zeros_like(c)
                delta[:, 0] = c[:, 0] - coords1[:, 0]
                delta[:, 1] = 123.0              # the reference zeroes this channel (core/raft_stereo.py:120)
                new, flow, corr = reg.update_and_lookup(coords1, delta)
                assert rel(new, g[f"coords_{k}"]) < 1e-6 and rel(corr, g[f"corr_{k}"]) < TOL32
                assert torch.equal(flow, new - coords0)
                coords1 = new


# ------------------------------------------------------------------ MFMA kernel vs the naive kernel
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 256, 3, 312, 312), (1, 96, 2, 70, 131), (1, 256, 2, 200, 45),
                                   (1, 256, 3, 180, 180), (1, 256, 2, 90, 150), (1, 256, 2, 752, 94)])
def test_build_mfma_equals_naive_kernel(R, dtype, shape):
    from raft_stereo_amd._lib import lib, check
    B, D, H, W1, W2 = shape
    gen = torch.Generator(device="cuda").manual_seed(7)
    # small integers: every product and partial sum is exact in all three types
    f1 = torch.randint(-3, 4, (B, D, H, W1), generator=gen, device="cuda").to(dtype)
    f2 = torch.randint(-3, 4, (B, D, H, W2), generator=gen, device="cuda").to(dtype)
    blk = R.CorrBlock1DHip(f1, f2, num_levels=4, radius=4)
    ref = torch.full_like(blk.volume, float("nan"))
    check(lib.rc_build_pyramid_naive(f1.data_ptr(), f2.data_ptr(), ref.data_ptr(), blk.layout, D, blk.in_dtype,
                                     torch.cuda.current_stream().cuda_stream), "naive")
    for i in range(4):
        a = blk.dense_level(i, volume=ref).float()
        b = blk.dense_level(i).float()
        assert torch.equal(a, b), f"level {i} differs: max {float((a - b).abs().max())}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 256, 5, 312, 312), (1, 256, 3, 136, 200), (1, 128, 2, 752, 752),
                                   (1, 256, 4, 384, 96), (1, 256, 16, 72, 312)])
def test_build_cache_policies_are_the_same_pyramid(R, dtype, shape, monkeypatch):
    """rc_build_pyramid picks by the pyramid's size whether f1's tiles are fetched past the Infinity Cache
    (up to 200 MB: yes, so the pyramid is still cached for the lookups; 64 KITTI pairs: no).  Only the bench's
    largest workload reaches the second kernel by size, so it is forced here: same bits as the first, on
    real-valued data, and both the same bits as the kernel without LDS-DMA -- and as either order in which the
    image rows are dealt to the XCDs (RC_BUILD_ROW_ORDER)."""
    B, D, H, W1, W2 = shape
    gen = torch.Generator(device="cuda").manual_seed(23)
    f1 = torch.randn((B, D, H, W1), generator=gen, device="cuda").to(dtype)
    f2 = torch.randn((B, D, H, W2), generator=gen, device="cuda").to(dtype)

    def at_offset(t, elems):  # the same values in storage that starts `elems` elements (16-byte pieces) into a line
        flat = torch.empty(t.numel() + elems, dtype=t.dtype, device=t.device)
        view = flat[elems:].view(t.shape)
        view.copy_(t)
        assert view.is_contiguous() and view.data_ptr() % 128 == (flat.data_ptr() + 2 * elems) % 128
        return view

    # (feature maps that do not start on a 128-byte line: the line-tile shift then comes from the pointer, too)
    f1, f2 = at_offset(f1, 8 * (H % 5)), at_offset(f2, 8 * (W1 % 7))
    vols = {}
    for name, env in (("stream", {"RC_BUILD_F1_POLICY": "stream"}), ("plain", {"RC_BUILD_F1_POLICY": "plain"}),
                      ("no_dma", {"RC_BUILD_NO_DMA": "1"}), ("interleaved", {"RC_BUILD_ROW_ORDER": "interleaved"}),
                      ("no_dma_interleaved", {"RC_BUILD_NO_DMA": "1", "RC_BUILD_ROW_ORDER": "interleaved"}),
                      ("line_tiles", {"RC_BUILD_F1_TILES": "line"}), ("row_tiles", {"RC_BUILD_F1_TILES": "row"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        blk = R.CorrBlock1DHip(f1, f2, num_levels=4, radius=4)
        vols[name] = [blk.dense_level(i).clone() for i in range(4)]
        for k in env:
            monkeypatch.delenv(k)
    for i in range(4):
        assert float(vols["stream"][i].float().abs().max()) > 0
        assert torch.equal(vols["stream"][i], vols["plain"][i]), f"level {i}: stream / plain"
        assert torch.equal(vols["stream"][i], vols["no_dma"][i]), f"level {i}: LDS-DMA / round 2's kernel"
        # which XCD gets which image rows (B*H = 10, 3, 2 rows: not multiples of 8) changes no bit either
        assert torch.equal(vols["stream"][i], vols["interleaved"][i]), f"level {i}: row order"
        assert torch.equal(vols["stream"][i], vols["no_dma_interleaved"][i]), f"level {i}: row order, round 2's kernel"
        # f1 tiles that start on 128-byte lines of memory (first tile narrower, columns before the row masked)
        assert torch.equal(vols["line_tiles"][i], vols["row_tiles"][i]), f"level {i}: f1 tiles on lines / on the row"
        assert torch.equal(vols["line_tiles"][i], vols["no_dma"][i]), f"level {i}: f1 tiles on lines / round 2's kernel"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,scale", [((2, 256, 5, 312, 312), 1.0), ((1, 256, 3, 136, 200), 1.0),
                                         ((1, 256, 4, 312, 312), 2e-3), ((1, 256, 2, 180, 180), 2e-3),
                                         ((1, 256, 3, 80, 752), 30.0)])
def test_stored_level2_is_the_twice_pooled_stored_level0(R, dtype, shape, scale):
    """On real-valued data the MFMA build cannot be compared with the naive kernel bit for bit (the fp32 sums are
    added in another order), but its pooling can be checked exactly against its OWN level 0: level 2 must be
    avg_pool2d's arithmetic applied twice -- the fp32 mean of two stored values, rounded once to the storage type
    (core/corr.py:42 on a 16-bit volume) -- whichever instructions the epilogue uses for it (round 3: mixed-precision
    FMAs for fp16).  The small scale puts values into fp16's subnormal range, the large one near its top."""
    B, D, H, W1, W2 = shape
    gen = torch.Generator(device="cuda").manual_seed(11)
    f1 = (torch.randn((B, D, H, W1), generator=gen, device="cuda") * scale).to(dtype)
    f2 = (torch.randn((B, D, H, W2), generator=gen, device="cuda") * scale).to(dtype)
    blk = R.CorrBlock1DHip(f1, f2, num_levels=4, radius=4)
    lvl = blk.dense_level(0)
    assert torch.isfinite(lvl.float()).all()
    for i in (1, 2):
        n = lvl.shape[-1] // 2
        lvl = ((lvl[..., 0:2 * n:2].float() + lvl[..., 1:2 * n:2].float()) * 0.5).to(dtype)
    got = blk.dense_level(2)
    assert torch.equal(got, lvl), f"{int((got != lvl).sum())} of {lvl.numel()} level-2 entries differ"


# ------------------------------------------------------------------ 16-bit against the rounded-input oracle
def _round(a, dtype):
    return torch.from_numpy(a).to(dtype).float().numpy()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", ["d256_b1_h4_w128", "odd_b1_d64_h2_w40x45", "wide_b1_d256_h2_w72x200"])
def test_16bit_volume_and_lookup(R, golden, case, dtype):
    g = golden(case)
    L, r = int(g["L"]), int(g["r"])
    f1r, f2r = _round(g["f1"], dtype), _round(g["f2"], dtype)
    pyr = O.pool_pyramid(O.corr_volume(f1r, f2r), L)
    blk = R.CorrBlock1DHip(dev(g["f1"], dtype), dev(g["f2"], dtype), num_levels=L, radius=r)
    assert blk.volume.dtype == dtype
    for i in range(L):
        assert rel(blk.dense_level(i), pyr[i]) < TOL16[dtype]
    for kind in KINDS:
        out = blk(dev(g[f"coords_{kind}"]))
        assert out.dtype == torch.float32
        assert rel(out, O.lookup_exact(pyr, g[f"coords_{kind}"], r)) < TOL16[dtype]
        alt = R.AltCorrBlock1DHip(dev(g["f1"], dtype), dev(g["f2"], dtype), num_levels=L, radius=r)(
            dev(g[f"coords_{kind}"]))
        assert rel(alt, O.lookup_exact(pyr, g[f"coords_{kind}"], r)) < TOL16[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_16bit_backward(R, golden, dtype):
    g = golden("d256_b1_h4_w128")
    L, r = int(g["L"]), int(g["r"])
    f1r, f2r = _round(g["f1"], dtype), _round(g["f2"], dtype)
    widths = [g[f"level{i}"].shape[-1] for i in range(L)]
    acc = None
    for k in KINDS:
        lg = O.lookup_bwd(g[f"coords_{k}"], g[f"gout_{k}"], widths, r, O.EXACT)
        acc = lg if acc is None else [a + b for a, b in zip(acc, lg)]
    rd1, rd2 = O.corr_volume_bwd(O.pool_pyramid_bwd(acc), f1r, f2r)
    f1 = dev(g["f1"], dtype).requires_grad_(True)
    f2 = dev(g["f2"], dtype).requires_grad_(True)
    blk = R.CorrBlock1DHip(f1, f2, num_levels=L, radius=r)
    sum((blk(dev(g[f"coords_{k}"])) * dev(g[f"gout_{k}"])).sum() for k in KINDS).backward()
    assert f1.grad.dtype == dtype
    assert rel(f1.grad, rd1) < TOL16[dtype] and rel(f2.grad, rd2) < TOL16[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(1, 32, 2, 20, 44), (2, 64, 1, 180, 180), (1, 16, 3, 12, 10), (1, 40, 2, 9, 14),
                                   (1, 24, 2, 8, 17), (2, 300, 1, 36, 28)])
def test_build_backward_rows_that_end_in_a_partial_piece(R, dtype, shape):
    """The contractions of the build's adjoint read feature rows in 16-byte pieces; a row that does not end on a
    piece boundary has its last piece read as the row's last 16 bytes and shifted into place (build_bwd.hip).
    Widths that are / are not multiples of the piece (8 elements of 16 bits, 4 of fp32), rows shorter than two
    pieces, odd widths (16-bit rows off the 4-byte grid: the element path), more channels than one workgroup takes:
    against the oracle on the rounded inputs."""
    B, D, H, W1, W2 = shape
    L, r = 2, 2
    rng = np.random.default_rng(W1 * 100 + W2)
    f1n = rng.standard_normal((B, D, H, W1)).astype(np.float32)
    f2n = rng.standard_normal((B, D, H, W2)).astype(np.float32)
    coords = O.synthetic_coords(B, H, W1, seed=3)
    coords[:, 0] *= W2 / W1
    gout = rng.standard_normal((B, L * (2 * r + 1), H, W1)).astype(np.float32)
    f1r, f2r = (f1n, f2n) if dtype == torch.float32 else (_round(f1n, dtype), _round(f2n, dtype))
    widths = [W2 >> i for i in range(L)]
    lg = O.lookup_bwd(coords, gout, widths, r, O.EXACT)
    rd1, rd2 = O.corr_volume_bwd(O.pool_pyramid_bwd(lg), f1r, f2r)
    f1 = dev(f1n, dtype).requires_grad_(True)
    f2 = dev(f2n, dtype).requires_grad_(True)
    blk = R.CorrBlock1DHip(f1, f2, num_levels=L, radius=r, coord_mode="exact")
    (blk(dev(coords)) * dev(gout)).sum().backward()
    tol = TOL32 if dtype == torch.float32 else TOL16[dtype]
    assert rel(f1.grad, rd1) < tol and rel(f2.grad, rd2) < tol
    assert f1.grad.abs().max() > 0 and f2.grad.abs().max() > 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mode", ["exact", "gridsample"])
@pytest.mark.parametrize("shape,L,r", [((2, 64, 3, 45, 45), 4, 4), ((1, 32, 2, 131, 70), 3, 2), ((1, 32, 5, 64, 257), 2, 7),
                                       ((8, 256, 96, 312, 312), 4, 4)])
def test_lookup_in_the_volumes_type_is_the_fp32_lookup_rounded_once(R, dtype, mode, shape, L, r):
    """rc_lookup_fwd_vt (what CorrBlockFast1D returns under mixed precision: the sampler's output has the volume's
    options, core/corr.py:44-51, sampler_kernel.cu:122-124): bit-equal to rc_lookup_fwd's fp32 taps rounded to
    nearest even, at ragged sizes and at the KITTI size; the gradient path takes the 16-bit gradients it gets."""
    B, D, H, W1, W2 = shape
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + W1)
    f1 = torch.randn((B, D, H, W1), generator=g, device="cuda").to(dtype)
    f2 = torch.randn((B, D, H, W2), generator=g, device="cuda").to(dtype)
    coords = dev(O.synthetic_coords(B, H, W1, seed=5))
    coords[:, 0] = coords[:, 0] * (W2 / W1) + torch.rand(coords[:, 0].shape, generator=g, device="cuda") * 9 - 4
    coords[0, 0, 0, :8] = torch.tensor([-40., -1.5, -0.25, 0., W2 - 1.0, W2 - 0.5, W2 + 3.25, 1e6], device="cuda")[:min(8, W1)]
    ref = R.CorrBlock1DHip(f1, f2, num_levels=L, radius=r, coord_mode=mode)
    blk = R.CorrBlock1DHip(f1, f2, num_levels=L, radius=r, coord_mode=mode, out_dtype=dtype)
    out, out32 = blk(coords), ref(coords)
    assert out.dtype == dtype and out32.dtype == torch.float32 and out.shape == out32.shape
    assert torch.equal(out, out32.to(dtype))
    assert out32.abs().max() > 0
    with pytest.raises(TypeError):
        R.CorrBlock1DHip(f1, f2, num_levels=L, radius=r, out_dtype=torch.float64)
    if B * H * W1 > 10000:
        return
    # gradients: the same feature gradients as the fp32-output block given the same (16-bit representable) gradient
    gout = torch.randn(out.shape, generator=g, device="cuda").to(dtype)
    grads = []
    for od in (dtype, None):
        a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
        o = R.CorrBlock1DHip(a, b, num_levels=L, radius=r, coord_mode=mode, out_dtype=od)(coords)
        (o * gout.to(o.dtype)).sum().backward()
        grads.append((a.grad, b.grad))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


def test_lookup_in_the_volumes_type_raw_abi_and_streamed_stores(R):
    """Through the C ABI: an fp32 pyramid's rc_lookup_fwd_vt is rc_lookup_fwd; an output beyond the
    non-temporal threshold (160 MB: 80 KITTI pairs of 16-bit taps) equals the fp32 one rounded."""
    from raft_stereo_amd._lib import lib
    g = torch.Generator(device="cuda").manual_seed(3)
    f1 = torch.randn((1, 64, 3, 50), generator=g, device="cuda")
    f2 = torch.randn((1, 64, 3, 50), generator=g, device="cuda")
    blk = R.CorrBlock1DHip(f1, f2, num_levels=4, radius=4)
    coords = dev(O.synthetic_coords(1, 3, 50, seed=2))
    a, b = torch.empty(blk._out_shape, device="cuda"), torch.empty(blk._out_shape, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for fn, o in ((lib.rc_lookup_fwd, a), (lib.rc_lookup_fwd_vt, b)):
        assert fn(blk._vol_ptr, blk._lay_ref, coords.data_ptr(), 2, o.data_ptr(), 4, blk.coord_mode, st) == 0
    assert torch.equal(a, b) and a.abs().max() > 0
    assert lib.rc_lookup_fwd_vt(blk._vol_ptr, blk._lay_ref, coords.data_ptr(), 2, None, 4, blk.coord_mode, st) != 0
    B, D, H, W = 80, 32, 96, 312
    h1 = torch.randn((B, D, H, W), generator=g, device="cuda").half()
    h2 = torch.randn((B, D, H, W), generator=g, device="cuda").half()
    c = dev(O.synthetic_coords(B, H, W, seed=4))
    o16 = R.CorrBlock1DHip(h1, h2, out_dtype=torch.float16)(c)
    assert o16.numel() * 2 > 160 << 20
    o32 = R.CorrBlock1DHip(h1, h2)(c)
    assert torch.equal(o16, o32.half())


# ------------------------------------------------------------------ corr_sampler twin
@pytest.mark.parametrize("case", ["tiny_b2_d32_h3_w24", "odd_b1_d64_h2_w40x45"])
@pytest.mark.parametrize("kind", KINDS)
def test_sampler_twin_forward_backward(R, golden, case, kind):
    g = golden(case)
    r = int(g["r"])
    for i in range(2):
        vol = dev(g[f"level{i}"]).requires_grad_(True)
        coords = dev((g[f"coords_{kind}"] / 2 ** i).astype(np.float32))
        out = R.CorrSamplerHip.apply(vol, coords, r)
        ref = O.lookup_exact([g[f"level{i}"]], (g[f"coords_{kind}"] / 2 ** i).astype(np.float32), r)
        assert tuple(out.shape) == ref.shape
        assert rel(out, ref) < 2e-6
        gout = g[f"gout_{kind}"][:, :2 * r + 1]
        out.backward(dev(gout))
        rg = O.lookup_bwd((g[f"coords_{kind}"] / 2 ** i).astype(np.float32), gout, [vol.shape[-1]], r, O.EXACT)[0]
        assert rel(vol.grad, rg) < 2e-6
    with pytest.raises(RuntimeError):
        R.corr_sampler_hip.forward(vol.detach().cpu(), coords, r)
    with pytest.raises(RuntimeError):
        R.corr_sampler_hip.forward(vol.detach().transpose(2, 3), coords, r)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", ["tiny_b2_d32_h3_w24", "odd_b1_d64_h2_w40x45"])
@pytest.mark.parametrize("kind", KINDS)
def test_sampler_twin_16bit_rounds_like_scalar_t(R, golden, case, kind, dtype):
    """The reference's kernels do their arithmetic in scalar_t (sampler_kernel.cu:50-56, :94-102): under AMP
    the lerp weights, every product and every `+=` are rounded to fp16.  The twin must reproduce those
    roundings bit for bit (fp32 products of two 11-bit significands are exact, so rounding them once is what
    the half multiply does).  bf16 has no dispatch in the reference (:126); the twin extends the same rule."""
    g = golden(case)
    r = int(g["r"])
    rnd = (lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)) if dtype == torch.float16 \
        else O.round_bf16
    for i in range(2):
        vol16 = dev(g[f"level{i}"], dtype)
        coords = (g[f"coords_{kind}"] / 2 ** i).astype(np.float32)
        out, = R.corr_sampler_hip.forward(vol16, dev(coords), r)
        assert out.dtype == dtype
        ref = O.sampler_fwd_rounded(vol16.float().cpu().numpy(), coords, r, rnd)
        assert np.array_equal(out.float().cpu().numpy(), ref), float(np.abs(out.float().cpu().numpy() - ref).max())
        gout16 = dev(g[f"gout_{kind}"][:, :2 * r + 1], dtype)
        gv, = R.corr_sampler_hip.backward(vol16, dev(coords), gout16, r)
        rg = O.sampler_bwd_rounded(coords, gout16.float().cpu().numpy(), vol16.shape[-1], r, rnd)
        assert np.array_equal(gv.float().cpu().numpy(), rg), float(np.abs(gv.float().cpu().numpy() - rg).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,r", [((2, 3, 7, 1040), 4), ((3, 5, 5, 312), 7), ((1, 2, 9, 72), 1),
                                      ((2, 3, 7, 44), 3)])
def test_sampler_twin_long_rows_and_waves_across_images(R, shape, r, dtype):
    """The backward's vector kernel: rows longer than one wave turn (W2 = 1040: 130 or 260 pieces), waves of 8
    pixel rows that straddle two images (H*W1 = 21, 25), the smallest and the largest radius, coordinates that
    put windows partly and wholly outside the row; W2 = 44 takes the element-store kernel for 16-bit types.
    Both directions against the scalar_t-rounding oracle, bit for bit for the 16-bit types."""
    B, H, W1, W2 = shape
    rng = np.random.default_rng(1000 + W2 + r)
    vol = dev(rng.standard_normal(shape).astype(np.float32), dtype)
    coords = np.zeros((B, 2, H, W1), np.float32)
    coords[:, 0] = rng.uniform(-2 * r - 3, W2 + 2 * r + 3, size=(B, H, W1)).astype(np.float32)
    coords[:, 0, 0, 0] = 0.0            # integer coordinates: dx = 0
    coords[:, 0, -1, -1] = W2 - 1.0
    gout = dev(rng.standard_normal((B, 2 * r + 1, H, W1)).astype(np.float32), dtype)
    if dtype == torch.float32:
        rnd = lambda a: np.asarray(a, np.float32)
    elif dtype == torch.float16:
        rnd = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)
    else:
        rnd = O.round_bf16
    # fp32: a product and the following sum may be one fused multiply-add (in the reference's kernel as well)
    same = (lambda a, b: np.allclose(a, b, rtol=1e-6, atol=1e-6)) if dtype == torch.float32 else np.array_equal
    out, = R.corr_sampler_hip.forward(vol, dev(coords), r)
    ref = O.sampler_fwd_rounded(vol.float().cpu().numpy(), coords, r, rnd)
    assert same(out.float().cpu().numpy(), ref)
    gv, = R.corr_sampler_hip.backward(vol, dev(coords), gout, r)
    rg = O.sampler_bwd_rounded(coords, gout.float().cpu().numpy(), W2, r, rnd)
    assert same(gv.float().cpu().numpy(), rg)


def test_sampler_twin_argument_checks(R):
    """sampler.cpp:20-22 CHECK_INPUT and the accessor types (packed_accessor32<float,4> for coords) raise on a
    wrong dtype or shape; the twin must not launch on them either."""
    vol = torch.zeros((1, 2, 8, 8), device="cuda", dtype=torch.float16)
    coords = torch.zeros((1, 2, 2, 8), device="cuda")
    g = torch.zeros((1, 9, 2, 8), device="cuda", dtype=torch.float16)
    for bad in (lambda: R.corr_sampler_hip.backward(vol, coords.half(), g, 4),
                lambda: R.corr_sampler_hip.backward(vol, coords, g[:, :7].contiguous(), 4),
                lambda: R.corr_sampler_hip.backward(vol, coords[:, :, :1].contiguous(), g, 4),
                lambda: R.corr_sampler_hip.backward(vol, coords, g.float(), 4),
                lambda: R.corr_sampler_hip.forward(vol, coords.double(), 4),
                lambda: R.corr_sampler_hip.forward(vol, coords, 8)):
        with pytest.raises(RuntimeError):
            bad()


# ------------------------------------------------------------------ full-size properties (BASELINE configs)
@pytest.mark.parametrize("dtype,shape", [(torch.float16, (8, 256, 96, 312)), (torch.float32, (1, 256, 96, 128)),
                                         (torch.float16, (1, 256, 60, 80)), (torch.bfloat16, (8, 256, 80, 180))])
def test_full_size_properties(R, dtype, shape):
    B, D, H, W = shape
    gen = torch.Generator(device="cuda").manual_seed(1234)
    f1 = torch.randn(shape, generator=gen, device="cuda").to(dtype)
    f2 = torch.randn(shape, generator=gen, device="cuda").to(dtype)
    blk = R.CorrBlock1DHip(f1, f2, num_levels=4, radius=4, coord_mode="exact")
    # (a) one image row against the oracle
    b, h = B - 1, H // 3
    pyr = O.pool_pyramid(O.corr_volume(f1[b:b + 1, :, h:h + 1].float().cpu().numpy(),
                                       f2[b:b + 1, :, h:h + 1].float().cpu().numpy()), 4)
    tol = TOL32 if dtype == torch.float32 else TOL16[dtype]
    for i in range(4):
        assert rel(blk.dense_level(i)[b:b + 1, h:h + 1], pyr[i]) < tol
    # (b) integer coordinates: the centre tap of level 0 is the volume entry itself (bit exact),
    #     and pooled levels at even coordinates likewise
    xs = torch.arange(W, device="cuda", dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
    disp = torch.randint(0, W // 3, (B, 1, H, W), generator=gen, device="cuda").float()
    coords = torch.cat([xs - disp, torch.zeros_like(xs)], 1).contiguous()
    out = blk(coords)
    lvl0 = blk.dense_level(0).float()
    idx = (xs - disp).long().clamp(0, W - 1)[:, 0]
    centre = torch.gather(lvl0, 3, idx.unsqueeze(-1))[..., 0]
    ok = ((xs - disp) >= 0)[:, 0]
    assert torch.equal(out[:, 4][ok], centre[ok])
    #     levels 1 and 3 are not in memory: the lookup derives them from levels 0 and 2 in registers, and at
    #     x divisible by 2^i the centre tap of level i must be the pooled entry bit for bit (fp32 mean of the
    #     stored values, rounded to the storage type -- avg_pool2d on a 16-bit volume, core/corr.py:42)
    assert [bool(s) for s in blk.layout.stored[:4]] == [True, False, True, False]
    for i in range(1, 4):
        lvl = blk.dense_level(i).float()
        xi = (xs - disp)[:, 0]
        sel = ok & (xi.long() % (1 << i) == 0) & ((xi.long() >> i) < lvl.shape[-1])
        assert int(sel.sum()) > 100
        ci = torch.gather(lvl, 3, (xi.long() >> i).clamp(0, lvl.shape[-1] - 1).unsqueeze(-1))[..., 0]
        assert torch.equal(out[:, 9 * i + 4][sel], ci[sel]), f"level {i}"
    # (c) linearity of the lookup in the volume and agreement of alt with reg on device
    c2 = torch.from_numpy(O.synthetic_coords(B, H, W, seed=5)).cuda()
    reg = blk(c2)
    alt = R.AltCorrBlock1DHip(f1, f2, num_levels=4, radius=4, coord_mode="exact")(c2)
    assert rel(alt, reg.cpu().numpy()) < (2e-5 if dtype == torch.float32 else tol)
    # (d) the adjoint identity <lookup(V), G> == <V, lookup^T(G)> through the build's gradient
    if dtype != torch.float32 or B * H * W < 20000:
        f1g = f1.clone().requires_grad_(True)
        blk2 = R.CorrBlock1DHip(f1g, f2, num_levels=4, radius=4)
        G = torch.randn(reg.shape, generator=gen, device="cuda")
        (blk2(c2) * G).sum().backward()
        eps_dir = torch.randn(f1.shape, generator=gen, device="cuda")
        lhs = float((f1g.grad.float() * eps_dir).sum())
        # correlation is linear in f1: <d/df1, E> = <lookup(C(E, f2)), G>
        rhs = float((R.CorrBlock1DHip(eps_dir.to(dtype), f2, num_levels=4, radius=4)(c2) * G).sum())
        assert abs(lhs - rhs) <= 2e-2 * max(abs(lhs), abs(rhs), 1.0)


def test_middlebury_full_size_alt_and_row_slabs(R):
    """BASELINE config 5: one Middlebury-F pair (2016x3008 padded, /4 -> [1,256,504,752] fp16) through `alt_hip`
    (core/corr.py:64-107, volume never built): (a) image rows against the oracle, (b) far-out-of-range
    coordinates give zeros, (c) the north star's row-slab claim: the 8 x 63-row split of shard_rows(504, 8, k)
    reproduces the full frame's rows bit for bit with no exchange between slabs, (d) the radius limit the
    header declares is enforced."""
    from raft_stereo_amd import _lib
    B, D, H, W, L, r = 1, 256, 504, 752, 4, 4
    gen = torch.Generator(device="cuda").manual_seed(55)
    f1 = torch.randn((B, D, H, W), generator=gen, device="cuda").half()
    f2 = torch.randn((B, D, H, W), generator=gen, device="cuda").half()
    coords_np = O.synthetic_coords(B, H, W, seed=11)
    coords = torch.from_numpy(coords_np).cuda()
    alt = R.make_corr_fn("alt_hip", f1, f2, radius=r, num_levels=L)
    assert alt.volume is None
    full = alt(coords)
    assert tuple(full.shape) == (B, L * (2 * r + 1), H, W) and full.dtype == torch.float32
    # (a) three image rows against the fp32 oracle on the rounded inputs (fp16 bar of SURVEY 8d)
    for h in (0, 251, 503):
        a, b = f1[:, :, h:h + 1].float().cpu().numpy(), f2[:, :, h:h + 1].float().cpu().numpy()
        ref = O.lookup_exact(O.pool_pyramid(O.corr_volume(a, b), L), coords_np[:, :, h:h + 1], r)
        assert rel(full[:, :, h:h + 1], ref) < TOL16[torch.float16], f"row {h}"
    # (b) nothing in reach -> zeros (grid_sample's zero padding / within_bounds of sampler_kernel.cu:49)
    far = coords.clone()
    far[:, 0] = 1.0e6
    far[:, 0, :, ::2] = -1.0e6
    assert float(alt(far).abs().max()) == 0.0
    # (c) row slabs: each rank's slab of both feature maps and of coords, nothing shared
    for k in range(8):
        r0, r1 = R.shard_rows(H, 8, k)
        assert r1 - r0 == 63
        slab = R.make_corr_fn("alt_hip", f1[:, :, r0:r1], f2[:, :, r0:r1], radius=r, num_levels=L)
        assert torch.equal(slab(coords[:, :, r0:r1].contiguous()), full[:, :, r0:r1]), f"slab {k}"
    # (d)
    with pytest.raises(ValueError):
        R.AltCorrBlock1DHip(f1[:, :, :2], f2[:, :, :2], num_levels=L, radius=_lib.RC_ALT_MAX_RADIUS + 1)


# ------------------------------------------------------------------ fused adjoint vs the pyramid adjoint
@pytest.mark.parametrize("dtype,mode", [(torch.float32, "gridsample"), (torch.float32, "exact"),
                                        (torch.float16, "exact"), (torch.bfloat16, "exact")])
@pytest.mark.parametrize("shape", [(2, 64, 3, 40, 45, 4, 4), (1, 256, 2, 180, 180, 4, 4), (1, 32, 2, 33, 20, 2, 3),
                                   (1, 64, 1, 312, 312, 4, 4), (1, 32, 1, 760, 752, 4, 4)])
def test_fused_adjoint_equals_pyramid_adjoint(R, dtype, mode, shape):
    """rc_lookup_bwd_dense + rc_build_bwd_dense against rc_lookup_bwd (per lookup) + rc_build_bwd, both
    through the C-ABI, on coordinates that include out-of-range and edge values."""
    import ctypes
    from raft_stereo_amd._lib import lib, check
    from raft_stereo_amd.corr import COORD_MODES, _DT
    B, D, H, W1, W2, L, r = shape
    gen = torch.Generator(device="cuda").manual_seed(3)
    f1 = torch.randn((B, D, H, W1), generator=gen, device="cuda").to(dtype)
    f2 = torch.randn((B, D, H, W2), generator=gen, device="cuda").to(dtype)
    blk = R.CorrBlock1DHip(f1, f2, num_levels=L, radius=r, coord_mode=mode)
    n = 5
    st = torch.cuda.current_stream().cuda_stream
    coords, gouts = [], []
    for t in range(n):
        x = torch.rand((B, 1, H, W1), generator=gen, device="cuda") * (W2 + 24) - 12
        if t == 0:
            x = torch.round(x)                          # integer-exact positions
        coords.append(torch.cat([x, torch.zeros_like(x)], 1).contiguous())
        gouts.append(torch.randn((B, L * (2 * r + 1), H, W1), generator=gen, device="cuda"))
    lay = blk.grad_layout
    # pyramid path
    gvol = torch.zeros(lay.total, dtype=torch.float32, device="cuda")
    for c, g in zip(coords, gouts):
        check(lib.rc_lookup_bwd(c.data_ptr(), 2, g.data_ptr(), gvol.data_ptr(), lay, r, COORD_MODES[mode], st), "bwd")
    work = torch.empty(int(lib.rc_build_bwd_work_bytes(lay, _DT[dtype])), dtype=torch.uint8, device="cuda")
    a1, a2 = torch.empty_like(f1), torch.empty_like(f2)
    check(lib.rc_build_bwd(gvol.data_ptr(), lay, f1.data_ptr(), f2.data_ptr(), a1.data_ptr(), a2.data_ptr(), D,
                           _DT[dtype], work.data_ptr(), st), "build_bwd")
    # fused path
    ld = int(lib.rc_dense_grad_ld(lay))
    dense = torch.full((B * H * W1 * ld,), float("nan"), dtype=dtype, device="cuda")
    cl = (ctypes.c_void_p * n)(*[c.data_ptr() for c in coords])
    gl = (ctypes.c_void_p * n)(*[g.data_ptr() for g in gouts])
    check(lib.rc_lookup_bwd_dense(cl, gl, n, 2, dense.data_ptr(), lay, D, _DT[dtype], r, COORD_MODES[mode], st), "dense")
    b1, b2 = torch.empty_like(f1), torch.empty_like(f2)
    check(lib.rc_build_bwd_dense(dense.data_ptr(), lay, f1.data_ptr(), f2.data_ptr(), b1.data_ptr(), b2.data_ptr(), D,
                                 _DT[dtype], st), "build_bwd_dense")
    # the dense gradient equals what the unskew step left in `work` up to summation order / one rounding
    ref_dense = work.view(dtype)[:dense.numel()]
    tol = TOL32 if dtype == torch.float32 else TOL16[dtype]
    assert rel(dense.float(), ref_dense.float().cpu().numpy()) < tol
    assert rel(b1.float(), a1.float().cpu().numpy()) < tol and rel(b2.float(), a2.float().cpu().numpy()) < tol


def test_more_lookups_than_one_fused_call(R):
    """33 lookups in one graph take the pyramid route; the result equals 33 x one lookup's gradient."""
    gen = torch.Generator(device="cuda").manual_seed(4)
    f1 = torch.randn((1, 32, 2, 48), generator=gen, device="cuda").requires_grad_(True)
    f2 = torch.randn((1, 32, 2, 48), generator=gen, device="cuda").requires_grad_(True)
    c = torch.from_numpy(O.synthetic_coords(1, 2, 48, seed=2)).cuda()
    G = torch.randn((1, 36, 2, 48), generator=gen, device="cuda")
    blk = R.CorrBlock1DHip(f1, f2)
    sum((blk(c) * G).sum() for _ in range(33)).backward()
    many1, many2 = f1.grad.clone(), f2.grad.clone()
    f1.grad = f2.grad = None
    blk = R.CorrBlock1DHip(f1, f2)
    (blk(c) * G).sum().backward()
    assert rel(many1, (33 * f1.grad).cpu().numpy()) < 1e-5 and rel(many2, (33 * f2.grad).cpu().numpy()) < 1e-5


def test_alt_more_lookups_than_one_fused_call(R):
    """alt_hip has no pyramid: 70 lookups go through three fused calls whose gradients are summed."""
    gen = torch.Generator(device="cuda").manual_seed(5)
    f1 = torch.randn((1, 32, 2, 48), generator=gen, device="cuda").requires_grad_(True)
    f2 = torch.randn((1, 32, 2, 48), generator=gen, device="cuda").requires_grad_(True)
    c = torch.from_numpy(O.synthetic_coords(1, 2, 48, seed=3)).cuda()
    G = torch.randn((1, 36, 2, 48), generator=gen, device="cuda")
    blk = R.AltCorrBlock1DHip(f1, f2)
    sum((blk(c) * G).sum() for _ in range(70)).backward()
    many1, many2 = f1.grad.clone(), f2.grad.clone()
    f1.grad = f2.grad = None
    blk = R.CorrBlock1DHip(f1, f2)
    (blk(c) * G).sum().backward()
    assert rel(many1, (70 * f1.grad).cpu().numpy()) < 1e-5 and rel(many2, (70 * f2.grad).cpu().numpy()) < 1e-5


# ------------------------------------------------------------------ every compiled radius / level count
@pytest.mark.parametrize("L,r", [(1, 1), (1, 8), (2, 2), (3, 5), (3, 6), (4, 7), (4, 8), (2, 4), (4, 1)])
def test_every_radius_and_level_count_against_the_oracle(R, L, r):
    """The golden vectors hold (4,4) and (2,3); the oracle (pinned by them) covers the other template
    instances: reg forward in both coordinate modes, fp32 and fp16, alt, the fused adjoint, convc1."""
    rng = np.random.default_rng(100 * L + r)
    B, D, H, W1, W2 = 1, 32, 2, 40, 70
    f1 = rng.standard_normal((B, D, H, W1)).astype(np.float32)
    f2 = rng.standard_normal((B, D, H, W2)).astype(np.float32)
    x = rng.uniform(-r - 2, W2 + r + 2, size=(B, 1, H, W1)).astype(np.float32)
    coords = np.concatenate([x, np.zeros_like(x)], axis=1)
    K = L * (2 * r + 1)
    G = rng.standard_normal((B, K, H, W1)).astype(np.float32)
    pyr = O.pool_pyramid(O.corr_volume(f1, f2), L)
    widths = [p.shape[-1] for p in pyr]
    for mode, name in ((O.GRIDSAMPLE, "gridsample"), (O.EXACT, "exact")):
        ref = O.lookup(pyr, coords, r, mode)
        t1, t2 = dev(f1).requires_grad_(True), dev(f2).requires_grad_(True)
        blk = R.CorrBlock1DHip(t1, t2, num_levels=L, radius=r, coord_mode=name)
        out = blk(dev(coords))
        assert rel(out, ref) < TOL32
        (out * dev(G)).sum().backward()
        d1, d2 = O.corr_volume_bwd(O.pool_pyramid_bwd(O.lookup_bwd(coords, G, widths, r, mode)), f1, f2)
        assert rel(t1.grad, d1) < TOL32 and rel(t2.grad, d2) < TOL32
        if r <= 6:
            alt = R.AltCorrBlock1DHip(dev(f1), dev(f2), num_levels=L, radius=r, coord_mode=name)(dev(coords))
            assert rel(alt, ref) < TOL32
        if r <= 7:
            w = rng.standard_normal((64, K)).astype(np.float32)
            bvec = rng.standard_normal(64).astype(np.float32)
            with torch.no_grad():
                act = R.CorrBlock1DHip(dev(f1), dev(f2), num_levels=L, radius=r, coord_mode=name).lookup_convc1(
                    dev(coords), dev(w), dev(bvec))
            assert rel(act, O.convc1_relu(ref, w, bvec)) < TOL32
    h1, h2 = f1.astype(np.float16).astype(np.float32), f2.astype(np.float16).astype(np.float32)
    rnd = lambda a: a.astype(np.float16).astype(np.float32)
    ref16 = O.lookup_exact(O.pool_pyramid(O.corr_volume(h1, h2), L, round_to=rnd), coords, r)
    out16 = R.CorrBlock1DHip(dev(h1, torch.float16), dev(h2, torch.float16), num_levels=L, radius=r)(dev(coords))
    assert rel(out16, ref16) < TOL16[torch.float16]


def _random_geometries(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L = int(rng.integers(1, 5))
        W2 = int(rng.integers(max(2 << (L - 1), 3), 150))
        out.append((int(rng.integers(1, 4)), int(rng.choice([1, 7, 24, 32, 50, 64, 100, 256])), int(rng.integers(1, 5)),
                    int(rng.integers(1, 140)), W2, L, int(rng.integers(1, 9))))
    return out


@pytest.mark.parametrize("geom", _random_geometries(14, seed=2025) +
                         [(1, 32, 1, 1, 16, 4, 4), (2, 7, 2, 5, 3, 1, 1), (1, 64, 1, 130, 17, 4, 8), (1, 256, 1, 64, 16, 4, 4)])
def test_ragged_geometries_against_the_oracle(R, geom):
    """Seeded random shapes: widths that are no multiple of anything, W1 != W2, single pixels and
    rows, feature depths that miss the MFMA chunk, every level count and radius."""
    B, D, H, W1, W2, L, r = geom
    rng = np.random.default_rng(hash(geom) % (1 << 31))
    f1 = rng.standard_normal((B, D, H, W1)).astype(np.float32)
    f2 = rng.standard_normal((B, D, H, W2)).astype(np.float32)
    x = rng.uniform(-r - 3, W2 + r + 3, size=(B, 1, H, W1)).astype(np.float32)
    coords = np.concatenate([x, np.zeros_like(x)], axis=1)
    G = rng.standard_normal((B, L * (2 * r + 1), H, W1)).astype(np.float32)
    pyr = O.pool_pyramid(O.corr_volume(f1, f2), L)
    widths = [p.shape[-1] for p in pyr]
    mode = O.GRIDSAMPLE if widths[-1] >= 2 else O.EXACT          # grid_sample needs every level >= 2 wide
    name = "gridsample" if mode == O.GRIDSAMPLE else "exact"
    t1, t2 = dev(f1).requires_grad_(True), dev(f2).requires_grad_(True)
    blk = R.CorrBlock1DHip(t1, t2, num_levels=L, radius=r, coord_mode=name)
    for i in range(L):
        assert rel(blk.dense_level(i), pyr[i]) < TOL32
    out = blk(dev(coords))
    assert rel(out, O.lookup(pyr, coords, r, mode)) < TOL32
    (out * dev(G)).sum().backward()
    d1, d2 = O.corr_volume_bwd(O.pool_pyramid_bwd(O.lookup_bwd(coords, G, widths, r, mode)), f1, f2)
    assert rel(t1.grad, d1) < TOL32 and rel(t2.grad, d2) < TOL32
    if r <= 6:
        alt = R.AltCorrBlock1DHip(dev(f1), dev(f2), num_levels=L, radius=r, coord_mode=name)(dev(coords))
        assert rel(alt, O.lookup(pyr, coords, r, mode)) < TOL32
    h1, h2 = dev(f1, torch.bfloat16), dev(f2, torch.bfloat16)
    ref16 = O.lookup_exact(O.pool_pyramid(O.corr_volume(h1.float().cpu().numpy(), h2.float().cpu().numpy()), L,
                                          round_to=O.round_bf16), coords, r)
    assert rel(R.CorrBlock1DHip(h1, h2, num_levels=L, radius=r)(dev(coords)), ref16) < TOL16[torch.bfloat16]


# ------------------------------------------------------------------ fused lookup + convc1
@pytest.mark.parametrize("case", CONVC1_CASES)
@pytest.mark.parametrize("kind", KINDS)
def test_lookup_convc1_fp32_matches_reference_motion_encoder(R, golden, case, kind):
    g, c = golden(case), golden("convc1_" + case)
    L, r = int(g["L"]), int(g["r"])
    with torch.no_grad():
        blk = R.CorrBlock1DHip(dev(g["f1"]), dev(g["f2"]), num_levels=L, radius=r)
        act = blk.lookup_convc1(dev(g[f"coords_{kind}"]), dev(c["weight"]), dev(c["bias"]))
        pre = blk.lookup_convc1(dev(g[f"coords_{kind}"]), dev(c["weight"]), dev(c["bias"]), relu=False)
    assert act.dtype == torch.float32 and tuple(act.shape) == c[f"act_{kind}"].shape
    assert rel(act, c[f"act_{kind}"]) < TOL32
    assert rel(pre, O.convc1_relu(g[f"lookup_{kind}"], c["weight"], c["bias"], relu=False)) < TOL32


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", CONVC1_CASES)
def test_lookup_convc1_16bit(R, golden, case, dtype):
    """16-bit pyramid: the taps and the weight are rounded to the pyramid's type before the
    convolution (autocast), accumulation in fp32; compared with the oracle on the rounded inputs."""
    g, c = golden(case), golden("convc1_" + case)
    L, r = int(g["L"]), int(g["r"])
    rnd = (lambda a: a.astype(np.float16).astype(np.float32)) if dtype == torch.float16 else O.round_bf16
    f1, f2 = rnd(g["f1"]), rnd(g["f2"])
    pyr = O.pool_pyramid(O.corr_volume(f1, f2), L, round_to=rnd)
    for kind in KINDS:
        ref = O.convc1_relu(O.lookup_exact(pyr, g[f"coords_{kind}"], r), c["weight"], c["bias"], round_to=rnd)
        with torch.no_grad():
            blk = R.CorrBlock1DHip(dev(f1, dtype), dev(f2, dtype), num_levels=L, radius=r)
            act = blk.lookup_convc1(dev(g[f"coords_{kind}"]), dev(c["weight"]), dev(c["bias"]))
        assert act.dtype == dtype
        assert rel(act, ref) < 2 * TOL16[dtype]


@pytest.mark.parametrize("dtype,shape,mode", [(torch.float16, (8, 256, 96, 312), "exact"),
                                               (torch.float32, (1, 256, 96, 128), "gridsample"),
                                               (torch.bfloat16, (2, 128, 5, 333), "gridsample")])
def test_lookup_convc1_equals_unfused_at_full_size(R, dtype, shape, mode):
    """BASELINE sizes: fused == conv2d(lookup) computed with torch on the library's own lookup."""
    gen = torch.Generator(device="cuda").manual_seed(9)
    B, D, H, W = shape
    f1 = torch.randn(shape, generator=gen, device="cuda").to(dtype)
    f2 = torch.randn(shape, generator=gen, device="cuda").to(dtype)
    c = torch.from_numpy(O.synthetic_coords(B, H, W, seed=5)).cuda()
    wgt = torch.randn((64, 36, 1, 1), generator=gen, device="cuda") * 0.3
    bias = torch.randn((64,), generator=gen, device="cuda")
    with torch.no_grad():
        blk = R.CorrBlock1DHip(f1, f2, coord_mode=mode)
        corr = blk(c)
        if dtype != torch.float32:
            ref = torch.relu(torch.nn.functional.conv2d(corr.to(dtype).float(), wgt.to(dtype).float(), bias))
        else:
            ref = torch.relu(torch.nn.functional.conv2d(corr.double(), wgt.double(), bias.double())).float()
        act = blk.lookup_convc1(c, wgt, bias)
    tol = TOL32 if dtype == torch.float32 else TOL16[dtype]
    assert rel(act, ref.cpu().numpy()) < tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_lookups_are_the_same_in_either_workgroup_order(R, dtype):
    """A launch of at most 1024 workgroups gives each XCD a contiguous eighth of the pixels, a larger one deals
    workgroups round-robin (lookup.hip, fused.hip).  12 images at once (1 404 workgroups: plain order) must equal
    the same images four at a time (468 workgroups, not a multiple of 8: the other order), bit for bit, for the
    plain lookup, the lookup with the coordinate update and the fused lookup + convc1."""
    gen = torch.Generator(device="cuda").manual_seed(31)
    B, D, H, W = 12, 64, 96, 312
    f1 = torch.randn((B, D, H, W), generator=gen, device="cuda").to(dtype)
    f2 = torch.randn((B, D, H, W), generator=gen, device="cuda").to(dtype)
    c = torch.from_numpy(O.synthetic_coords(B, H, W, seed=8)).cuda()
    delta = torch.randn((B, 2, H, W), generator=gen, device="cuda")
    wgt = torch.randn((64, 36, 1, 1), generator=gen, device="cuda") * 0.3
    bias = torch.randn((64,), generator=gen, device="cuda")
    with torch.no_grad():
        whole = R.CorrBlock1DHip(f1, f2)
        out, act = whole(c), whole.lookup_convc1(c, wgt, bias)
        new, flow, out_u = whole.update_and_lookup(c, delta)
        for k in range(0, B, 4):
            sl = slice(k, k + 4)
            part = R.CorrBlock1DHip(f1[sl].contiguous(), f2[sl].contiguous())
            assert torch.equal(part(c[sl].contiguous()), out[sl])
            assert torch.equal(part.lookup_convc1(c[sl].contiguous(), wgt, bias), act[sl])
            n2, f2_, o2 = part.update_and_lookup(c[sl].contiguous(), delta[sl].contiguous())
            assert torch.equal(n2, new[sl]) and torch.equal(f2_, flow[sl]) and torch.equal(o2, out_u[sl])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("levels,radius", [(4, 1), (4, 2), (3, 3), (4, 5), (2, 6), (4, 7), (1, 4)])
def test_lookup_convc1_tile_path_radii_levels_partial_blocks(R, dtype, levels, radius):
    """The LDS-tile path of the fused kernel (16-bit outputs made of whole 16-byte pieces) at every count of MFMA
    steps (ceil((2r+1)/4) = 1..4) and level count, on images of 5.47 workgroups (H*W = 1400 = 8 * 175: the last
    workgroup of every image is part empty, one of its waves wholly), three images, with and without the ReLU:
    fused == conv2d(lookup) computed with torch on the library's own lookup."""
    gen = torch.Generator(device="cuda").manual_seed(100 + 10 * levels + radius)
    B, D, H, W = 3, 64, 7, 200
    f1 = torch.randn((B, D, H, W), generator=gen, device="cuda").to(dtype)
    f2 = torch.randn((B, D, H, W), generator=gen, device="cuda").to(dtype)
    c = torch.from_numpy(O.synthetic_coords(B, H, W, seed=7)).cuda()
    K = levels * (2 * radius + 1)
    wgt = torch.randn((64, K, 1, 1), generator=gen, device="cuda") * 0.3
    bias = torch.randn((64,), generator=gen, device="cuda")
    with torch.no_grad():
        blk = R.CorrBlock1DHip(f1, f2, num_levels=levels, radius=radius)
        lin = torch.nn.functional.conv2d(blk(c).to(dtype).float(), wgt.to(dtype).float(), bias)
        for relu in (True, False):
            act = blk.lookup_convc1(c, wgt, bias, relu=relu)
            assert act.dtype == dtype and tuple(act.shape) == (B, 64, H, W)
            ref = torch.relu(lin) if relu else lin
            assert rel(act, ref.cpu().numpy()) < TOL16[dtype]
        assert (blk.lookup_convc1(c, wgt, bias, relu=False) < 0).any()


def test_lookup_convc1_rejects_bad_weights(R):
    f = torch.randn(1, 32, 2, 32, device="cuda")
    blk = R.CorrBlock1DHip(f, f)
    c = torch.zeros(1, 2, 2, 32, device="cuda")
    with pytest.raises(ValueError):
        blk.lookup_convc1(c, torch.zeros(64, 35, device="cuda"), torch.zeros(64, device="cuda"))
    # parameters made under torch.inference_mode() have no version counter: accepted, same result
    g = torch.Generator(device="cuda").manual_seed(1)
    w, b = torch.randn(64, 36, generator=g, device="cuda"), torch.randn(64, generator=g, device="cuda")
    ref = blk.lookup_convc1(c, w, b)
    with torch.inference_mode():
        wi, bi = w.clone(), b.clone()
        out = blk.lookup_convc1(c, wi, bi)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("case", CONVC1_CASES)
@pytest.mark.parametrize("kind", KINDS)
def test_lookup_convc1_backward_fp32_matches_reference_autograd(R, golden, case, kind):
    """Training through the fused layer: weight / bias gradients against the reference's BasicMotionEncoder
    autograd, feature-map gradients against the two-step route (reg_hip lookup fed the reference's d corr),
    which is itself pinned on the reference's CorrBlock1D autograd."""
    g, c = golden(case), golden("convc1_" + case)
    L, r = int(g["L"]), int(g["r"])
    f1, f2 = dev(g["f1"]).requires_grad_(True), dev(g["f2"]).requires_grad_(True)
    w, b = dev(c["weight"]).requires_grad_(True), dev(c["bias"]).requires_grad_(True)
    coords = dev(g[f"coords_{kind}"])
    blk = R.CorrBlock1DHip(f1, f2, num_levels=L, radius=r)
    act = blk.lookup_convc1(coords, w, b)
    assert rel(act, c[f"act_{kind}"]) < TOL32
    act.backward(dev(c[f"gact_{kind}"]))
    assert rel(w.grad, c[f"d_weight_{kind}"]) < TOL32
    assert rel(b.grad, c[f"d_bias_{kind}"]) < TOL32
    g1, g2 = dev(g["f1"]).requires_grad_(True), dev(g["f2"]).requires_grad_(True)
    R.CorrBlock1DHip(g1, g2, num_levels=L, radius=r)(coords).backward(dev(c[f"d_corr_{kind}"]))
    assert rel(f1.grad, g1.grad.cpu().numpy()) < TOL32
    assert rel(f2.grad, g2.grad.cpu().numpy()) < TOL32


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_lookup_convc1_backward_equals_unfused_at_training_size(R, dtype):
    """SceneFlow training shape, three lookups accumulated, frozen feature maps in one run (only the layer's
    parameters train) and everything trainable in the other: against lookup -> torch conv -> relu."""
    torch.manual_seed(8)
    B, D, H, W = 8, 256, 80, 180
    f1 = torch.randn(B, D, H, W, device="cuda").to(dtype)
    f2 = torch.randn(B, D, H, W, device="cuda").to(dtype)
    w = (torch.randn(64, 36, 1, 1, device="cuda") * 0.2)
    b = torch.randn(64, device="cuda")
    coords = [dev(O.synthetic_coords(B, H, W, seed=s)) for s in (1, 2, 3)]
    gouts = [torch.randn(B, 64, H, W, device="cuda").to(dtype) for _ in coords]
    tol = 2e-4 if dtype == torch.float32 else TOL16[dtype]

    def run(fused, train_fmaps):
        a1, a2 = f1.clone().requires_grad_(train_fmaps), f2.clone().requires_grad_(train_fmaps)
        ww, bb = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        blk = R.CorrBlock1DHip(a1, a2)
        outs = []
        if fused:
            outs = [blk.lookup_convc1(c, ww, bb) for c in coords]
            torch.autograd.backward(outs, gouts)
        elif dtype == torch.float32:
            outs = [torch.relu(torch.nn.functional.conv2d(blk(c), ww, bb)) for c in coords]
            torch.autograd.backward(outs, gouts)
        else:
            # 16-bit: the same arithmetic spelled out in fp32 on the fused run's own ReLU masks (torch's
            # 16-bit convolution backward is too coarse to serve as a reference)
            ww.grad, bb.grad = torch.zeros_like(ww), torch.zeros_like(bb)
            corrs, dcorrs = [blk(c) for c in coords], []
            for corr, act, go in zip(corrs, fused_outs, gouts):
                gm = go.float() * (act > 0)
                ww.grad += torch.einsum("bohw,bkhw->ok", gm, corr.detach()).view_as(ww)
                bb.grad += gm.sum(dim=(0, 2, 3))
                outs.append(act)
                dcorrs.append(torch.einsum("ok,bohw->bkhw", ww.detach().view(64, 36), gm))
            if train_fmaps:
                torch.autograd.backward(corrs, dcorrs)
        return [t.grad for t in (ww, bb, a1, a2)], outs

    for train_fmaps in (False, True):
        mine, fused_outs = run(True, train_fmaps)
        ref, o2 = run(False, train_fmaps)
        for x, y in zip(fused_outs, o2):
            assert rel(x, y.detach().float().cpu().numpy()) < tol
        for name, x, y in zip(("dW", "db", "df1", "df2"), mine, ref):
            if y is None:
                assert x is None
                continue
            assert rel(x, y.float().cpu().numpy()) < tol, (name, train_fmaps)


# ------------------------------------------------------------------ coordinate glue in the lookup
@pytest.mark.parametrize("dtype,shape", [(torch.float32, (2, 64, 3, 45)), (torch.float16, (2, 256, 7, 312)),
                                          (torch.bfloat16, (1, 32, 2, 130))])
def test_update_and_lookup_is_bit_identical_to_the_separate_steps(R, dtype, shape):
    """core/raft_stereo.py:109-123 with torch ops + the library's lookup, against the one-kernel form."""
    gen = torch.Generator(device="cuda").manual_seed(21)
    B, D, H, W = shape
    f1 = torch.randn(shape, generator=gen, device="cuda").to(dtype)
    f2 = torch.randn(shape, generator=gen, device="cuda").to(dtype)
    coords1 = torch.from_numpy(O.synthetic_coords(B, H, W, seed=8)).cuda()
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    coords0 = torch.stack([xs, ys], dim=0).float()[None].repeat(B, 1, 1, 1)       # utils.coords_grid
    with torch.no_grad():
        blk = R.CorrBlock1DHip(f1, f2)
        for it in range(3):
            delta = torch.randn((B, 2, H, W), generator=gen, device="cuda") * 2.5
            ref_delta = delta.clone()
            ref_delta[:, 1] = 0.0
            ref_coords = coords1 + ref_delta
            ref_corr, ref_flow = blk(ref_coords), ref_coords - coords0
            new, flow, corr = blk.update_and_lookup(coords1, delta)
            assert torch.equal(new, ref_coords) and torch.equal(flow, ref_flow) and torch.equal(corr, ref_corr)
            coords1 = new
        # 1-channel coords (what CorrBlockFast1D passes, core/corr.py:48): y comes out as the row index
        new, flow, corr = blk.update_and_lookup(coords1[:, :1].contiguous(), delta)
        assert torch.equal(new[:, 1], coords0[:, 1]) and torch.equal(flow[:, 1], torch.zeros_like(flow[:, 1]))
        assert torch.equal(corr, blk(new))


@pytest.mark.parametrize("dtype,shape,ddt", [(torch.float32, (2, 64, 3, 45), torch.float32),
                                              (torch.float16, (2, 256, 7, 312), torch.float16),
                                              (torch.float16, (1, 256, 60, 80), torch.float32),
                                              (torch.bfloat16, (1, 32, 2, 130), torch.bfloat16)])
def test_update_lookup_convc1_is_bit_identical_to_the_separate_steps(R, dtype, shape, ddt):
    """core/raft_stereo.py:109-111,120-123 + core/update.py:78: one kernel (rc_lookup_convc1_fwd_update) against
    torch's coordinate arithmetic followed by the fused lookup + convc1.  The flow head's delta arrives in the
    autocast dtype and is promoted to fp32 by torch's `+`; the first iteration has no delta."""
    gen = torch.Generator(device="cuda").manual_seed(23)
    B, D, H, W = shape
    f1 = torch.randn(shape, generator=gen, device="cuda").to(dtype)
    f2 = torch.randn(shape, generator=gen, device="cuda").to(dtype)
    w = torch.randn((64, 36, 1, 1), generator=gen, device="cuda") * 0.2
    bias = torch.randn((64,), generator=gen, device="cuda")
    coords1 = torch.from_numpy(O.synthetic_coords(B, H, W, seed=9)).cuda()
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    coords0 = torch.stack([xs, ys], dim=0).float()[None].repeat(B, 1, 1, 1)
    with torch.no_grad():
        blk = R.CorrBlock1DHip(f1, f2)
        delta = None
        for it in range(3):
            if delta is None:
                ref_coords = coords1
            else:
                ref_delta = delta.clone()
                ref_delta[:, 1] = 0.0
                ref_coords = coords1 + ref_delta                 # fp32 + 16-bit promotes to fp32
            assert ref_coords.dtype == torch.float32
            ref_act, ref_flow = blk.lookup_convc1(ref_coords, w, bias), ref_coords - coords0
            new, flow, act = blk.update_lookup_convc1(coords1, delta, w, bias)
            assert torch.equal(new, ref_coords) and torch.equal(flow, ref_flow) and torch.equal(act, ref_act)
            coords1 = new
            delta = (torch.randn((B, 2, H, W), generator=gen, device="cuda") * 2.5).to(ddt)
        new, flow, act = blk.update_lookup_convc1(coords1[:, :1].contiguous(), delta, w, bias, relu=False)
        assert torch.equal(new[:, 1], coords0[:, 1]) and torch.equal(flow[:, 1], torch.zeros_like(flow[:, 1]))
        assert torch.equal(act, blk.lookup_convc1(new, w, bias, relu=False))
    with pytest.raises(RuntimeError):       # forward-only
        R.CorrBlock1DHip(f1.requires_grad_(True), f2).update_lookup_convc1(coords1, None, w, bias)


# ------------------------------------------------------------------ HIP graph capture
def test_build_and_lookups_replay_from_a_hip_graph(R):
    """The library neither allocates nor synchronises, so a caller can capture its GRU loop: the
    replayed graph gives the eager results (reg_hip with the fused update, alt_hip, fused convc1)."""
    gen = torch.Generator(device="cuda").manual_seed(31)
    f1 = torch.randn((1, 256, 12, 80), generator=gen, device="cuda").half()
    f2 = torch.randn((1, 256, 12, 80), generator=gen, device="cuda").half()
    coords = torch.from_numpy(O.synthetic_coords(1, 12, 80, seed=4)).cuda()
    delta = torch.randn((1, 2, 12, 80), generator=gen, device="cuda")
    w = torch.randn((64, 36), generator=gen, device="cuda") * 0.2
    bias = torch.randn((64,), generator=gen, device="cuda")

    def loop():
        reg, alt = R.CorrBlock1DHip(f1, f2), R.AltCorrBlock1DHip(f1, f2)
        c, outs = coords, []
        for _ in range(3):
            c, flow, corr = reg.update_and_lookup(c, delta)
            outs += [corr, flow, alt(c), reg.lookup_convc1(c, w, bias).float()]
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


def test_training_step_replays_from_a_hip_graph(R):
    """Forward (build, lookups, fused convc1) AND backward to the feature maps and the layer's parameters
    captured in one HIP graph: the replayed gradients equal the eager ones bit for bit.  The leaves are made
    fresh and their first passes run on the capture's side stream (torch's rule for whole-network capture)."""
    gen = torch.Generator(device="cuda").manual_seed(77)
    B, D, H, W = 2, 64, 6, 96
    coords = [torch.from_numpy(O.synthetic_coords(B, H, W, seed=s)).cuda() for s in (1, 2, 3)]
    gout = torch.randn((B, 36, H, W), generator=gen, device="cuda")
    gact = torch.randn((B, 64, H, W), generator=gen, device="cuda").bfloat16()
    init = [torch.randn((B, D, H, W), generator=gen, device="cuda").bfloat16() for _ in range(2)]
    init += [torch.randn((64, 36), generator=gen, device="cuda") * 0.2, torch.randn((64,), generator=gen, device="cuda")]

    def leaves():
        return [t.clone().requires_grad_(True) for t in init]

    def step(f1, f2, w, b):
        blk = R.CorrBlock1DHip(f1, f2)
        outs = [blk(c) for c in coords[:2]] + [blk.lookup_convc1(coords[2], w, b)]
        torch.autograd.backward(outs, [gout, gout, gact])

    ref = leaves()
    step(*ref)
    torch.cuda.synchronize()
    cap = leaves()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            for t in cap:
                t.grad = None
            step(*cap)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for t in cap:
        t.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(*cap)
    for t in cap:
        t.grad.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(cap, ref):
        assert a.grad is not None and torch.equal(a.grad, b.grad)
    assert float(cap[0].grad.float().abs().max()) > 0


# ------------------------------------------------------------------ argument checking
def test_argument_errors(R):
    f = torch.randn(1, 32, 2, 32, device="cuda")
    with pytest.raises(RuntimeError):
        R.CorrBlock1DHip(f.cpu(), f.cpu())
    with pytest.raises(RuntimeError):          # same failure as the reference's avg_pool2d
        R.CorrBlock1DHip(f[..., :12].contiguous(), f[..., :12].contiguous(), num_levels=4)
    with pytest.raises(ValueError):
        R.CorrBlock1DHip(f, f[:, :16].contiguous())
    blk = R.CorrBlock1DHip(f, f)
    with pytest.raises(ValueError):
        blk(torch.zeros(1, 2, 3, 32, device="cuda"))
    with pytest.raises(ValueError):
        R.make_corr_fn("reg_cuda", f, f)
    out = R.make_corr_fn("reg_hip", f.half(), f.half(), radius=4, num_levels=4)(torch.zeros(1, 2, 2, 32, device="cuda"))
    assert out.shape == (1, 36, 2, 32)
    out = R.make_corr_fn("reg_hip", f.half(), f.half(), radius=4, num_levels=4, mixed_precision=True,
                         out_dtype=torch.float16)(torch.zeros(1, 2, 2, 32, device="cuda"))
    assert out.shape == (1, 36, 2, 32) and out.dtype == torch.float16
    with pytest.raises(ValueError):
        R.make_corr_fn("alt_hip", f.half(), f.half(), mixed_precision=True, out_dtype=torch.float16)
    with pytest.raises(TypeError):             # fp32 feature maps (no mixed precision) have an fp32 pyramid
        R.make_corr_fn("reg_hip", f, f, out_dtype=torch.float16)
    # non-contiguous views (shared-backbone split, core/raft_stereo.py:80) are accepted
    both = torch.randn(2, 32, 2, 32, device="cuda")
    a, b = both.split(1, dim=0)
    R.CorrBlock1DHip(a.transpose(2, 3).transpose(2, 3), b)(torch.zeros(1, 2, 2, 32, device="cuda"))


def test_training_step_frees_the_pyramid_without_the_cyclic_collector(R):
    """The build node of the autograd graph holds its block weakly (block -> token -> node -> ctx -> block was a
    reference cycle): after a training step, dropping the block and its outputs must return every byte to
    torch's allocator by reference counting alone, with Python's cyclic collector switched off."""
    import gc
    g = torch.Generator(device="cuda").manual_seed(3)
    f1 = torch.randn((2, 256, 24, 96), generator=g, device="cuda").half().requires_grad_(True)
    f2 = torch.randn((2, 256, 24, 96), generator=g, device="cuda").half().requires_grad_(True)
    coords = dev(O.synthetic_coords(2, 24, 96, seed=2))

    def step():
        blk = R.CorrBlock1DHip(f1, f2, num_levels=4, radius=4)
        outs = [blk(coords), blk(coords + 0.25), blk.lookup_convc1(coords, torch.randn(64, 36, device="cuda"),
                                                                    torch.randn(64, device="cuda")).float().mean(1, keepdim=True)]
        sum(o.sum() for o in outs).backward()
        f1.grad = f2.grad = None

    step()                              # warm-up: allocator pools, library workspaces
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        step()
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == base
        # forward only, gradients enabled, no backward: nothing may wait for the collector either
        blk = R.CorrBlock1DHip(f1, f2, num_levels=4, radius=4)
        out = blk(coords)
        del blk, out
        assert torch.cuda.memory_allocated() == base
    finally:
        gc.enable()
