"""The allocation-free variant of rc_convc1_bwd (static device scratch array) against rc_convc1_bwd with a
caller-provided workspace, through the C-ABI: identical results on two calls in a row, the second on a smaller
input (the array keeps the first call's contents).  Collected last (file name) so that a failure here cannot hide
another test under `pytest -x`."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, check, fn_static, B, K, H, W, dtype, rc_dtype, relu, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    corr = torch.randn((B, K, H, W), generator=g, device="cuda")
    weight = torch.randn((64, K), generator=g, device="cuda") * 0.2
    act = torch.relu(torch.randn((B, 64, H, W), generator=g, device="cuda")).to(dtype)
    gact = torch.randn((B, 64, H, W), generator=g, device="cuda").to(dtype)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for static in (False, True):
        gcorr = torch.full_like(corr, float("nan"))
        gw = torch.full((64, K), float("nan"), device="cuda")
        gb = torch.full((64,), float("nan"), device="cuda")
        if static:
            before = torch.cuda.memory_stats()["allocation.all.allocated"]     # allocation COUNT (frees by the gc do not move it)
            check(fn_static(corr.data_ptr(), act.data_ptr(), gact.data_ptr(), weight.data_ptr(), gcorr.data_ptr(),
                            gw.data_ptr(), gb.data_ptr(), B, K, H * W, rc_dtype, relu, st), "rc_convc1_bwd_static")
            assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
        else:
            work = torch.full((int(lib.rc_convc1_bwd_work_bytes()),), 0x7f, dtype=torch.uint8, device="cuda")
            check(lib.rc_convc1_bwd(corr.data_ptr(), act.data_ptr(), gact.data_ptr(), weight.data_ptr(),
                                    gcorr.data_ptr(), gw.data_ptr(), gb.data_ptr(), work.data_ptr(), B, K, H * W,
                                    rc_dtype, relu, st), "rc_convc1_bwd")
        torch.cuda.synchronize()
        outs.append((gcorr, gw, gb))
    for a, b in zip(*outs):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b)          # same kernels, fixed summation order: bit for bit


def test_static_scratch_variant_matches_the_workspace_variant():
    from raft_stereo_amd import _lib
    from raft_stereo_amd._lib import lib, check
    assert torch.cuda.is_available()
    # first a large call, then smaller ones that leave most of the array holding the first call's partials
    _run(lib, check, lib.rc_convc1_bwd_static, 4, 36, 40, 96, torch.float16, _lib.RC_F16, 1, 1)
    _run(lib, check, lib.rc_convc1_bwd_static, 1, 36, 6, 24, torch.float16, _lib.RC_F16, 1, 2)
    _run(lib, check, lib.rc_convc1_bwd_static, 2, 14, 5, 33, torch.float32, _lib.RC_F32, 0, 3)
    _run(lib, check, lib.rc_convc1_bwd_static, 1, 68, 3, 17, torch.bfloat16, _lib.RC_BF16, 1, 4)
    # wider than the array was sized for: refused before any device work, with a clear message
    p = torch.zeros(16, device="cuda").data_ptr()
    rc = lib.rc_convc1_bwd_static(p, p, p, p, p, p, p, 1, 69, 4, _lib.RC_F32, 1, None)
    assert rc == -2 and b"static scratch" in lib.rc_last_error_string()
