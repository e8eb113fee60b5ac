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
            # driver-level free memory: the library is called through ctypes, so torch's allocator statistics would
            # not see a hipMalloc inside it
            torch.cuda.synchronize()
            before = torch.cuda.mem_get_info()[0]
            check(fn_static(corr.data_ptr(), act.data_ptr(), gact.data_ptr(), weight.data_ptr(), gcorr.data_ptr(),
                            gw.data_ptr(), gb.data_ptr(), B, K, H * W, rc_dtype, relu, st), "rc_convc1_bwd_static")
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] == before
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


def _captured(lib, check):
    """The use the entry point is for: inside a stream capture (a hipMalloc there fails the capture), replayed,
    against the workspace variant run eagerly."""
    from raft_stereo_amd import _lib
    B, K, H, W = 2, 36, 12, 40
    g = torch.Generator(device="cuda").manual_seed(9)
    corr = torch.randn((B, K, H, W), generator=g, device="cuda")
    weight = torch.randn((64, K), generator=g, device="cuda") * 0.2
    act = torch.relu(torch.randn((B, 64, H, W), generator=g, device="cuda")).half()
    gact = torch.randn((B, 64, H, W), generator=g, device="cuda").half()
    ref = (torch.empty_like(corr), torch.empty((64, K), device="cuda"), torch.empty((64,), device="cuda"))
    work = torch.empty((int(lib.rc_convc1_bwd_work_bytes()),), dtype=torch.uint8, device="cuda")
    check(lib.rc_convc1_bwd(corr.data_ptr(), act.data_ptr(), gact.data_ptr(), weight.data_ptr(), ref[0].data_ptr(),
                            ref[1].data_ptr(), ref[2].data_ptr(), work.data_ptr(), B, K, H * W, _lib.RC_F16, 1,
                            torch.cuda.current_stream().cuda_stream), "rc_convc1_bwd")
    out = (torch.full_like(corr, float("nan")), torch.full((64, K), float("nan"), device="cuda"),
           torch.full((64,), float("nan"), device="cuda"))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # one call outside the capture: the array's address is looked up there
        check(lib.rc_convc1_bwd_static(corr.data_ptr(), act.data_ptr(), gact.data_ptr(), weight.data_ptr(),
                                       out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), B, K, H * W,
                                       _lib.RC_F16, 1, side.cuda_stream), "rc_convc1_bwd_static")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for t in out:
        t.fill_(float("nan"))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        check(lib.rc_convc1_bwd_static(corr.data_ptr(), act.data_ptr(), gact.data_ptr(), weight.data_ptr(),
                                       out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), B, K, H * W,
                                       _lib.RC_F16, 1, torch.cuda.current_stream().cuda_stream),
              "rc_convc1_bwd_static (captured)")
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(ref, out):
        assert torch.equal(a, b)


def test_static_scratch_variant_matches_the_workspace_variant():
    from raft_stereo_amd import _lib
    from raft_stereo_amd._lib import lib, check
    assert torch.cuda.is_available()
    # first a large call, then smaller ones that leave most of the array holding the first call's partials
    _run(lib, check, lib.rc_convc1_bwd_static, 4, 36, 40, 96, torch.float16, _lib.RC_F16, 1, 1)
    _run(lib, check, lib.rc_convc1_bwd_static, 1, 36, 6, 24, torch.float16, _lib.RC_F16, 1, 2)
    _run(lib, check, lib.rc_convc1_bwd_static, 2, 14, 5, 33, torch.float32, _lib.RC_F32, 0, 3)
    _run(lib, check, lib.rc_convc1_bwd_static, 1, 68, 3, 17, torch.bfloat16, _lib.RC_BF16, 1, 4)
    _captured(lib, check)
    # wider than the array was sized for: refused before any device work, with a clear message
    p = torch.zeros(16, device="cuda").data_ptr()
    rc = lib.rc_convc1_bwd_static(p, p, p, p, p, p, p, 1, 69, 4, _lib.RC_F32, 1, None)
    assert rc == -2 and b"static scratch" in lib.rc_last_error_string()
