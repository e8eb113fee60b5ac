"""End-to-end parity of everything after the encoders: refine_disparity (correlation block + update block +
convex upsampling, all on this package's operators) against a forward/backward pass of the reference's whole
RAFTStereo on CPU, fed the feature maps, hidden states and context triples its encoders produced.

Bar: SURVEY 8(d) model level, mean |delta disparity| <= 0.01 px (the result here is orders of magnitude
tighter in fp32); 16-bit correlation under the same bar.
"""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import corr_oracle as O
from conftest import REFINE_CASE

pytestmark = pytest.mark.gpu

ARGS = Namespace(hidden_dims=[32] * 3, corr_levels=4, corr_radius=4, n_downsample=2, slow_fast_gru=True,
                 n_gru_layers=3, mixed_precision=False)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel(a, b):
    return O.rel_to_max(a.detach().float().cpu().numpy(), b)


@pytest.fixture(scope="module")
def setup(golden):
    import raft_stereo_amd as R
    g = golden(REFINE_CASE)
    blk = R.BasicMultiUpdateBlockHip(ARGS, hidden_dims=ARGS.hidden_dims).cuda()
    blk.load_state_dict({k[2:]: dev(v) for k, v in g.items() if k.startswith("w_")}, strict=True)
    return R, g, blk


def _context(g):
    net = [dev(g[f"net{i}"]) for i in range(3)]
    inp = [list(dev(g[f"ctx{i}"]).split(split_size=32, dim=1)) for i in range(3)]
    return net, inp


@pytest.mark.parametrize("impl", ["reg_hip", "alt_hip"])
def test_inference_matches_the_reference_model(setup, impl):
    R, g, blk = setup
    net, inp = _context(g)
    with torch.no_grad():
        corr_fn = R.make_corr_fn(impl, dev(g["f1"]), dev(g["f2"]), radius=4, num_levels=4)
        flow_lo, flow_up = R.refine_disparity(blk, corr_fn, net, inp, ARGS, iters=4, test_mode=True)
    assert tuple(flow_up.shape) == g["test_flow_up"].shape
    assert float((flow_up.cpu() - torch.from_numpy(g["test_flow_up"])).abs().mean()) <= 0.01      # the stated bar
    assert rel(flow_up, g["test_flow_up"]) < 1e-4
    assert rel(flow_lo, g["test_flow_lowres"]) < 1e-4
    assert float(flow_lo[:, 1].abs().max()) == 0.0                      # epipolar: no vertical motion


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_inference_with_a_16bit_pyramid_stays_within_the_disparity_bar(setup, dtype):
    """Only the correlation path changes precision (16-bit feature maps -> 16-bit pyramid, fused lookup + convc1 in
    that type); the rest of the update block stays fp32, so the deviation is the correlation path's alone."""
    R, g, blk = setup
    net, inp = _context(g)
    with torch.no_grad():
        corr_fn = R.make_corr_fn("reg_hip", dev(g["f1"]).to(dtype), dev(g["f2"]).to(dtype), radius=4, num_levels=4,
                                 mixed_precision=True)
        assert corr_fn.volume.dtype == dtype
        _, flow_up = R.refine_disparity(blk, corr_fn, net, inp, ARGS, iters=4, test_mode=True)
    dev_px = float((flow_up.cpu() - torch.from_numpy(g["test_flow_up"])).abs().mean())
    assert dev_px <= (0.01 if dtype == torch.float16 else 0.05), dev_px


def test_inference_under_full_autocast_runs_and_stays_close(setup):
    """Everything under fp16 autocast (evaluate_stereo.py:230): torch's fp16 convolutions are part of the
    difference here, on a random-weight model with amplified weights; a sanity bar, not the parity bar."""
    R, g, blk = setup
    net, inp = _context(g)
    amp = Namespace(**{**vars(ARGS), "mixed_precision": True})
    with torch.no_grad():
        corr_fn = R.make_corr_fn("reg_hip", dev(g["f1"]).half(), dev(g["f2"]).half(), radius=4, num_levels=4,
                                 mixed_precision=True)
        _, flow_up = R.refine_disparity(blk, corr_fn, [n.half() for n in net],
                                        [[t.half() for t in trip] for trip in inp], amp, iters=4, test_mode=True)
    assert flow_up.dtype == torch.float32
    assert float((flow_up.cpu() - torch.from_numpy(g["test_flow_up"])).abs().mean()) <= 0.05


def test_training_matches_the_reference_model(setup):
    R, g, blk = setup
    net, inp = _context(g)
    f1, f2 = dev(g["f1"]).requires_grad_(True), dev(g["f2"]).requires_grad_(True)
    blk.zero_grad()
    corr_fn = R.make_corr_fn("reg_hip", f1, f2, radius=4, num_levels=4)
    preds = R.refine_disparity(blk, corr_fn, net, inp, ARGS, iters=3, test_mode=False)
    assert len(preds) == 3
    for i, p in enumerate(preds):
        assert rel(p, g[f"pred_{i}"]) < 1e-4, i
    torch.autograd.backward(preds, [dev(g[f"cot_{i}"]) for i in range(3)])
    assert rel(f1.grad, g["d_f1"]) < 1e-3
    assert rel(f2.grad, g["d_f2"]) < 1e-3
    params = dict(blk.named_parameters())
    for k in ("encoder.convc1.weight", "gru08.convq.weight"):
        assert rel(params[k].grad, g["dw_" + k]) < 1e-3, k
    # and the loss on those predictions
    gt = torch.zeros_like(preds[0])
    valid = torch.ones(gt.shape[0], *gt.shape[2:], device="cuda")
    loss, metrics = R.sequence_loss_hip([p.detach() for p in preds], gt, valid)
    from oracle import update_oracle as U
    ref, ref_m, _ = U.sequence_loss([g[f"pred_{i}"] for i in range(3)], gt.cpu().numpy(), valid.cpu().numpy())
    assert abs(loss.item() - ref) <= 1e-4 * abs(ref) and abs(metrics["epe"] - ref_m["epe"]) <= 1e-4 * ref_m["epe"]
