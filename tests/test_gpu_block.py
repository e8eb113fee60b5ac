"""GPU parity of the multi-scale update block wired from this package's kernels (BasicMultiUpdateBlockHip)
against vectors captured from the reference's BasicMultiUpdateBlock running on its own CorrBlock1D lookup.

The block's convolutions, pooling and resizing are torch's on both sides (CPU there, MIOpen here), so the bar is
1e-4 of max for outputs and 1e-3 for gradients; the library's own arithmetic inside (GRU gates, fused
lookup + convc1) is held to 1e-5 in test_gpu_update.py / test_gpu_parity.py.
"""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import corr_oracle as O
from conftest import BLOCK_CASE

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel(a, b):
    return O.rel_to_max(a.detach().float().cpu().numpy(), b)


@pytest.fixture(scope="module")
def setup(golden):
    import raft_stereo_amd as R
    g = golden(BLOCK_CASE)
    args = Namespace(corr_levels=4, corr_radius=4, n_gru_layers=3, n_downsample=2)
    blk = R.BasicMultiUpdateBlockHip(args, hidden_dims=[16] * 3).cuda()
    missing = blk.load_state_dict({k[2:]: dev(v) for k, v in g.items() if k.startswith("w_")}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys        # the reference's names, as they are
    return R, g, blk


def _inputs(g, grad=False):
    net = [dev(g[f"net{i}"]).requires_grad_(grad) for i in range(3)]
    inp = [list(dev(g[f"ctx{i}"]).split(split_size=16, dim=1)) for i in range(3)]
    return net, inp


def test_partial_call_matches_reference(setup):
    R, g, blk = setup
    net, inp = _inputs(g)
    with torch.no_grad():
        out = blk(net, inp, iter32=True, iter16=False, iter08=False, update=False)
    assert rel(out[2], g["partial_net2"]) < 1e-4
    assert torch.equal(out[0], dev(g["net0"])) and torch.equal(out[1], dev(g["net1"]))


@pytest.mark.parametrize("fused", [False, True])
def test_full_call_matches_reference(setup, fused):
    """`fused`: the block gets the correlation block and the coordinates instead of a looked-up tensor
    (lookup + convc1 + ReLU in one kernel)."""
    R, g, blk = setup
    net, inp = _inputs(g)
    with torch.no_grad():
        if fused:
            corr_fn = R.CorrBlock1DHip(dev(g["f1"]), dev(g["f2"]), num_levels=4, radius=4)
            out_net, mask, delta = blk(net, inp, flow=dev(g["flow"]), corr_fn=corr_fn, coords=dev(g["coords"]))
        else:
            out_net, mask, delta = blk(net, inp, dev(g["corr"]), dev(g["flow"]))
    for i in range(3):
        assert rel(out_net[i], g[f"out_net{i}"]) < 1e-4, i
    assert rel(mask, g["mask"]) < 1e-4
    assert rel(delta, g["delta_flow"]) < 1e-4


@pytest.mark.parametrize("fused", [False, True])
def test_backward_matches_reference_autograd(setup, fused):
    R, g, blk = setup
    net, inp = _inputs(g, grad=True)
    blk.zero_grad()
    if fused:
        corr_fn = R.CorrBlock1DHip(dev(g["f1"]), dev(g["f2"]), num_levels=4, radius=4)
        out_net, mask, delta = blk(list(net), inp, flow=dev(g["flow"]), corr_fn=corr_fn, coords=dev(g["coords"]))
    else:
        out_net, mask, delta = blk(list(net), inp, dev(g["corr"]), dev(g["flow"]))
    torch.autograd.backward([out_net[0], mask, delta], [dev(g["cot_net0"]), dev(g["cot_mask"]), dev(g["cot_delta"])])
    for i in range(3):
        assert rel(net[i].grad, g[f"d_net{i}"]) < 1e-3, i
    params = dict(blk.named_parameters())
    for k in g:
        if k.startswith("dw_"):
            assert rel(params[k[3:]].grad, g[k]) < 1e-3, k


def test_two_layer_realtime_block_matches_reference(golden):
    """n_gru_layers=2, n_downsample=3 (the realtime configuration): partial mid-resolution call and full call."""
    import raft_stereo_amd as R
    g = golden("block_l2_h16")
    args = Namespace(corr_levels=4, corr_radius=4, n_gru_layers=2, n_downsample=3)
    blk = R.BasicMultiUpdateBlockHip(args, hidden_dims=[16] * 3).cuda()
    blk.load_state_dict({k[2:]: dev(v) for k, v in g.items() if k.startswith("w_")}, strict=True)

    def fresh():
        net = [dev(g["net0"]), dev(g["net1"]), torch.zeros(1, 16, 1, 1, device="cuda")]
        inp = [list(dev(g[f"ctx{i}"]).split(split_size=16, dim=1)) for i in range(2)] + [None]
        return net, inp
    with torch.no_grad():
        net, inp = fresh()
        part = blk(net, inp, iter32=False, iter16=True, iter08=False, update=False)
        assert rel(part[1], g["partial_net1"]) < 1e-4
        net, inp = fresh()
        out_net, mask, delta = blk(net, inp, dev(g["corr"]), dev(g["flow"]), iter32=False, iter16=True)
    assert rel(out_net[0], g["out_net0"]) < 1e-4 and rel(out_net[1], g["out_net1"]) < 1e-4
    assert tuple(mask.shape) == g["mask"].shape and rel(mask, g["mask"]) < 1e-4
    assert rel(delta, g["delta_flow"]) < 1e-4
