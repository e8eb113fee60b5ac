"""Golden vectors for the multi-scale update block, from the reference's own Python.

Run in the build container only (the reference tree is not shipped anywhere else):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_block.py [/root/reference]

It imports `core.update.BasicMultiUpdateBlock` and `core.corr.CorrBlock1D` from the reference (the unused
`opt_einsum` import is satisfied by an empty stand-in module), builds the block with 3 GRU layers and small
hidden sizes (seeded random weights, scaled up so the gates leave their linear range), and runs on CPU in fp32:
  * one slow-fast style partial call (`iter08=False, iter16=False, update=False`, core/raft_stereo.py:113-114),
  * one full call on the lookup of the reference's own CorrBlock1D for seeded feature maps and coordinates,
then back-propagates a random cotangent through the full call.  Stored: parameters, inputs, outputs and the
gradients of the hidden states and of a few parameters, in `block_l3_h16.npz`.  Only data is stored.
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.modules.setdefault("opt_einsum", types.SimpleNamespace(contract=None))
from core.update import BasicMultiUpdateBlock  # noqa: E402
from core.corr import CorrBlock1D  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

def two_layer_case():
    """The realtime configuration's block (n_gru_layers=2, n_downsample=3): one mid-resolution partial call
    (core/raft_stereo.py:115-116) and one full call.  Forward only."""
    torch.manual_seed(78)
    hid, H, W = 16, 10, 14
    args = Namespace(corr_levels=4, corr_radius=4, n_gru_layers=2, n_downsample=3)
    blk = BasicMultiUpdateBlock(args, hidden_dims=[hid] * 3)
    with torch.no_grad():
        for p in blk.parameters():
            p.mul_(3.0)
        sizes = [(H, W), (H // 2, W // 2)]
        net = [torch.tanh(torch.randn(1, hid, h, w)) for h, w in sizes] + [torch.zeros(1, hid, 1, 1)]
        ctx = [torch.randn(1, 3 * hid, h, w) for h, w in sizes]
        inp = [list(c.split(split_size=hid, dim=1)) for c in ctx] + [None]
        corr, flow = torch.randn(1, 36, H, W), torch.randn(1, 2, H, W)
        rec = {"corr": corr, "flow": flow, "net0": net[0], "net1": net[1], "ctx0": ctx[0], "ctx1": ctx[1]}
        part = blk([n.clone() for n in net], inp, iter32=False, iter16=True, iter08=False, update=False)
        rec["partial_net1"] = part[1]
        out_net, mask, delta = blk([n.clone() for n in net], inp, corr, flow, iter32=False, iter16=True)
        rec.update(out_net0=out_net[0], out_net1=out_net[1], mask=mask, delta_flow=delta)
        for k, v in blk.state_dict().items():
            rec["w_" + k] = v
    rec = {k: v.detach().numpy().copy() for k, v in rec.items()}
    path = os.path.join(HERE, "block_l2_h16.npz")
    np.savez_compressed(path, **rec)
    print("l2", {k: v.shape for k, v in rec.items() if not k.startswith("w_")}, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.set_num_threads(4)
    two_layer_case()
    torch.manual_seed(77)
    hid, H, W, D = 16, 16, 24, 32
    args = Namespace(corr_levels=4, corr_radius=4, n_gru_layers=3, n_downsample=2)
    blk = BasicMultiUpdateBlock(args, hidden_dims=[hid] * 3)
    with torch.no_grad():
        for p in blk.parameters():
            p.mul_(3.0)
    sizes = [(H, W), (H // 2, W // 2), (H // 4, W // 4)]
    net = [torch.tanh(torch.randn(1, hid, h, w)).requires_grad_(True) for h, w in sizes]
    ctx = [torch.randn(1, 3 * hid, h, w) for h, w in sizes]
    inp = [list(c.split(split_size=hid, dim=1)) for c in ctx]
    f1, f2 = torch.randn(1, D, H, W), torch.randn(1, D, H, W)
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(1, 1, H, W)
    ys = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1).expand(1, 1, H, W)
    coords = torch.cat([xs - torch.rand(1, 1, H, W) * 6.0, ys], dim=1).contiguous()
    flow = coords - torch.cat([xs, ys], dim=1)
    corr = CorrBlock1D(f1, f2, num_levels=4, radius=4)(coords)

    rec = {"f1": f1, "f2": f2, "coords": coords, "flow": flow, "corr": corr}
    for i in range(3):
        rec[f"net{i}"], rec[f"ctx{i}"] = net[i], ctx[i]
    part = blk([n.detach() for n in net], inp, iter32=True, iter16=False, iter08=False, update=False)
    rec["partial_net2"] = part[2]
    out_net, mask, delta = blk(list(net), inp, corr, flow)
    for i in range(3):
        rec[f"out_net{i}"] = out_net[i]
    rec["mask"], rec["delta_flow"] = mask, delta
    g = torch.Generator().manual_seed(5)
    cot = [torch.randn(o.shape, generator=g) for o in (out_net[0], mask, delta)]
    torch.autograd.backward([out_net[0], mask, delta], cot)
    rec["cot_net0"], rec["cot_mask"], rec["cot_delta"] = cot
    for i in range(3):
        rec[f"d_net{i}"] = net[i].grad
    for name in ("encoder.convc1.weight", "encoder.convc1.bias", "gru08.convz.weight", "gru16.convq.bias",
                 "gru32.convr.weight", "flow_head.conv2.weight"):
        rec["dw_" + name] = dict(blk.named_parameters())[name].grad
    for k, v in blk.state_dict().items():
        rec["w_" + k] = v
    rec = {k: v.detach().numpy().copy() for k, v in rec.items()}
    path = os.path.join(HERE, "block_l3_h16.npz")
    np.savez_compressed(path, **rec)
    print({k: v.shape for k, v in rec.items() if not k.startswith("w_")}, os.path.getsize(path) // 1024, "KiB")
