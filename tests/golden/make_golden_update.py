"""Golden vectors for the GRU gate arithmetic and the convex upsampling, from the reference's own Python.

Run in the build container only (the reference tree is not shipped anywhere else):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_update.py [/root/reference]

It imports `core.update.ConvGRU` and `core.raft_stereo.RAFTStereo` from the reference (the unused
`opt_einsum` import is satisfied by an empty stand-in module).

ConvGRU (core/update.py:16-32): seeded random weights and inputs on CPU in fp32; cz/cr/cq are the
`.split(dim=1)` views of one tensor as in core/raft_stereo.py:88.  Forward hooks on the three convolutions
record their outputs (the pre-activations) and keep their gradients, so the pointwise arithmetic can be
checked apart from the convolutions.  A random grad_out is back-propagated through the module.

upsample_flow (core/raft_stereo.py:55-67): the method is called unbound on a stand-in `self` that only
carries `args.n_downsample`; fp32 masks for factors 2, 4, 8 and one fp16 mask (the autocast case).

pool2x / interp (core/update.py:87-95): the reference's two helper functions on odd and even sizes.

Only data is stored, in `update_*.npz`.
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
from core.update import ConvGRU, pool2x, interp  # noqa: E402
from core.raft_stereo import RAFTStereo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name, N, hidden, input widths of x_list, H, W
GRU_CASES = [("gru_n2_c32_h6_w10", 2, 32, (24, 24), 6, 10),
             ("gru_n1_c7_h3_w5", 1, 7, (9,), 3, 5),
             ("gru_n3_c16_h4_w12", 3, 16, (16, 8, 8), 4, 12)]
# name, N, D, H, W, n_downsample
UP_CASES = [("up_f4_n2_h5_w7", 2, 2, 5, 7, 2),
            ("up_f8_n1_h3_w70", 1, 2, 3, 70, 3),
            ("up_f2_n1_h4_w9", 1, 2, 4, 9, 1)]


def run_gru(name, N, C, xdims, H, W, seed):
    torch.manual_seed(seed)
    gru = ConvGRU(C, sum(xdims))
    with torch.no_grad():
        for p in gru.parameters():
            p.mul_(4.0)                        # default init keeps the gates near 0.5; spread them
    h = torch.tanh(torch.randn(N, C, H, W)).requires_grad_(True)
    ctx = torch.randn(N, 3 * C, H, W).requires_grad_(True)
    xs = [torch.randn(N, d, H, W).requires_grad_(True) for d in xdims]
    pre = {}

    def hook(key):
        def fn(_m, _i, out):
            out.retain_grad()
            pre[key] = out
        return fn
    for key in ("convz", "convr", "convq"):
        getattr(gru, key).register_forward_hook(hook(key))
    cz, cr, cq = ctx.split(split_size=C, dim=1)
    out = gru(h, cz, cr, cq, *xs)
    gout = torch.randn_like(out)
    out.backward(gout)
    rec = {"h": h, "ctx": ctx, "h_out": out, "grad_out": gout,
           "zc": pre["convz"], "rc": pre["convr"], "qc": pre["convq"],
           "d_zc": pre["convz"].grad, "d_rc": pre["convr"].grad, "d_qc": pre["convq"].grad,
           "d_h": h.grad, "d_ctx": ctx.grad}
    for i, x in enumerate(xs):
        rec[f"x{i}"], rec[f"d_x{i}"] = x, x.grad
    for k, p in gru.named_parameters():
        rec["w_" + k], rec["dw_" + k] = p, p.grad
    rec = {k: v.detach().numpy().copy() for k, v in rec.items()}
    rec["n_x"] = np.int64(len(xs))
    rec["C"] = np.int64(C)
    return rec


def run_up(name, N, D, H, W, nd, seed):
    torch.manual_seed(seed)
    f = 2 ** nd
    me = types.SimpleNamespace(args=Namespace(n_downsample=nd))
    flow = torch.randn(N, D, H, W) * 5.0
    mask = torch.randn(N, 9 * f * f, H, W) * 2.0
    with torch.no_grad():
        rec = {"flow": flow, "mask": mask, "up": RAFTStereo.upsample_flow(me, flow, mask),
               "up_mask16": RAFTStereo.upsample_flow(me, flow, mask.half())}
    # autograd through the same method: fp32 mask, and the fp16 mask of the autocast case
    gout = torch.randn_like(rec["up"])
    rec["grad_out"] = gout
    for tag, mk in (("", mask), ("_mask16", mask.half())):
        fl, mk = flow.clone().requires_grad_(True), mk.clone().requires_grad_(True)
        RAFTStereo.upsample_flow(me, fl, mk).backward(gout)
        rec["d_flow" + tag], rec["d_mask" + tag] = fl.grad, mk.grad.float()
    rec = {k: v.detach().numpy().copy() for k, v in rec.items()}
    rec["n_downsample"] = np.int64(nd)
    return rec


def run_resample(seed):
    """pool2x / interp of core/update.py:87-95 on odd and even sizes."""
    torch.manual_seed(seed)
    rec = {}
    for i, shp in enumerate([(2, 5, 7, 9), (1, 3, 8, 12), (1, 2, 1, 5), (1, 4, 12, 80)]):
        x = torch.randn(shp).requires_grad_(True)
        y = pool2x(x)
        g = torch.randn(y.shape)
        y.backward(g)
        rec[f"pool_in{i}"], rec[f"pool_out{i}"], rec[f"pool_cot{i}"], rec[f"pool_grad{i}"] = x, y, g, x.grad
    for i, (shp, dst) in enumerate([((1, 3, 4, 6), (8, 12)), ((2, 5, 4, 5), (7, 9)), ((1, 2, 3, 3), (3, 3)),
                                    ((1, 4, 6, 40), (12, 80)), ((1, 2, 2, 2), (1, 5))]):
        x = torch.randn(shp).requires_grad_(True)
        y = interp(x, torch.empty(1, 1, *dst))
        g = torch.randn(y.shape)
        y.backward(g)
        rec[f"interp_in{i}"], rec[f"interp_out{i}"], rec[f"interp_cot{i}"], rec[f"interp_grad{i}"] = x, y, g, x.grad
    return {k: v.detach().numpy().copy() for k, v in rec.items()}


if __name__ == "__main__":
    torch.set_num_threads(4)
    rec = run_resample(400)
    path = os.path.join(HERE, "update_resample.npz")
    np.savez_compressed(path, **rec)
    print("resample", {k: v.shape for k, v in rec.items() if "out" in k}, os.path.getsize(path) // 1024, "KiB")
    for n, case in enumerate(GRU_CASES):
        rec = run_gru(*case, seed=100 + n)
        path = os.path.join(HERE, "update_" + case[0] + ".npz")
        np.savez_compressed(path, **rec)
        print(case[0], os.path.getsize(path) // 1024, "KiB")
    for n, case in enumerate(UP_CASES):
        rec = run_up(*case, seed=200 + n)
        assert rec["up"].dtype == np.float32 and rec["up_mask16"].dtype == np.float32
        path = os.path.join(HERE, "update_" + case[0] + ".npz")
        np.savez_compressed(path, **rec)
        print(case[0], {k: getattr(v, "shape", v) for k, v in rec.items()}, os.path.getsize(path) // 1024, "KiB")
