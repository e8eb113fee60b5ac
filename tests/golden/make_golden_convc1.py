"""Golden vectors for the fused lookup + convc1 path, from the reference's own Python.

Run in the build container only (the reference tree is not shipped anywhere else), after
make_golden.py:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_convc1.py [/root/reference]

It imports `core.update.BasicMotionEncoder` from the reference (its unused `opt_einsum` import is
satisfied by an empty stand-in module), builds the encoder for the case's (levels, radius) with
seeded random weights and applies its own `F.relu(self.convc1(corr))` (core/update.py:71,78) to the
lookup results already stored by make_golden.py (outputs of the reference's CorrBlock1D).  Stored:
convc1 weight, bias, the activations and, for a seeded random grad_act, the autograd gradients w.r.t. the
lookup result, the weight and the bias, in `convc1_<case>.npz`.  Only data is stored.
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch
import torch.nn.functional as F

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.modules.setdefault("opt_einsum", types.SimpleNamespace(contract=None))
from core.update import BasicMotionEncoder  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["tiny_b2_d32_h3_w24", "odd_b1_d64_h2_w40x45", "d256_b1_h4_w128", "lv2_r3_b1_d32_h2_w33"]

if __name__ == "__main__":
    torch.set_num_threads(4)
    for n, case in enumerate(CASES):
        g = np.load(os.path.join(HERE, case + ".npz"))
        L, r = int(g["L"]), int(g["r"])
        torch.manual_seed(4321 + n)
        enc = BasicMotionEncoder(Namespace(corr_levels=L, corr_radius=r))
        with torch.no_grad():
            enc.convc1.weight.mul_(3.0)          # default init is tiny next to |corr|; keep the ReLU busy
            enc.convc1.bias.uniform_(-2.0, 2.0)
            out = {"weight": enc.convc1.weight.numpy().copy(), "bias": enc.convc1.bias.numpy().copy()}
            for k in ("smooth", "random", "edge"):
                out[f"act_{k}"] = F.relu(enc.convc1(torch.from_numpy(g[f"lookup_{k}"]))).numpy()
        # autograd through the same two lines: gradients of a random grad_act w.r.t. corr, weight and bias
        for j, k in enumerate(("smooth", "random", "edge")):
            corr = torch.from_numpy(g[f"lookup_{k}"]).requires_grad_(True)
            enc.zero_grad()
            act = F.relu(enc.convc1(corr))
            gact = torch.randn(act.shape, generator=torch.Generator().manual_seed(9000 + 10 * n + j))
            act.backward(gact)
            out[f"gact_{k}"], out[f"d_corr_{k}"] = gact.numpy(), corr.grad.numpy()
            out[f"d_weight_{k}"], out[f"d_bias_{k}"] = enc.convc1.weight.grad.numpy().copy(), \
                enc.convc1.bias.grad.numpy().copy()
        path = os.path.join(HERE, "convc1_" + case + ".npz")
        np.savez_compressed(path, **out)
        print(case, {k: v.shape for k, v in out.items()}, os.path.getsize(path) // 1024, "KiB")
