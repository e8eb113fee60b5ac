"""Golden vectors from a forward pass of the reference's whole model.

Run in the build container only (the reference tree is not shipped anywhere else):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_model.py [/root/reference]

It imports `core.raft_stereo.RAFTStereo` from the reference (its unused `opt_einsum` import is
satisfied by an empty stand-in module), builds the default model with seeded random weights (no
checkpoints exist offline), runs it on one seeded synthetic 96x128 pair on CPU in fp32 with
`corr_implementation="reg"`, and records what crosses the correlation block's interface inside
that run: the two feature maps (core/raft_stereo.py:83), the coords of every GRU iteration
(:109-110) with the block's output, the motion encoder's relu(convc1(corr)) (core/update.py:78)
with its parameters, and the final flow.  Only data is stored, in `model_b1_h24_w32.npz`.
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
import core.raft_stereo as rs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ITERS = 4

if __name__ == "__main__":
    torch.set_num_threads(4)
    torch.manual_seed(2024)
    rec = {"coords": [], "corr": []}

    class Recording(rs.CorrBlock1D):
        def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
            rec["f1"], rec["f2"] = fmap1.detach().numpy().copy(), fmap2.detach().numpy().copy()
            super().__init__(fmap1, fmap2, num_levels=num_levels, radius=radius)

        def __call__(self, coords):
            out = super().__call__(coords)
            rec["coords"].append(coords.detach().numpy().copy())
            rec["corr"].append(out.detach().numpy().copy())
            return out

    rs.CorrBlock1D = Recording
    args = Namespace(hidden_dims=[128] * 3, corr_implementation="reg", shared_backbone=False, corr_levels=4,
                     corr_radius=4, n_downsample=2, slow_fast_gru=False, n_gru_layers=3, mixed_precision=False)
    model = rs.RAFTStereo(args, "RGB").eval()
    g = torch.Generator().manual_seed(7)
    left = torch.rand((1, 3, 96, 128), generator=g) * 255.0
    right = torch.roll(left, shifts=-5, dims=3) + torch.rand((1, 3, 96, 128), generator=g) * 8.0
    with torch.no_grad():
        flow_lo, flow_up = model(left, right, iters=ITERS, test_mode=True)
        enc = model.update_block.encoder
        acts = [F.relu(enc.convc1(torch.from_numpy(c))).numpy() for c in rec["corr"]]
    out = {"f1": rec["f1"], "f2": rec["f2"], "L": np.int64(4), "r": np.int64(4),
           "convc1_weight": enc.convc1.weight.detach().numpy().copy(),
           "convc1_bias": enc.convc1.bias.detach().numpy().copy(),
           "flow_lowres": flow_lo.numpy(), "n_iters": np.int64(len(rec["coords"]))}
    for k, (c, o, a) in enumerate(zip(rec["coords"], rec["corr"], acts)):
        out[f"coords_{k}"], out[f"corr_{k}"], out[f"act_{k}"] = c, o, a
    path = os.path.join(HERE, "model_b1_h24_w32.npz")
    np.savez_compressed(path, **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()}, os.path.getsize(path) // 1024, "KiB")
