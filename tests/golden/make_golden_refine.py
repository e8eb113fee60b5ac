"""Golden vectors for the refinement loop, from a forward/backward pass of the reference's whole model.

Run in the build container only (the reference tree is not shipped anywhere else):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_refine.py [/root/reference]

It imports `core.raft_stereo.RAFTStereo` from the reference (the unused `opt_einsum` import is satisfied by an
empty stand-in module), builds a narrow model (hidden_dims 32, 3 GRU layers, slow_fast_gru on, seeded random
weights; no checkpoints exist offline) and runs one seeded synthetic 64x96 pair on CPU in fp32 with
`corr_implementation="reg"`, once in test mode (4 iterations) and once in training mode (3 iterations, loss =
sum of the predictions times a random cotangent).  Recorded: what the encoders hand to the loop (feature maps,
hidden states, context triples: core/raft_stereo.py:83-88), the update block's parameters, the test-mode
outputs, every training-mode prediction and the gradients of the feature maps.  Only data is stored, in
`refine_h16_w24.npz`.
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
import core.raft_stereo as rs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    torch.set_num_threads(4)
    torch.manual_seed(4242)
    rec = {}

    class Recording(rs.CorrBlock1D):
        def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
            rec["fmaps"] = (fmap1, fmap2)
            fmap1.retain_grad() if fmap1.requires_grad else None
            fmap2.retain_grad() if fmap2.requires_grad else None
            super().__init__(fmap1, fmap2, num_levels=num_levels, radius=radius)

    rs.CorrBlock1D = Recording
    args = Namespace(hidden_dims=[32] * 3, corr_implementation="reg", shared_backbone=False, corr_levels=4,
                     corr_radius=4, n_downsample=2, slow_fast_gru=True, n_gru_layers=3, mixed_precision=False)
    model = rs.RAFTStereo(args, "RGB").eval()
    with torch.no_grad():
        for p in model.update_block.parameters():
            p.mul_(2.0)
    ctx = {}
    orig = model.update_block.forward
    def spy(net, inp, *a, **k):
        if "net" not in ctx:
            ctx["net"], ctx["inp"] = [t for t in net], [[t for t in trip] for trip in inp]
        return orig(net, inp, *a, **k)
    model.update_block.forward = spy
    g = torch.Generator().manual_seed(9)
    left = torch.rand((1, 3, 64, 96), generator=g) * 255.0
    right = torch.roll(left, shifts=-4, dims=3) + torch.rand((1, 3, 64, 96), generator=g) * 6.0

    out = {}
    with torch.no_grad():
        flow_lo, flow_up = model(left, right, iters=4, test_mode=True)
    f1, f2 = rec["fmaps"]
    out.update(f1=f1, f2=f2, test_flow_lowres=flow_lo, test_flow_up=flow_up)
    for i in range(3):
        out[f"net{i}"] = ctx["net"][i]
        out[f"ctx{i}"] = torch.cat(ctx["inp"][i], dim=1)
    # training mode: gradients flow back to the feature maps (and everything else) through 3 iterations
    ctx.clear()
    preds = model(left, right, iters=3, test_mode=False)
    f1t, f2t = rec["fmaps"]
    gen = torch.Generator().manual_seed(3)
    cots = [torch.randn(p.shape, generator=gen) for p in preds]
    torch.autograd.backward(preds, cots)
    for i, (p, c) in enumerate(zip(preds, cots)):
        out[f"pred_{i}"], out[f"cot_{i}"] = p, c
    out["d_f1"], out["d_f2"] = f1t.grad, f2t.grad
    out["dw_encoder.convc1.weight"] = model.update_block.encoder.convc1.weight.grad
    out["dw_gru08.convq.weight"] = model.update_block.gru08.convq.weight.grad
    for k, v in model.update_block.state_dict().items():
        out["w_" + k] = v
    out = {k: v.detach().numpy().copy() for k, v in out.items()}
    path = os.path.join(HERE, "refine_h16_w24.npz")
    np.savez_compressed(path, **out)
    print({k: v.shape for k, v in out.items() if not k.startswith("w_")}, os.path.getsize(path) // 1024, "KiB")
