"""Golden vectors for sequence_loss, from the reference's own function.

Run in the build container only (the reference tree is not shipped anywhere else):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_loss.py [/root/reference]

`train_stereo.py` cannot be imported here (its module-level imports need tensorboard, cv2, matplotlib and
skimage, none of which is installed), so this script parses the file, takes the `sequence_loss` function
definition (train_stereo.py:35-70) out of the syntax tree, compiles THAT NODE ALONE and runs it with `torch`
as its only global: the reference's own code computes every number stored here; none of its text is kept.
Cases: 22 predictions (the SceneFlow training setting) with holes in the validity map and some |gt| above
max_flow; 5 predictions with a tight max_flow; 2 predictions taken as `[:, :1]` slices.  Stored: inputs, loss,
metrics and the autograd gradient of every prediction, in `update_loss_*.npz`.
"""
import ast
import os
import sys

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_sequence_loss():
    path = os.path.join(REF, "train_stereo.py")
    tree = ast.parse(open(path).read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "sequence_loss")
    glb = {"torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), glb)
    return glb["sequence_loss"]


# name, n predictions, N, H, W, max_flow, loss_gamma
CASES = [("loss_n22_b2_h12_w40", 22, 2, 12, 40, 700, 0.9),
         ("loss_n5_b1_h9_w33", 5, 1, 9, 33, 30, 0.8),
         ("loss_n2_b3_h4_w16", 2, 3, 4, 16, 700, 0.9)]

if __name__ == "__main__":
    torch.set_num_threads(4)
    sequence_loss = reference_sequence_loss()
    for k, (name, n, N, H, W, max_flow, gamma) in enumerate(CASES):
        torch.manual_seed(300 + k)
        gt = -torch.rand(N, 1, H, W) * 60.0
        gt[:, :, :2, :5] = -800.0 if max_flow == 700 else -45.0          # beyond max_flow
        valid = (torch.rand(N, H, W) > 0.2).float()
        preds = []
        for i in range(n):
            p = (gt + torch.randn(N, 1, H, W) * (8.0 / (i + 1))).requires_grad_(True)
            preds.append(p)
        with torch.no_grad():
            preds[-1][0, 0, -1, :4] = gt[0, 0, -1, :4]                   # exact hits: sign(0) = 0 in the gradient
        loss, metrics = sequence_loss(preds, gt, valid, loss_gamma=gamma, max_flow=max_flow)
        loss.backward()
        rec = {"gt": gt, "valid": valid, "loss": loss.detach(), "max_flow": torch.tensor(float(max_flow)),
               "gamma": torch.tensor(gamma, dtype=torch.float64), "n": torch.tensor(n),
               "metrics": torch.tensor([metrics[m] for m in ("epe", "1px", "3px", "5px")], dtype=torch.float64)}
        for i, p in enumerate(preds):
            rec[f"pred_{i}"], rec[f"d_pred_{i}"] = p.detach(), p.grad
        rec = {a: b.numpy().copy() for a, b in rec.items()}
        path = os.path.join(HERE, "update_" + name + ".npz")
        np.savez_compressed(path, **rec)
        print(name, float(loss), metrics, os.path.getsize(path) // 1024, "KiB")
