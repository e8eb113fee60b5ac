import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
_NPZ = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))
GOLDEN_CASES = [n for n in _NPZ if not n.startswith(("convc1_", "model_", "update_", "block_", "refine_"))]
# what crossed the correlation block's interface in a forward pass of the reference's whole model
MODEL_CASE = "model_b1_h24_w32"
# cases that also have the reference's relu(convc1(lookup)) activations (make_golden_convc1.py)
CONVC1_CASES = [n[len("convc1_"):] for n in _NPZ if n.startswith("convc1_")]
# the reference's ConvGRU and upsample_flow (make_golden_update.py)
GRU_CASES = [n for n in _NPZ if n.startswith("update_gru_")]
RESAMPLE_CASE = "update_resample"
UPSAMPLE_CASES = [n for n in _NPZ if n.startswith("update_up_")]
# the reference's BasicMultiUpdateBlock on its own CorrBlock1D lookup (make_golden_block.py)
BLOCK_CASE = "block_l3_h16"
# what the reference's encoders hand to its refinement loop, and what the loop returns (make_golden_refine.py)
REFINE_CASE = "refine_h16_w24"
# the reference's sequence_loss (make_golden_loss.py)
LOSS_CASES = [n for n in _NPZ if n.startswith("update_loss_")]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]
    return load
