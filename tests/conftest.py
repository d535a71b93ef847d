import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

GOLDEN_CASES = ["tiny_u1_k5", "tiny_u3_k5_N", "small_u8_k19", "small_u8_k19_mse",
                "tandem_u3_k5", "mid_u8_k19_L200", "c1_u100_k19_L200"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """One tests/golden/*.npz fixture (generated from the imported reference by
    tools/make_golden.py): inputs + the reference's outputs."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.U, self.k, self.L, self.T, self.B, self.n_batches = [int(v) for v in self.z["cfg"]]
        self.loss_kind = str(self.z["loss_kind"])
        self.codes = self.z["codes"]
        self.y = self.z["y"]

    def group(self, prefix):
        return {k[len(prefix):]: self.z[k] for k in self.z.files if k.startswith(prefix)}

    def sd(self, prefix="sd/"):
        return self.group(prefix)

    def onehot(self, batch=0):
        c = self.codes[batch * self.B:(batch + 1) * self.B]
        x = np.zeros((c.shape[0], 4, c.shape[1]), dtype=np.float32)
        for a in range(4):
            x[:, a, :] = (c == a)
        return x

    def targets(self, batch=0):
        return self.y[batch * self.B:(batch + 1) * self.B]

    def keep_mask(self):
        bits = self.z["drop/keep_bits"]
        return np.unpackbits(bits, axis=1)[:, :100 * self.U]


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return Golden(request.param)
