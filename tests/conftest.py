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


_MARGINS = {}


def record_margin(what, err_over_scale, tol):
    """Remember the worst error/bound ratio seen per comparison label; written at session end to
    gpurun_out/parity_margins.txt (the evidence for how much headroom each tolerance has)."""
    cur = _MARGINS.get(what)
    if cur is None or err_over_scale / tol > cur[0] / cur[1]:
        _MARGINS[what] = (float(err_over_scale), float(tol))


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_margins.txt"), "w") as f:
            f.write("# worst error (relative to the comparison's scale) per label, its bound, and the ratio\n")
            for what, (e, t) in sorted(_MARGINS.items(), key=lambda kv: -kv[1][0] / kv[1][1]):
                f.write("%-70s err %.3e  bound %.1e  used %.3f\n" % (what, e, t, e / t))
    except OSError:
        pass


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
