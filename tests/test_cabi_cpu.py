"""CPU-only checks of the C-ABI boundary: the shared library builds, loads, and exports exactly
the symbols include/explainn_hip.h declares.  No compute call is made (there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from explainn_amd import _lib
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "explainn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(explainn_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from explainn_amd import _lib
    declared = _declared_symbols()
    assert declared, "header declares entry points"
    for name in declared:
        assert hasattr(lib, name), "missing export " + name
    assert sorted(_lib.EXPORTS) == declared


def test_struct_layout_matches_header():
    """ctypes Structures list the same fields, in the same order, as the C structs."""
    from explainn_amd import _lib
    text = open(os.path.join(ROOT, "include", "explainn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct explainn_params \{(.*?)\} explainn_params;", text, re.S).group(1)
    fields = re.findall(r"\*\s*([a-z0-9_]+)\s*;", body)
    assert tuple(fields) == _lib.PARAM_FIELDS
    body = re.search(r"typedef struct explainn_grads \{(.*?)\} explainn_grads;", text, re.S).group(1)
    fields = re.findall(r"\*\s*([a-z0-9_]+)\s*;", body)
    assert tuple(fields) == _lib.GRAD_FIELDS


def test_model_surface_on_cpu():
    """Constructor, _options, state_dict keys/shapes, deepcopy -- and a loud failure (no CPU
    fallback) when forward is called off-device."""
    import copy
    import torch
    from explainn_amd import ExplaiNN
    from oracle import explainn_oracle as orc
    m = ExplaiNN(6, 19, 200, 3)
    assert m._options == dict(cnn_units=6, kernel_size=19, sequence_length=200, n_features=3,
                              weights_file=None)
    ref = orc.random_state_dict(6, 19, 200, 3)
    sd = m.state_dict()
    assert list(sd.keys()) == [
        "linears.0.weight", "linears.0.bias",
        "linears.1.weight", "linears.1.bias", "linears.1.running_mean", "linears.1.running_var",
        "linears.1.num_batches_tracked",
        "linears.6.weight", "linears.6.bias",
        "linears.7.weight", "linears.7.bias", "linears.7.running_mean", "linears.7.running_var",
        "linears.7.num_batches_tracked",
        "linears.10.weight", "linears.10.bias",
        "linears.11.weight", "linears.11.bias", "linears.11.running_mean", "linears.11.running_var",
        "linears.11.num_batches_tracked",
        "final.weight", "final.bias"]
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert m.__class__.__name__ == "ExplaiNN"
    m2 = copy.deepcopy(m)
    assert m2.linears._owner() is m2 and m2._rt is not m._rt
    assert isinstance(m.linears[0].weight, torch.nn.Parameter)
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        m(torch.zeros(2, 4, 200))
