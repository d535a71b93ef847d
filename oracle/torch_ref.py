"""Stock-PyTorch CPU restatement of ExplaiNN's train step -- TEST/BASELINE INFRASTRUCTURE.

It dispatches the same ATen CPU kernels the reference module does (grouped conv1d on the
`x.repeat(1,U,1)` input, batch_norm, exp, max_pool1d, dropout, grouped 1x1 convs, linear; reference
explainn/architectures/__init__.py:72-114), written functionally over a plain dict of tensors with
the reference's state_dict keys.  Used only (a) by tests, pinned against tests/golden/*.npz, and
(b) by bench.py's `cpu_baseline` leg, where it is timed on the GPU box's host cores ("port").
"""
import math

import torch
import torch.nn.functional as F

PARAMS = ("linears.0.weight", "linears.0.bias", "linears.1.weight", "linears.1.bias",
          "linears.6.weight", "linears.6.bias", "linears.7.weight", "linears.7.bias",
          "linears.10.weight", "linears.10.bias", "linears.11.weight", "linears.11.bias",
          "final.weight", "final.bias")


def init_state(U, k, L, T, seed=0):
    """Default-PyTorch-style initialisation (uniform +-1/sqrt(fan_in)), reference shapes."""
    g = torch.Generator().manual_seed(seed)
    n = (L - k + 1) // 7

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    sd = {
        "linears.0.weight": uni((U, 4, k), 4 * k), "linears.0.bias": uni((U,), 4 * k),
        "linears.6.weight": uni((100 * U, n, 1), n), "linears.6.bias": uni((100 * U,), n),
        "linears.10.weight": uni((U, 100, 1), 100), "linears.10.bias": uni((U,), 100),
        "final.weight": uni((T, U), U), "final.bias": uni((T,), U),
    }
    for key, c in (("linears.1", U), ("linears.7", 100 * U), ("linears.11", U)):
        sd[key + ".weight"] = torch.ones(c); sd[key + ".bias"] = torch.zeros(c)
        sd[key + ".running_mean"] = torch.zeros(c); sd[key + ".running_var"] = torch.ones(c)
        sd[key + ".num_batches_tracked"] = torch.tensor(0)
    return sd


def forward(sd, x, training, p=0.3, keep_mask=None):
    """keep_mask: optional (B, 100U) 0/1 tensor replacing dropout's own draw (what nn.Dropout does
    with its mask: multiply and rescale by 1/(1-p))."""
    U = sd["linears.0.weight"].shape[0]

    def bn(t, key):
        return F.batch_norm(t, sd[key + ".running_mean"], sd[key + ".running_var"],
                            sd[key + ".weight"], sd[key + ".bias"], training, 0.1, 1e-5)

    h = F.conv1d(x.repeat(1, U, 1), sd["linears.0.weight"], sd["linears.0.bias"], groups=U)
    h = torch.exp(bn(h, "linears.1"))
    h = F.max_pool1d(h, 7, 7).flatten(1).unsqueeze(-1)
    h = F.conv1d(h, sd["linears.6.weight"], sd["linears.6.bias"], groups=U)
    if keep_mask is not None and training:
        h = F.relu(bn(h, "linears.7")) * keep_mask.to(h.dtype).unsqueeze(-1) / (1.0 - p)
    else:
        h = F.dropout(F.relu(bn(h, "linears.7")), p, training)
    h = F.conv1d(h, sd["linears.10.weight"], sd["linears.10.bias"], groups=U)
    h = F.relu(bn(h, "linears.11")).flatten(1)
    return F.linear(h, sd["final.weight"], sd["final.bias"])


def train_step(sd, x, y, loss="binary", p=0.3, keep_mask=None):
    """train-mode forward + loss + backward; returns (loss, logits, grads dict)."""
    leaves = {k: sd[k].requires_grad_(True) for k in PARAMS}
    for v in leaves.values():
        v.grad = None
    logits = forward(sd, x, True, p, keep_mask)
    lval = (F.binary_cross_entropy_with_logits(logits, y) if loss == "binary"
            else F.mse_loss(logits, y))
    lval.backward()
    return lval.detach(), logits.detach(), {k: v.grad for k, v in leaves.items()}
