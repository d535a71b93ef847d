"""CPU oracle for the ExplaiNN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-numpy restatement of the reference's algorithm for the path
`ExplaiNN.forward` + its autograd backward + one Adam step.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this module; the product (`explainn_amd`) never does and fails loudly when its
HIP library is missing.

Pinning: the reference has no tests or golden vectors of its own for this path
(SURVEY.md section 4), so the oracle is pinned by fixtures generated from the
imported reference in the build container (`tools/make_golden.py` ->
`tests/golden/*.npz`, checked by `tests/test_oracle_golden.py`).

Every function cites the reference file:line (relative to
`/root/reference/explainn/`) whose behaviour it restates.  The math is the
straightforward layer-by-layer chain (dense conv, dense batch-norm backward);
none of the algebraic shortcuts of the HIP path are used here, so the two are
independent derivations.

State-dict keys and shapes are the reference's (`architectures/__init__.py:72-104`):
    linears.0.weight (U,4,k)   linears.0.bias (U)          grouped Conv1d(4U->U)
    linears.1.{weight,bias,running_mean,running_var} (U)   BatchNorm1d(U)
    linears.6.weight (100U,n,1) linears.6.bias (100U)      grouped 1x1 conv = per-unit Linear(n->100)
    linears.7.*  (100U)                                    BatchNorm1d(100U)
    linears.10.weight (U,100,1) linears.10.bias (U)        per-unit Linear(100->1)
    linears.11.* (U)                                       BatchNorm1d(U)
    final.weight (T,U)  final.bias (T)                     nn.Linear(U,T)
"""
import math

import numpy as np

BN_EPS = 1e-5        # architectures/__init__.py:79,90,99 (torch default / explicit 1e-05)
BN_MOMENTUM = 0.1    # idem
POOL = 7             # architectures/__init__.py:81  MaxPool1d(7, 7)
FC_HIDDEN = 100      # architectures/__init__.py:86
DROPOUT_P = 0.3      # architectures/__init__.py:92

PARAM_KEYS = (
    "linears.0.weight", "linears.0.bias", "linears.1.weight", "linears.1.bias",
    "linears.6.weight", "linears.6.bias", "linears.7.weight", "linears.7.bias",
    "linears.10.weight", "linears.10.bias", "linears.11.weight", "linears.11.bias",
    "final.weight", "final.bias",
)
BUFFER_KEYS = (
    "linears.1.running_mean", "linears.1.running_var", "linears.1.num_batches_tracked",
    "linears.7.running_mean", "linears.7.running_var", "linears.7.num_batches_tracked",
    "linears.11.running_mean", "linears.11.running_var", "linears.11.num_batches_tracked",
)


# ----------------------------------------------------------------------------
# input encoding
# ----------------------------------------------------------------------------
def one_hot_encode(seq):
    """sequence/__init__.py:8-28 -- rows A,C,G,T; anything else is an all-zero column."""
    seq = seq.upper()
    out = np.zeros((4, len(seq)), dtype=np.float64)
    for i, ch in enumerate(seq):
        j = "ACGT".find(ch)
        if j >= 0:
            out[j, i] = 1.0
    return out


def one_hot_encode_many(seqs):
    """sequence/__init__.py:4-6"""
    return np.array([one_hot_encode(s) for s in seqs])


def rc_one_hot_encoding(x):
    """sequence/__init__.py:59-61 -- reverse complement = flip both axes of the last two dims."""
    return x[..., ::-1, ::-1]


def pooled_len(sequence_length, kernel_size):
    """architectures/__init__.py:69 -- n = floor((L-k+1)/7)."""
    return int(math.floor((sequence_length - kernel_size + 1) / 7.0))


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------
def _bn_forward(x, axes, gamma, beta, rmean, rvar, training, shape):
    """torch.nn.BatchNorm1d semantics as used at architectures/__init__.py:79,90,99:
    biased variance normalises; running_var receives the unbiased estimate."""
    if training:
        cnt = 1
        for a in axes:
            cnt *= x.shape[a]
        mu = x.mean(axis=axes)
        var = x.var(axis=axes)                      # biased
        new_rmean = (1 - BN_MOMENTUM) * rmean + BN_MOMENTUM * mu
        new_rvar = (1 - BN_MOMENTUM) * rvar + BN_MOMENTUM * var * (cnt / max(cnt - 1, 1))
    else:
        mu, var = rmean, rvar
        new_rmean, new_rvar = rmean, rvar
    inv = 1.0 / np.sqrt(var + x.dtype.type(BN_EPS))
    xhat = (x - mu.reshape(shape)) * inv.reshape(shape)
    y = gamma.reshape(shape) * xhat + beta.reshape(shape)
    return y, xhat, inv, new_rmean.astype(rmean.dtype), new_rvar.astype(rvar.dtype)


def forward(sd, x, training=False, dropout_mask=None, p=DROPOUT_P, dtype=np.float32,
            return_cache=False):
    """ExplaiNN.forward, architectures/__init__.py:109-114 over the Sequential at :72-102.

    sd            state dict (numpy arrays, reference keys)
    x             (B,4,L) one-hot (N = zero column)
    dropout_mask  (B,100U) keep-mask in {0,1}; None -> no dropout (p treated as 0)
    returns logits (B,T) [, cache, new_buffers]
    """
    f = lambda a: np.asarray(a, dtype=dtype)
    x = f(x)
    B, _, L = x.shape
    W = f(sd["linears.0.weight"]); U, _, k = W.shape
    Lo = L - k + 1
    n = pooled_len(L, k)
    if training and B == 1:
        # torch raises in BN2 (architectures/__init__.py:90); reason for train.py:297-302
        raise ValueError("Expected more than 1 value per channel when training, "
                         "got input size [1, %d, 1]" % (FC_HIDDEN * U))
    # linears[0]: grouped Conv1d -- every unit sees the same 4 rows (x.repeat at :111)
    win = np.lib.stride_tricks.sliding_window_view(x, k, axis=2)          # (B,4,Lo,k)
    c = np.einsum("bapj,uaj->bup", win, W, optimize=True).astype(dtype)
    c = c + f(sd["linears.0.bias"])[None, :, None]
    # linears[1]: BatchNorm1d(U) over (b,p)
    y, chat, inv1, rm1, rv1 = _bn_forward(
        c, (0, 2), f(sd["linears.1.weight"]), f(sd["linears.1.bias"]),
        f(sd["linears.1.running_mean"]), f(sd["linears.1.running_var"]), training, (1, U, 1))
    # linears[2]: ExpAct (:12-19)
    e = np.exp(y)
    # linears[3]: MaxPool1d(7,7), floor mode, first index on ties
    ew = e[:, :, :POOL * n].reshape(B, U, n, POOL)
    arg = ew.argmax(axis=3)
    q = np.take_along_axis(ew, arg[..., None], axis=3)[..., 0]           # (B,U,n)
    # linears[4..6]: Flatten/UnSqueeze + grouped 1x1 conv == per-unit Linear(n->100)
    V1 = f(sd["linears.6.weight"]).reshape(U, FC_HIDDEN, n)
    h = np.einsum("buw,urw->bur", q, V1, optimize=True).astype(dtype)
    h = h + f(sd["linears.6.bias"]).reshape(1, U, FC_HIDDEN)
    # linears[7]: BatchNorm1d(100U) over b
    h2 = h.reshape(B, U * FC_HIDDEN)
    y2, hhat, inv2, rm2, rv2 = _bn_forward(
        h2, (0,), f(sd["linears.7.weight"]), f(sd["linears.7.bias"]),
        f(sd["linears.7.running_mean"]), f(sd["linears.7.running_var"]), training,
        (1, U * FC_HIDDEN))
    # linears[8]: ReLU ; linears[9]: Dropout(0.3) (train only)
    r2 = np.maximum(y2, 0)
    if training and dropout_mask is not None:
        scale = dtype(1.0 / (1.0 - p))
        keep = f(dropout_mask).reshape(B, U * FC_HIDDEN) * scale
        a = r2 * keep
    else:
        keep = None
        a = r2
    # linears[10]: per-unit Linear(100->1)
    V2 = f(sd["linears.10.weight"]).reshape(U, FC_HIDDEN)
    z = np.einsum("bur,ur->bu", a.reshape(B, U, FC_HIDDEN), V2).astype(dtype)
    z = z + f(sd["linears.10.bias"])[None, :]
    # linears[11..13]: BatchNorm1d(U) over b, ReLU, Flatten
    y3, zhat, inv3, rm3, rv3 = _bn_forward(
        z, (0,), f(sd["linears.11.weight"]), f(sd["linears.11.bias"]),
        f(sd["linears.11.running_mean"]), f(sd["linears.11.running_var"]), training, (1, U))
    o = np.maximum(y3, 0)
    # final: nn.Linear(U,T) (:104)
    Wf = f(sd["final.weight"])
    logits = o @ Wf.T + f(sd["final.bias"])[None, :]
    if not return_cache:
        return logits
    cache = dict(x=x, win=win, c=c, chat=chat, inv1=inv1, e=e, arg=arg, q=q, V1=V1, hhat=hhat,
                 inv2=inv2, y2=y2, keep=keep, a=a, V2=V2, zhat=zhat, inv3=inv3, y3=y3, o=o,
                 Wf=Wf, n=n, sd=sd, dtype=dtype, acts=e)
    nb = {
        "linears.1.running_mean": rm1, "linears.1.running_var": rv1,
        "linears.7.running_mean": rm2, "linears.7.running_var": rv2,
        "linears.11.running_mean": rm3, "linears.11.running_var": rv3,
    }
    for key in ("linears.1", "linears.7", "linears.11"):
        nbt = np.asarray(sd[key + ".num_batches_tracked"], dtype=np.int64)
        nb[key + ".num_batches_tracked"] = nbt + (1 if training else 0)
    return logits, cache, nb


def unit_activations(sd, x, dtype=np.float32):
    """`model.linears[:3](x_rep)` in eval mode, as test.py:159-160 extracts it:
    exp(BN(conv)) per position, shape (B,U,Lo)."""
    _, cache, _ = forward(sd, x, training=False, dtype=dtype, return_cache=True)
    return cache["acts"]


def unit_outputs(sd, x, dtype=np.float32):
    """`model.linears(x_rep)` in eval mode (test.py:151): per-unit outputs (B,U)."""
    _, cache, _ = forward(sd, x, training=False, dtype=dtype, return_cache=True)
    return cache["o"]


def predict_fwd_rev(sd, x, dtype=np.float32):
    """predict.py:75-94 -- [Fwd, Rev, Mean, Max] of eval-mode logits, shape (B,T,4) float64."""
    fwd = forward(sd, x, dtype=dtype)[:, :, None].astype(np.float32)
    rev = forward(sd, np.ascontiguousarray(rc_one_hot_encoding(x)), dtype=dtype)[:, :, None]
    rev = rev.astype(np.float32)
    fr = np.concatenate((fwd, rev), axis=2)
    out = np.concatenate((fwd, rev, fr.mean(axis=2, keepdims=True),
                          fr.max(axis=2, keepdims=True)), axis=2)
    return out.astype(np.float64)


def pwm_scan(pwms, x, scoring="sum", dtype=np.float64):
    """The reference `PWM.forward` (architectures/__init__.py:157-168): valid cross-correlation of
    each (4,k) matrix with x and with its reverse complement (both axes flipped), then the max or
    the sum over all 2*(L-k+1) window scores -> (B,G).  Bias is zero by construction (:151)."""
    pwms = np.asarray(pwms, dtype=dtype)
    x = np.asarray(x, dtype=dtype)
    G, _, k = pwms.shape
    Lo = x.shape[2] - k + 1

    def scan(xx):
        win = np.stack([xx[:, :, j:j + Lo] for j in range(k)], axis=3)       # (B,4,Lo,k)
        return np.einsum("balk,gak->bgl", win, pwms)

    o = np.concatenate((scan(x), scan(x[:, ::-1, ::-1])), axis=2)
    return o.max(axis=2) if scoring == "max" else o.sum(axis=2)


# ----------------------------------------------------------------------------
# losses (architectures/__init__.py:446-456), mean reduction
# ----------------------------------------------------------------------------
def bce_with_logits(logits, y):
    """nn.BCEWithLogitsLoss(): mean of max(x,0) - x*y + log1p(exp(-|x|)); returns (loss, dlogits)."""
    x = logits
    y = np.asarray(y, dtype=x.dtype)
    loss = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
    sig = 1.0 / (1.0 + np.exp(-x))
    return loss.mean(), ((sig - y) / x.size).astype(x.dtype)


def mse(logits, y):
    """nn.MSELoss(): mean (x-y)^2; returns (loss, dlogits)."""
    y = np.asarray(y, dtype=logits.dtype)
    d = logits - y
    return (d * d).mean(), (2.0 * d / d.size).astype(logits.dtype)


# ----------------------------------------------------------------------------
# backward (what autograd computes for the chain above; train mode)
# ----------------------------------------------------------------------------
def _bn_backward(dy, xhat, gamma, inv, axes, shape):
    """d/dx of train-mode batch norm: (gamma*inv) * (dy - mean dy - xhat*mean(dy*xhat))."""
    dgamma = (dy * xhat).sum(axis=axes)
    dbeta = dy.sum(axis=axes)
    m1 = dy.mean(axis=axes).reshape(shape)
    m2 = (dy * xhat).mean(axis=axes).reshape(shape)
    dx = (gamma * inv).reshape(shape) * (dy - m1 - xhat * m2)
    return dx, dgamma, dbeta


def backward(cache, dlogits, freeze_top_n_filters=0):
    """Gradients of all 14 parameters for a train-mode forward (selene/__init__.py:290-291).

    freeze_top_n_filters: rows [0:n) of the filter gradient are zeroed, as the hook at
    selene/__init__.py:509-515 does."""
    sd = cache["sd"]; dt = cache["dtype"]
    f = lambda a: np.asarray(a, dtype=dt)
    B = dlogits.shape[0]
    U = cache["V2"].shape[0]; n = cache["n"]
    g = {}
    dlogits = f(dlogits)
    g["final.weight"] = dlogits.T @ cache["o"]
    g["final.bias"] = dlogits.sum(axis=0)
    do = dlogits @ cache["Wf"]
    d3 = do * (cache["y3"] > 0)
    dz, g["linears.11.weight"], g["linears.11.bias"] = _bn_backward(
        d3, cache["zhat"], f(sd["linears.11.weight"]), cache["inv3"], (0,), (1, U))
    a3 = cache["a"].reshape(B, U, FC_HIDDEN)
    g["linears.10.weight"] = np.einsum("bu,bur->ur", dz, a3).reshape(U, FC_HIDDEN, 1)
    g["linears.10.bias"] = dz.sum(axis=0)
    da = (dz[:, :, None] * cache["V2"][None]).reshape(B, U * FC_HIDDEN)
    if cache["keep"] is not None:
        da = da * cache["keep"]
    d2 = da * (cache["y2"] > 0)
    dh, g["linears.7.weight"], g["linears.7.bias"] = _bn_backward(
        d2, cache["hhat"], f(sd["linears.7.weight"]), cache["inv2"], (0,), (1, U * FC_HIDDEN))
    dh3 = dh.reshape(B, U, FC_HIDDEN)
    g["linears.6.weight"] = np.einsum("bur,buw->urw", dh3, cache["q"], optimize=True
                                      ).reshape(U * FC_HIDDEN, n, 1)
    g["linears.6.bias"] = dh.sum(axis=0)
    dq = np.einsum("bur,urw->buw", dh3, cache["V1"], optimize=True)
    # max-pool routing, then exp' = e
    Lo = cache["c"].shape[2]
    de = np.zeros((B, U, Lo), dtype=dt)
    dew = de[:, :, :POOL * n].reshape(B, U, n, POOL)
    np.put_along_axis(dew, cache["arg"][..., None], dq[..., None], axis=3)
    de[:, :, :POOL * n] = dew.reshape(B, U, POOL * n)
    dy1 = de * cache["e"]
    dc, g["linears.1.weight"], g["linears.1.bias"] = _bn_backward(
        dy1, cache["chat"], f(sd["linears.1.weight"]), cache["inv1"], (0, 2), (1, U, 1))
    gW = np.einsum("bup,bapj->uaj", dc, cache["win"], optimize=True)
    if freeze_top_n_filters > 0:
        gW[:freeze_top_n_filters] = 0
    g["linears.0.weight"] = gW
    g["linears.0.bias"] = dc.sum(axis=(0, 2))
    return {k_: np.asarray(v, dtype=dt) for k_, v in g.items()}


# ----------------------------------------------------------------------------
# optimiser: torch.optim.Adam defaults (architectures/__init__.py:463-464)
# ----------------------------------------------------------------------------
def adam_init(sd):
    return {k: dict(step=0, exp_avg=np.zeros_like(sd[k]), exp_avg_sq=np.zeros_like(sd[k]))
            for k in PARAM_KEYS}


def adam_step(sd, grads, state, lr=0.003, betas=(0.9, 0.999), eps=1e-8):
    """One torch.optim.Adam step (no weight decay, no amsgrad), in place on sd/state."""
    b1, b2 = betas
    for k in PARAM_KEYS:
        st = state[k]; g = grads[k].reshape(sd[k].shape).astype(sd[k].dtype)
        st["step"] += 1
        st["exp_avg"] = b1 * st["exp_avg"] + (1 - b1) * g
        st["exp_avg_sq"] = b2 * st["exp_avg_sq"] + (1 - b2) * g * g
        bc1 = 1 - b1 ** st["step"]; bc2 = 1 - b2 ** st["step"]
        denom = np.sqrt(st["exp_avg_sq"]) / math.sqrt(bc2) + eps
        sd[k] = (sd[k] - (lr / bc1) * st["exp_avg"] / denom).astype(sd[k].dtype)


def train_step(sd, state, x, y, loss="binary", dropout_mask=None, lr=0.003, dtype=np.float32,
               freeze_top_n_filters=0):
    """selene/__init__.py:283-292: train-mode forward, loss, backward, Adam; returns
    (loss, logits, grads); sd/state updated in place, BN buffers included."""
    logits, cache, nb = forward(sd, x, training=True, dropout_mask=dropout_mask, dtype=dtype,
                                return_cache=True)
    lfun = bce_with_logits if loss == "binary" else mse
    lval, dlogits = lfun(logits, y)
    grads = backward(cache, dlogits, freeze_top_n_filters)
    adam_step(sd, grads, state, lr=lr)
    for k, v in nb.items():
        sd[k] = v
    return float(lval), logits, grads


# ----------------------------------------------------------------------------
# helpers for tests / benchmarks
# ----------------------------------------------------------------------------
def random_state_dict(U, k, L, T, seed=0, perturb=True, dtype=np.float32):
    """A random state dict of the reference's shapes (NOT torch's init; parity tests that need
    the reference's init load the golden state dicts instead)."""
    rng = np.random.default_rng(seed)
    n = pooled_len(L, k)
    u = lambda shape, bound: rng.uniform(-bound, bound, size=shape).astype(dtype)
    sd = {
        "linears.0.weight": u((U, 4, k), 1 / math.sqrt(4 * k)),
        "linears.0.bias": u((U,), 1 / math.sqrt(4 * k)),
        "linears.6.weight": u((FC_HIDDEN * U, n, 1), 1 / math.sqrt(max(n, 1))),
        "linears.6.bias": u((FC_HIDDEN * U,), 1 / math.sqrt(max(n, 1))),
        "linears.10.weight": u((U, FC_HIDDEN, 1), 0.1),
        "linears.10.bias": u((U,), 0.1),
        "final.weight": u((T, U), 1 / math.sqrt(U)),
        "final.bias": u((T,), 1 / math.sqrt(U)),
    }
    for key, c in (("linears.1", U), ("linears.7", FC_HIDDEN * U), ("linears.11", U)):
        if perturb:
            sd[key + ".weight"] = (1 + 0.5 * rng.standard_normal(c)).astype(dtype)
            sd[key + ".bias"] = (0.3 * rng.standard_normal(c)).astype(dtype)
            sd[key + ".running_mean"] = (0.2 * rng.standard_normal(c)).astype(dtype)
            sd[key + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(dtype)
        else:
            sd[key + ".weight"] = np.ones(c, dtype); sd[key + ".bias"] = np.zeros(c, dtype)
            sd[key + ".running_mean"] = np.zeros(c, dtype)
            sd[key + ".running_var"] = np.ones(c, dtype)
        sd[key + ".num_batches_tracked"] = np.asarray(0, dtype=np.int64)
    return sd


def random_onehot(B, L, seed=1, n_frac=0.0, dtype=np.float32):
    """Synthetic uniform one-hot batch; a fraction n_frac of positions become N (zero columns)."""
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 4, size=(B, L))
    x = np.zeros((B, 4, L), dtype=dtype)
    np.put_along_axis(x, idx[:, None, :], 1.0, axis=1)
    if n_frac > 0:
        holes = rng.random((B, L)) < n_frac
        x[np.broadcast_to(holes[:, None, :], x.shape)] = 0
    return x
