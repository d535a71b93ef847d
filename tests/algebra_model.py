"""Numpy model of the ALGEBRA the HIP kernels use, kept in tests/ as a design
check: it must reproduce the layer-by-layer oracle to rounding in fp64.  It mirrors the kernel
pipeline step by step (same intermediate quantities, same names as the .hip sources):

  pack -> pair counts -> gram -> prep1 -> conv_pool -> qmoments -> prep2 -> fc_fwd -> head_fwd
  head_bwd -> passA -> mid_bwd -> passB -> conv_bwd -> fin_bwd

Nothing here is product code.
"""
import numpy as np

EPS = 1e-5
POOL = 7
H = 100


def codes_from_onehot(x):
    s = np.full((x.shape[0], x.shape[2]), 4, dtype=np.int64)
    for a in range(4):
        s[x[:, a, :] == 1] = a
    return s


def gram_from_codes(s, k):
    """pair counts cnt[d][a][a'][q] -> m (4k), G (4k x 4k); index (a,j) -> a*k + j."""
    B, L = s.shape
    Lo = L - k + 1
    N1 = B * Lo
    cnt = np.zeros((k, 4, 4, L), dtype=np.int64)
    for d in range(k):
        s0 = s[:, :L - d]; s1 = s[:, d:]
        for a in range(4):
            for a2 in range(4):
                cnt[d, a, a2, :L - d] = ((s0 == a) & (s1 == a2)).sum(axis=0)
    G = np.zeros((4 * k, 4 * k))
    for a in range(4):
        for j in range(k):
            for a2 in range(4):
                for j2 in range(j, k):
                    v = cnt[j2 - j, a, a2, j:j + Lo].sum()
                    G[a * k + j, a2 * k + j2] = v
                    G[a2 * k + j2, a * k + j] = v
    m = np.diag(G).copy()
    return m / N1, G / N1, N1


def train_forward_backward(sd, x, dlogits_fn, keep=None, p=0.3):
    """Returns logits, grads (reference keys), new running stats -- all via the kernel algebra."""
    f = lambda a: np.asarray(a, dtype=np.float64)
    x = f(x)
    B, _, L = x.shape
    W = f(sd["linears.0.weight"]); U, _, k = W.shape
    Lo = L - k + 1; n = Lo // POOL
    s = codes_from_onehot(x)
    m, G, N1 = gram_from_codes(s, k)
    w = W.reshape(U, 4 * k)
    cb = f(sd["linears.0.bias"]); g1 = f(sd["linears.1.weight"]); b1 = f(sd["linears.1.bias"])
    # ---- prep1 ----
    Gw = w @ G                      # (U,4k)
    mug = w @ m
    var1 = (w * Gw).sum(1) - mug ** 2
    sig1 = np.sqrt(var1 + EPS)
    alpha = g1 / sig1
    sh = b1 - alpha * mug
    new = {"linears.1.running_mean": 0.9 * f(sd["linears.1.running_mean"]) + 0.1 * (cb + mug),
           "linears.1.running_var": 0.9 * f(sd["linears.1.running_var"]) + 0.1 * var1 * N1 / (N1 - 1)}
    # ---- conv_pool: raw gather sums (no bias), sign-aware pooling ----
    Wp = np.concatenate([W, np.zeros((U, 1, k))], axis=1)          # code 4 -> zeros
    g = np.zeros((B, U, Lo))
    for j in range(k):
        g += Wp[:, s[:, j:j + Lo], j].transpose(1, 0, 2)
    gw = g[:, :, :POOL * n].reshape(B, U, n, POOL)
    sgn = np.where(alpha >= 0, 1.0, -1.0)[None, :, None, None]
    idx = (gw * sgn).argmax(axis=3)
    ext = np.take_along_axis(gw, idx[..., None], axis=3)[..., 0]
    q = np.exp(alpha[None, :, None] * ext + sh[None, :, None])      # (B,U,n)
    # ---- qmoments (shifted by sequence 0) + prep2 ----
    s0 = q[0]
    S1 = (q - s0).sum(0)                                            # (U,n)
    S2 = np.einsum("buw,buv->uwv", q - s0, q - s0)
    qbar = s0 + S1 / B
    C = S2 / B - np.einsum("uw,uv->uwv", S1 / B, S1 / B)
    V1 = f(sd["linears.6.weight"]).reshape(U, H, n); c1 = f(sd["linears.6.bias"]).reshape(U, H)
    g2 = f(sd["linears.7.weight"]).reshape(U, H); b2 = f(sd["linears.7.bias"]).reshape(U, H)
    mu2 = c1 + np.einsum("urw,uw->ur", V1, qbar)
    var2 = np.einsum("urw,uwv,urv->ur", V1, C, V1)
    sig2 = np.sqrt(var2 + EPS)
    A2 = (g2 / sig2)[:, :, None] * V1
    sh2 = b2 - np.einsum("urw,uw->ur", A2, qbar)
    new["linears.7.running_mean"] = (0.9 * f(sd["linears.7.running_mean"]) + 0.1 * mu2.reshape(-1))
    new["linears.7.running_var"] = (0.9 * f(sd["linears.7.running_var"])
                                    + 0.1 * var2.reshape(-1) * B / (B - 1))
    # ---- fc_fwd ----
    y2 = np.einsum("urw,buw->bur", A2, q) + sh2[None]
    sc = 1.0 / (1.0 - p) if keep is not None else 1.0
    kp = f(keep).reshape(B, U, H) if keep is not None else np.ones((B, U, H))
    bit = (y2 > 0) * (kp > 0)
    a = y2 * bit * sc
    V2 = f(sd["linears.10.weight"]).reshape(U, H); c2 = f(sd["linears.10.bias"])
    z = np.einsum("bur,ur->bu", a, V2)                              # c2 cancels in train BN3
    # ---- head_fwd ----
    g3 = f(sd["linears.11.weight"]); b3 = f(sd["linears.11.bias"])
    mu3 = z.mean(0); var3 = z.var(0); sig3 = np.sqrt(var3 + EPS)
    zhat = (z - mu3) / sig3
    y3 = g3 * zhat + b3
    o = np.maximum(y3, 0)
    new["linears.11.running_mean"] = 0.9 * f(sd["linears.11.running_mean"]) + 0.1 * (mu3 + c2)
    new["linears.11.running_var"] = (0.9 * f(sd["linears.11.running_var"])
                                     + 0.1 * var3 * B / (B - 1))
    Wf = f(sd["final.weight"]); bf = f(sd["final.bias"])
    logits = o @ Wf.T + bf
    # ================= backward =================
    dl = f(dlogits_fn(logits))
    gr = {"final.weight": dl.T @ o, "final.bias": dl.sum(0)}
    do = dl @ Wf
    d3 = do * (y3 > 0)
    gr["linears.11.weight"] = (d3 * zhat).sum(0); gr["linears.11.bias"] = d3.sum(0)
    dz = (g3 / sig3) * (d3 - d3.mean(0) - zhat * (d3 * zhat).mean(0))
    gr["linears.10.bias"] = np.zeros(U)
    # ---- passA ----
    e = dz[:, :, None] * bit                                        # (B,U,H)
    EQ = np.einsum("bur,buw->urw", e, q)
    Se = e.sum(0)                                                   # (U,H)
    # ---- mid_bwd ----
    gr["linears.10.weight"] = (sc * (np.einsum("urw,urw->ur", A2, EQ) + sh2 * Se)).reshape(U, H, 1)
    dbeta2 = sc * V2 * Se
    dgamma2 = sc * V2 / sig2 * np.einsum("urw,urw->ur", V1, EQ - Se[:, :, None] * qbar[:, None, :])
    gr["linears.7.bias"] = dbeta2.reshape(-1); gr["linears.7.weight"] = dgamma2.reshape(-1)
    md2 = dbeta2 / B; md2h = dgamma2 / B
    HQ = (B / sig2)[:, :, None] * np.einsum("urw,uwv->urv", V1, C)
    dV1 = (g2 / sig2)[:, :, None] * (sc * V2[:, :, None] * EQ
                                     - md2[:, :, None] * B * qbar[:, None, :]
                                     - md2h[:, :, None] * HQ)
    gr["linears.6.weight"] = dV1.reshape(U * H, n, 1)
    gr["linears.6.bias"] = np.zeros(U * H)
    T = sc * V2[:, :, None] * A2                                    # (U,H,n)
    k0 = np.einsum("urw,ur->uw", A2, md2)
    M = np.einsum("ur,urv,urw->uvw", md2h / sig2, V1, A2)           # M[u][w'][w]
    k0p = k0 - np.einsum("uv,uvw->uw", qbar, M)
    # ---- passB ----
    dq = np.einsum("bur,urw->buw", e, T) - k0p[None] - np.einsum("buv,uvw->buw", q, M)
    dy = dq * q
    chat = (ext - mug[None, :, None]) / sig1[None, :, None]
    S1b = dy.sum((0, 2)); S2b = (dy * chat).sum((0, 2))
    gr["linears.1.bias"] = S1b; gr["linears.1.weight"] = S2b
    # ---- conv_bwd scatter ----
    Dsp = np.zeros((U, 4, k))
    pstar = POOL * np.arange(n)[None, None, :] + idx                # (B,U,n)
    for j in range(k):
        cj = s[np.arange(B)[:, None, None], pstar + j]              # (B,U,n)
        for a in range(4):
            Dsp[:, a, j] = (dy * (cj == a)).sum((0, 2))
    # ---- fin_bwd ----
    mk = m.reshape(4, k)
    gr["linears.0.weight"] = alpha[:, None, None] * (
        Dsp - S1b[:, None, None] * mk[None]
        - (S2b / sig1)[:, None, None] * (Gw.reshape(U, 4, k) - mug[:, None, None] * mk[None]))
    gr["linears.0.bias"] = np.zeros(U)
    return logits, gr, new
