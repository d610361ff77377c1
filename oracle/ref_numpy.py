"""CPU oracle for the 21cmVAE predict()/train() hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch numpy restatement of the arithmetic the reference
delegates to TensorFlow/Keras.  It is the checker the parity tests compare the HIP
path against.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package (``21cmvae_amd``) never does.

Parity status
-------------
* pre/post-processing (A12): PINNED -- checked against the reference's own
  ``VeryAccurateEmulator/preprocess.py`` (imported by file path in the build
  container; golden vectors in ``tests/golden/preprocess_golden.npz``).
* dense forward (A2/A3/A4/A7): pinned by the reference's shipped Keras weight
  files (``tests/golden/ae_path_weights.npz``): structure, (in,out) kernel layout and
  the encoder(decoder(z)) ~ z self-consistency of the trained stack.  Bit-level
  outputs of TF's Dense kernels: PARITY UNPINNED (TensorFlow is not installable
  here and the reference ships no golden output vectors).
* loss (A8/A9): formula pinned by ``tests/test_emulator.py:24-33`` of the reference
  (identity restated in tests/test_oracle.py); numerics unpinned.
* Adam (A10), fit loop (A11): Keras 2.7 semantics restated from the published
  algorithm; PARITY UNPINNED at bit level.  Pinned indirectly by the optimizer
  ``iter`` counters in the shipped files (17,568 = 183 x 96; 13,536 = 141 x 96 steps,
  i.e. ceil(N/256) = 96 steps per epoch with the partial last batch kept) and by
  finite-difference / torch-autograd cross-checks of the gradients.

* KL/ELBO + reparameterisation (A13): absent from the reference snapshot; the
  restatement here checks the build-side extension only.  PARITY UNPINNED.

Every function cites the reference lines it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- #
# A12  pre/post-processing  (VeryAccurateEmulator/preprocess.py)
# --------------------------------------------------------------------------- #


def signal_stats(signal_train):
    """Per-bin mean and global scalar std, as preprocess.py:22-23 / :44-45 use them."""
    return np.mean(signal_train, axis=0), np.std(signal_train)


def preproc(signal, signal_train):
    """preprocess.py:4-24: (signal - mean_j) / std, dtype of `signal` preserved."""
    mean, std = signal_stats(signal_train)
    out = np.array(signal, copy=True)
    out -= mean
    out /= std
    return out


def unpreproc(signal, signal_train):
    """preprocess.py:27-46: signal * std + mean_j."""
    mean, std = signal_stats(signal_train)
    out = signal * std
    out += mean
    return out


def _log_cols(p):
    """preprocess.py:74-86: log10 of columns 0..2, fx == 0 -> 1e-6 first; float64 out."""
    p = np.asarray(p)
    q = np.empty(p.shape)  # float64, like np.empty default (preprocess.py:81)
    q[:, :2] = np.log10(p[:, :2])
    fx = p[:, 2].copy()
    fx[fx == 0] = 10 ** (-6)
    q[:, 2] = np.log10(fx)
    q[:, 3:] = p[:, 3:]
    return q


def par_limits(params_train):
    """preprocess.py:89-101: column min/max of the log-transformed training set."""
    t = _log_cols(params_train)
    return np.min(t, axis=0), np.max(t, axis=0)


def par_transform(parameters, params_train):
    """preprocess.py:49-110 (1-D -> (1,7); log10; affine map of train box to [-1,1])."""
    parameters = np.asarray(parameters)
    if parameters.ndim == 1:
        parameters = parameters[None, :]
    lo, hi = par_limits(params_train)
    q = _log_cols(parameters)
    q -= lo
    q /= hi - lo
    q *= 2
    q -= 1
    return q


# --------------------------------------------------------------------------- #
# A1  model factory (emulator.py:12-48)
# --------------------------------------------------------------------------- #


def layer_dims(in_dim, hidden_dims, out_dim):
    """emulator.py:37-47: Dense(h, act) per hidden dim then a linear Dense(out_dim)."""
    return [int(in_dim)] + [int(h) for h in hidden_dims] + [int(out_dim)]


def glorot_uniform(rng, fan_in, fan_out, dtype=np.float32):
    """[K] Keras Dense default kernel init: U(-l, l), l = sqrt(6/(fan_in+fan_out))."""
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


def init_mlp(dims, seed, dtype=np.float32):
    """Kernels (in,out) row-major + zero biases, the layout of the shipped .h5 files."""
    rng = np.random.default_rng(seed)
    Ws, bs = [], []
    for k, n in zip(dims[:-1], dims[1:]):
        Ws.append(glorot_uniform(rng, k, n, dtype))
        bs.append(np.zeros(n, dtype))
    return Ws, bs


def flatten_params(Ws, bs):
    """Keras get_weights() order: W0, b0, W1, b1, ... in one flat vector."""
    return np.concatenate([a.ravel() for W, b in zip(Ws, bs) for a in (W, b)])


def unflatten_params(flat, dims):
    Ws, bs, o = [], [], 0
    for k, n in zip(dims[:-1], dims[1:]):
        Ws.append(flat[o:o + k * n].reshape(k, n)); o += k * n
        bs.append(flat[o:o + n]); o += n
    assert o == flat.size
    return Ws, bs


# --------------------------------------------------------------------------- #
# A2  dense forward (Keras Dense: act(x @ W + b); emulator.py:43,45)
# --------------------------------------------------------------------------- #


def mlp_forward(Ws, bs, x, dtype=np.float64, keep=False):
    """ReLU on hidden layers, identity on the last.  keep=True returns all activations."""
    h = np.asarray(x, dtype=dtype)
    acts = [h]
    L = len(Ws)
    for l, (W, b) in enumerate(zip(Ws, bs)):
        z = h @ W.astype(dtype) + b.astype(dtype)
        h = np.maximum(z, 0) if l < L - 1 else z
        acts.append(h)
    return acts if keep else h


# --------------------------------------------------------------------------- #
# A8/A9  losses (emulator.py:51-83; tf.keras.metrics.mean_squared_error)
# --------------------------------------------------------------------------- #


def relative_mse_row_weight(y_true, signal_train):
    """emulator.py:70-80: loss_i = mean_j (y-p)^2 / amp_i^2,
    amp_i = max_j |y_ij + mean_j/std|.  Returned as w_i = 1/(D amp_i^2) so that
    loss_i = w_i * sum_j (y-p)^2; depends on the target row only."""
    mean, std = signal_stats(signal_train)
    shift = (mean / std).astype(y_true.dtype)
    amp = np.max(np.abs(y_true + shift), axis=1)
    return 1.0 / (y_true.shape[1] * amp.astype(np.float64) ** 2)


def mse_row_weight(y_true):
    """Plain MSE (notebooks/Training.ipynb cell 10): w_i = 1/D."""
    return np.full(y_true.shape[0], 1.0 / y_true.shape[1])


def per_sample_loss(pred, y, w):
    d = pred.astype(np.float64) - y.astype(np.float64)
    return w * np.sum(d * d, axis=1)


def batch_loss_and_grad(pred, y, w, denom=None):
    """[K] batch loss = mean over the batch of per-sample losses; dL/dpred."""
    B = pred.shape[0] if denom is None else denom
    d = pred - y
    loss = float(np.sum(w * np.sum(d.astype(np.float64) ** 2, axis=1)) / B)
    g = (2.0 / B) * w[:, None].astype(pred.dtype) * d
    return loss, g


# --------------------------------------------------------------------------- #
# K3  backward of the dense stack
# --------------------------------------------------------------------------- #


def mlp_backward(Ws, acts, dout):
    """acts = mlp_forward(..., keep=True).  Returns (dWs, dbs, dx)."""
    L = len(Ws)
    dWs, dbs = [None] * L, [None] * L
    dz = dout
    for l in range(L - 1, -1, -1):
        h_in = acts[l]
        dWs[l] = h_in.T @ dz
        dbs[l] = dz.sum(axis=0)
        dh = dz @ Ws[l].astype(dz.dtype).T
        if l > 0:
            dz = dh * (acts[l] > 0)
        else:
            dz = dh
    return dWs, dbs, dz


# --------------------------------------------------------------------------- #
# A10  Adam, Keras 2.7 flavour (epsilon added to the UN-corrected sqrt(v))
# --------------------------------------------------------------------------- #


class AdamState:
    def __init__(self, n, dtype=np.float32, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        self.m = np.zeros(n, dtype)
        self.v = np.zeros(n, dtype)
        self.t = 0
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.dtype = dtype


def adam_alpha(lr, beta1, beta2, t, dtype=np.float32):
    """[K] alpha_t = lr sqrt(1-b2^t)/(1-b1^t), evaluated in the variable dtype."""
    f = dtype
    b1p = f(np.power(f(beta1), f(t)))
    b2p = f(np.power(f(beta2), f(t)))
    return f(f(lr) * f(np.sqrt(f(1) - b2p)) / (f(1) - b1p))


def adam_step(w, g, st: AdamState):
    """[K] ResourceApplyAdam: m += (g-m)(1-b1); v += (g^2-v)(1-b2);
    w -= alpha m/(sqrt(v)+eps)."""
    f = st.dtype
    st.t += 1
    a = adam_alpha(st.lr, st.beta1, st.beta2, st.t, f)
    g = g.astype(f)
    st.m += (g - st.m) * f(f(1) - f(st.beta1))
    st.v += (g * g - st.v) * f(f(1) - f(st.beta2))
    w -= (st.m * a) / (np.sqrt(st.v) + f(st.eps))
    return w


# --------------------------------------------------------------------------- #
# A11  Keras fit() epoch loop (call sites emulator.py:369-378, 739-747, 756-764)
# --------------------------------------------------------------------------- #


def epoch_permutation(n, seed, epoch):
    """Shuffle order shared by the oracle and the device epoch driver: a fresh
    full permutation per epoch from PCG64(seed, epoch).  ([K] Keras reshuffles every
    epoch; its RNG stream is not reproducible outside TF, so the order is ours.)"""
    return np.random.Generator(np.random.PCG64([seed, epoch])).permutation(n).astype(np.int32)


def train_step(Ws, bs, st, x, y, w_row, dtype):
    dims = [Ws[0].shape[0]] + [W.shape[1] for W in Ws]
    acts = mlp_forward(Ws, bs, x, dtype=dtype, keep=True)
    loss, g = batch_loss_and_grad(acts[-1], y.astype(dtype), w_row)
    dWs, dbs, _ = mlp_backward(Ws, acts, g.astype(dtype))
    flat = flatten_params(Ws, bs)
    gflat = flatten_params(dWs, dbs)
    flat = adam_step(flat, gflat, st)
    Ws2, bs2 = unflatten_params(flat, dims)
    return Ws2, bs2, loss, gflat


def evaluate(Ws, bs, x, y, w_row, batch=256, dtype=np.float64):
    """[K] validation pass: sample-weighted mean of batch losses."""
    n, tot = x.shape[0], 0.0
    for s in range(0, n, batch):
        p = mlp_forward(Ws, bs, x[s:s + batch], dtype=dtype)
        tot += float(np.sum(per_sample_loss(p, y[s:s + batch], w_row[s:s + batch])))
    return tot / n


def fit(Ws, bs, st, x, y, w_row, epochs, batch=256, seed=0, val=None, dtype=np.float32,
        start_epoch=0, shuffle=True):
    """Returns (Ws, bs, history) with history = {"loss": [...], "val_loss": [...]}.
    Epoch loss = sum(batch_loss * n_b)/N with the partial last batch kept [K]."""
    n = x.shape[0]
    hist = {"loss": [], "val_loss": []}
    for ep in range(start_epoch, start_epoch + epochs):
        perm = epoch_permutation(n, seed, ep) if shuffle else np.arange(n)
        tot = 0.0
        for s in range(0, n, batch):
            idx = perm[s:s + batch]
            Ws, bs, loss, _ = train_step(Ws, bs, st, x[idx], y[idx], w_row[idx], dtype)
            tot += loss * len(idx)
        hist["loss"].append(tot / n)
        if val is not None:
            hist["val_loss"].append(evaluate(Ws, bs, val[0], val[1], val[2], batch, dtype))
    return Ws, bs, hist


# --------------------------------------------------------------------------- #
# A13  variational latent layer: KL / ELBO, reparameterisation
#      (ABSENT from the reference snapshot -- only the layer name `z_mean` survives in
#      models/autoencoder_based_emulator/encoder.h5; this is the checker of the BUILD-SIDE
#      extension include/v21.h:V21_ACT_GAUSS.  PARITY UNPINNED: no reference arithmetic
#      exists; pinned by finite differences and by "kl_weight = 0, eps = 0 reduces to A7".)
# --------------------------------------------------------------------------- #

_M64 = (1 << 64) - 1


def gauss_eps(seed, step, rows, L, row0=0):
    """Counter-based standard normals, the same stream as csrc/train_kernels.h:gauss_eps:
    splitmix64 finaliser of (seed, step, row, d) -> two 24-bit uniforms -> Box-Muller
    (cosine branch), float32 arithmetic.  Returns (rows, L) float32."""
    r = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
    d = np.arange(L, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        x = (np.uint64(seed & _M64) + np.uint64(step) * np.uint64(0x9E3779B97F4A7C15)
             + r * np.uint64(0xD1B54A32D192ED03) + d * np.uint64(0x8CB92BA72F3D8DD7))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    f = np.float32
    u1 = ((x >> np.uint64(40)) + np.uint64(1)).astype(f) * f(1.0 / 16777216.0)
    u2 = ((x >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(f) * f(1.0 / 16777216.0)
    return (np.sqrt(f(-2.0) * np.log(u1)) * np.cos(f(6.28318530717958647692) * u2)).astype(f)


def vae_forward(Ws, bs, gl, x, eps, dtype=np.float64):
    """Dense stack whose layer `gl` is the (z_mean | z_log_var) head: W[gl] is (K, 2L);
    the next layer sees z = mu + exp(lv/2) eps.  ReLU on every other hidden layer, identity
    on the last.  Returns (acts, mu, lv): acts[l] = input of layer l, acts[-1] = output."""
    h = np.asarray(x, dtype=dtype)
    acts, L, mu, lv = [h], len(Ws), None, None
    for l, (W, b) in enumerate(zip(Ws, bs)):
        z = h @ W.astype(dtype) + b.astype(dtype)
        if l == gl:
            nl = W.shape[1] // 2
            mu, lv = z[:, :nl], z[:, nl:]
            h = mu + np.exp(0.5 * lv) * eps.astype(dtype)
        else:
            h = np.maximum(z, 0) if l < L - 1 else z
        acts.append(h)
    return acts, mu, lv


def vae_loss_and_grads(Ws, bs, gl, x, y, w_row, eps, kl_weight, dtype=np.float64, denom=None):
    """loss = mean_i [ w_i sum_j (p - y)^2 + kl_weight * KL_i ],
    KL_i = -1/2 sum_d (1 + lv - mu^2 - exp lv).  Returns (loss, flat gradient in arena order)."""
    acts, mu, lv = vae_forward(Ws, bs, gl, x, eps, dtype)
    B = x.shape[0] if denom is None else denom
    recon, dz = batch_loss_and_grad(acts[-1], y.astype(dtype), w_row, denom=B)
    kl = -0.5 * np.sum(1.0 + lv - mu * mu - np.exp(lv), axis=1)
    loss = recon + float(kl_weight * np.sum(kl.astype(np.float64)) / B)
    L = len(Ws)
    dWs, dbs = [None] * L, [None] * L
    dz = dz.astype(dtype)
    for l in range(L - 1, -1, -1):
        dWs[l] = acts[l].T @ dz
        dbs[l] = dz.sum(axis=0)
        if l == 0:
            break
        dh = dz @ Ws[l].astype(dtype).T
        if l - 1 == gl:  # through the sampling: d mu = dh + beta mu; d lv = dh eps sd/2 + beta (e^lv - 1)/2
            sd = np.exp(0.5 * lv)
            beta = dtype(kl_weight) / B
            dz = np.concatenate([dh + beta * mu, dh * eps.astype(dtype) * 0.5 * sd + beta * 0.5 * (sd * sd - 1.0)], axis=1)
        else:
            dz = dh * (acts[l] > 0)
    return loss, flatten_params(dWs, dbs)


# --------------------------------------------------------------------------- #
# A3/A4  predict pipelines (emulator.py:383-407, 770-795)
# --------------------------------------------------------------------------- #


def direct_predict(Ws, bs, params, par_train, signal_train, dtype=np.float32):
    x = par_transform(params, par_train)
    p = mlp_forward(Ws, bs, x.astype(np.float32), dtype=dtype).astype(np.float32)
    out = unpreproc(p, signal_train)
    return out[0] if out.shape[0] == 1 else out  # emulator.py:404-407


def ae_predict(em, dec, params, par_train, signal_train, dtype=np.float32):
    x = par_transform(params, par_train)
    z = mlp_forward(em[0], em[1], x.astype(np.float32), dtype=dtype).astype(np.float32)
    p = mlp_forward(dec[0], dec[1], z, dtype=dtype).astype(np.float32)
    out = unpreproc(p, signal_train)
    return out[0] if out.shape[0] == 1 else out  # emulator.py:792-795


# --------------------------------------------------------------------------- #
# A14  error metric (emulator.py:129-192)
# --------------------------------------------------------------------------- #


def error(true_signal, pred_signal, relative=True):
    t = np.atleast_2d(true_signal)
    p = np.atleast_2d(pred_signal)
    err = np.sqrt(np.mean((p - t) ** 2, axis=1))
    if relative:
        err = err / np.max(np.abs(t), axis=1) * 100
    return err
