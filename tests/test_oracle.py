"""CPU tests of the oracle: it must agree with everything the reference itself pins
(golden vectors generated from the reference's preprocess.py, its shipped weights,
the formula identities of its own tests, its optimizer step counters) before the
GPU path is compared with it."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, pkg
from oracle import ref_numpy as ora


@pytest.fixture(scope="module")
def pg():
    return np.load(os.path.join(GOLDEN, "preprocess_golden.npz"))


def test_par_transform_matches_reference_golden(pg):
    out = ora.par_transform(pg["par_in"], pg["par_train"])
    assert out.dtype == np.float64  # preprocess.py:81
    np.testing.assert_array_equal(out, pg["par_out"])
    np.testing.assert_array_equal(ora.par_transform(pg["par_in"][5], pg["par_train"]), pg["par_out_1d"])
    np.testing.assert_array_equal(
        ora.par_transform([0.0003, 4.2, 0.0, 0.055, 1.0, 0.1, 10.0], pg["par_train"]), pg["par_out_list"])


F32_CASES = [  # (golden key, parameters, params_train): every dtype combination the reference's branch sees
    ("par_out_in32_tr32", lambda g: g["par_in"].astype(np.float32), lambda g: g["par_train"].astype(np.float32)),
    ("par_out_in32_tr64", lambda g: g["par_in"].astype(np.float32), lambda g: g["par_train"]),
    ("par_out_in64_tr32", lambda g: g["par_in"], lambda g: g["par_train"].astype(np.float32)),
    ("par_out_1d_in32_tr32", lambda g: g["par_in"].astype(np.float32)[3], lambda g: g["par_train"].astype(np.float32)),
    ("par_out_1d_in32_tr64", lambda g: g["par_in"].astype(np.float32)[5], lambda g: g["par_train"]),
    ("par_train32_out", lambda g: g["par_train"].astype(np.float32), lambda g: g["par_train"].astype(np.float32)),
]


@pytest.mark.parametrize("key,par,train", F32_CASES, ids=[c[0] for c in F32_CASES])
def test_par_transform_float32_inputs_match_reference_golden(pg, key, par, train):
    """preprocess.py:74-78 / 89-93: floor and log10 in the dtype of the array handed over, float64 result.
    Oracle AND product, bit for bit against outputs of the reference run in the build container."""
    pp = pkg("preprocess")
    for fn in (ora.par_transform, pp.par_transform):
        out = fn(par(pg), train(pg))
        assert out.dtype == np.float64
        np.testing.assert_array_equal(out, pg[key])
    assert not np.array_equal(pg["par_out_in32_tr32"], pg["par_out"])  # the float32 branch IS a different result


def test_par_transform_integer_array_is_the_documented_deviation(pg):
    """an integer `parameters` array truncates the 1e-6 floor to 0 in the reference (-inf); the
    product takes non-floating arrays to float64 first.  Lists of floats behave like the reference."""
    pp = pkg("preprocess")
    ints = np.array([[1, 5, 0, 1, 1, 1, 20]])
    out = pp.par_transform(ints, pg["par_train"])
    assert np.all(np.isfinite(out))
    np.testing.assert_array_equal(out, pp.par_transform(ints.astype(np.float64), pg["par_train"]))


def test_par_transform_train_box_is_unit_box(pg):
    # reference tests/test_preprocess.py:21-26
    t = ora.par_transform(pg["par_train"], pg["par_train"])
    np.testing.assert_allclose(t.max(axis=0), 1.0)
    np.testing.assert_allclose(t.min(axis=0), -1.0)
    np.testing.assert_array_equal(t, pg["par_train_out"])


def test_preproc_unpreproc_match_reference_golden(pg):
    pre = ora.preproc(pg["sig_in"], pg["sig_train"])
    assert pre.dtype == np.float32  # dtype preserved, preprocess.py:21-23
    np.testing.assert_array_equal(pre, pg["pre_out"])
    np.testing.assert_array_equal(ora.unpreproc(pre, pg["sig_train"]), pg["unpre_out"])
    np.testing.assert_array_equal(ora.preproc(pg["sig_in"].astype(np.float64), pg["sig_train"]), pg["pre_out_f64"])
    # reference tests/test_preprocess.py:12-18
    full = ora.preproc(pg["sig_train"], pg["sig_train"])
    np.testing.assert_allclose(full.mean(axis=0), 0.0, atol=1e-3)
    np.testing.assert_allclose(ora.unpreproc(full, pg["sig_train"]), pg["sig_train"], atol=5e-5)


def test_product_preprocess_equals_reference_golden(pg):
    pp = pkg("preprocess")
    np.testing.assert_array_equal(pp.par_transform(pg["par_in"], pg["par_train"]), pg["par_out"])
    np.testing.assert_array_equal(pp.par_transform(pg["par_in"][5], pg["par_train"]), pg["par_out_1d"])
    np.testing.assert_array_equal(pp.preproc(pg["sig_in"], pg["sig_train"]), pg["pre_out"])
    np.testing.assert_array_equal(pp.unpreproc(pg["pre_out"], pg["sig_train"]), pg["unpre_out"])
    # cached statistics must notice an in-place edit of the training array
    tr = pg["sig_train"].copy()
    a = pp.preproc(pg["sig_in"], tr)
    tr *= 2.0
    b = pp.preproc(pg["sig_in"], tr)
    assert not np.array_equal(a, b)
    np.testing.assert_array_equal(b, ora.preproc(pg["sig_in"], tr))


def test_shipped_weights_structure(shipped):
    d = shipped["raw"]
    Ws, bs = shipped["ae_emulator"]
    assert [W.shape for W in Ws] == [(7, 352), (352, 352), (352, 352), (352, 224), (224, 9)]
    assert sum(W.size + b.size for W, b in zip(Ws, bs)) == 332425  # notebooks/Training.ipynb cell 9
    We, be = shipped["encoder"]
    Wd, bd = shipped["decoder"]
    assert sum(W.size + b.size for W, b in zip(We, be)) == 162281
    assert sum(W.size + b.size for W, b in zip(Wd, bd)) == 171139
    assert bool(d["autoencoder/equals_encoder_plus_decoder"])
    # Keras kept the partial last batch: 96 steps per epoch
    assert int(d["ae_emulator/adam_iter"]) == 183 * 96 and int(d["autoencoder/adam_iter"]) == 141 * 96
    assert float(d["ae_emulator/adam_epsilon"]) == 1e-7


def test_shipped_stack_self_consistency(shipped):
    """KAT on the trained weights: encoder(decoder(z)) reproduces the latent z that the
    latent emulator predicts (layout, activation placement and bias order all right)."""
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(512, 7))
    z = ora.mlp_forward(*shipped["ae_emulator"], x)
    rec = ora.mlp_forward(*shipped["encoder"], ora.mlp_forward(*shipped["decoder"], z))
    rel = np.sqrt(np.mean((rec - z) ** 2)) / np.sqrt(np.mean(z ** 2))
    assert rel < 0.06, rel
    # fp32 evaluation of the same stack stays within the stated fp32 tolerance
    p64 = ora.mlp_forward(*shipped["decoder"], z)
    p32 = ora.mlp_forward(*shipped["decoder"], z.astype(np.float32), dtype=np.float32)
    np.testing.assert_allclose(p32, p64, atol=2e-5, rtol=1e-5)


def test_relative_mse_identity():
    """reference tests/test_emulator.py:24-33: loss == mse / max|signal/std|^2."""
    synth = pkg("synth")
    sig = synth.make_signals(64, seed=5)
    y_true = ora.preproc(sig[:10], sig)
    y_pred = ora.preproc(sig[-10:], sig)
    w = ora.relative_mse_row_weight(y_true, sig)
    loss = ora.per_sample_loss(y_pred, y_true, w)
    mse = np.mean((y_true.astype(np.float64) - y_pred) ** 2, axis=1)
    amp = np.max(np.abs(sig[:10] / np.std(sig)), axis=1)
    np.testing.assert_allclose(loss, mse / amp ** 2, rtol=2e-5)


def test_error_metric():
    synth = pkg("synth")
    sig = synth.make_signals(16, seed=6)
    np.testing.assert_array_equal(ora.error(sig, sig), 0.0)  # tests/test_emulator.py:42-47
    e = ora.error(sig[0], sig[0] + 1.0)
    np.testing.assert_allclose(e, 100.0 / np.max(np.abs(sig[0])), rtol=1e-6)


def test_backward_matches_finite_differences():
    rng = np.random.default_rng(3)
    dims = [7, 12, 9, 5]
    Ws, bs = ora.init_mlp(dims, seed=4, dtype=np.float64)
    bs = [rng.normal(size=b.shape) * 0.1 for b in bs]
    x = rng.normal(size=(6, 7)); y = rng.normal(size=(6, 5)); w = rng.uniform(0.5, 2.0, size=6)
    acts = ora.mlp_forward(Ws, bs, x, keep=True)
    loss, g = ora.batch_loss_and_grad(acts[-1], y, w)
    dWs, dbs, _ = ora.mlp_backward(Ws, acts, g)
    flat = ora.flatten_params(Ws, bs)
    gflat = ora.flatten_params(dWs, dbs)
    eps = 1e-6
    for i in rng.choice(flat.size, size=25, replace=False):
        f2 = flat.copy(); f2[i] += eps
        lp, _ = ora.batch_loss_and_grad(ora.mlp_forward(*ora.unflatten_params(f2, dims), x), y, w)
        f2[i] -= 2 * eps
        lm, _ = ora.batch_loss_and_grad(ora.mlp_forward(*ora.unflatten_params(f2, dims), x), y, w)
        np.testing.assert_allclose(gflat[i], (lp - lm) / (2 * eps), rtol=1e-4, atol=1e-8)


def test_backward_and_adam_match_torch():
    """Independent second implementation: torch autograd for the gradient, and the Keras
    epsilon placement restated by hand (torch.optim.Adam places epsilon differently)."""
    torch = pytest.importorskip("torch")
    dims = [7, 16, 8, 11]
    Ws, bs = ora.init_mlp(dims, seed=9, dtype=np.float64)
    rng = np.random.default_rng(10)
    x = rng.normal(size=(32, 7)); y = rng.normal(size=(32, 11)); w = rng.uniform(0.1, 1.0, size=32)
    tW = [torch.tensor(W, requires_grad=True) for W in Ws]
    tb = [torch.tensor(b, requires_grad=True) for b in bs]
    h = torch.tensor(x)
    for l in range(3):
        h = h @ tW[l] + tb[l]
        if l < 2:
            h = torch.relu(h)
    loss_t = (torch.tensor(w) * ((h - torch.tensor(y)) ** 2).sum(dim=1)).mean()
    loss_t.backward()
    acts = ora.mlp_forward(Ws, bs, x, keep=True)
    loss, g = ora.batch_loss_and_grad(acts[-1], y, w)
    dWs, dbs, _ = ora.mlp_backward(Ws, acts, g)
    np.testing.assert_allclose(loss, loss_t.item(), rtol=1e-12)
    for l in range(3):
        np.testing.assert_allclose(dWs[l], tW[l].grad.numpy(), rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(dbs[l], tb[l].grad.numpy(), rtol=1e-10, atol=1e-14)
    # three Adam steps in float64 against the closed form
    st = ora.AdamState(4, dtype=np.float64, lr=0.01)
    wv = np.array([1.0, -2.0, 0.5, 3.0]); gv = np.array([0.1, -0.2, 0.3, 0.0])
    m = np.zeros(4); v = np.zeros(4); ref = wv.copy()
    for t in range(1, 4):
        m = 0.9 * m + 0.1 * gv; v = 0.999 * v + 0.001 * gv * gv
        ref -= 0.01 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-7)
        wv = ora.adam_step(wv, gv, st)
    np.testing.assert_allclose(wv, ref, rtol=1e-12)


def test_fit_loop_keras_bookkeeping():
    """partial last batch kept, sample-weighted epoch loss, one entry per epoch."""
    dims = [7, 8, 5]
    Ws, bs = ora.init_mlp(dims, seed=1, dtype=np.float64)
    rng = np.random.default_rng(2)
    x = rng.normal(size=(70, 7)); y = rng.normal(size=(70, 5)); w = ora.mse_row_weight(y)
    st = ora.AdamState(ora.flatten_params(Ws, bs).size, dtype=np.float64, lr=1e-2)
    Ws2, bs2, hist = ora.fit(Ws, bs, st, x, y, w, epochs=3, batch=32, seed=7, val=(x, y, w), dtype=np.float64)
    assert st.t == 3 * 3 and len(hist["loss"]) == 3 and len(hist["val_loss"]) == 3
    assert hist["loss"][-1] < hist["loss"][0]
    p = ora.epoch_permutation(70, 7, 0)
    assert sorted(p.tolist()) == list(range(70))
    assert not np.array_equal(p, ora.epoch_permutation(70, 7, 1))


def test_c_oracle_agrees_with_numpy_oracle(shipped):
    """oracle/mlp_oracle.c (gcc) vs oracle/ref_numpy.py on the reference's trained decoder."""
    import ctypes as C
    import subprocess
    from conftest import ROOT
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    Ws, bs = shipped["decoder"]
    flat = ora.flatten_params(Ws, bs).astype(np.float32)
    dims = (C.c_int * 4)(9, 32, 352, 451); act = (C.c_int * 3)(1, 1, 0)
    x = np.random.default_rng(0).normal(size=(40, 9)).astype(np.float32)
    y = np.empty((40, 451), np.float32)
    fp = C.POINTER(C.c_float)
    assert lib.oracle_mlp_forward(flat.ctypes.data_as(fp), dims, act, 3, x.ctypes.data_as(fp), C.c_long(40),
                                  y.ctypes.data_as(fp), 1) == 0
    np.testing.assert_allclose(y, ora.mlp_forward(Ws, bs, x), atol=2e-6, rtol=1e-6)
    # Keras-Adam step, f32
    rng = np.random.default_rng(1)
    w = rng.normal(size=100).astype(np.float32); g = rng.normal(size=100).astype(np.float32)
    st = ora.AdamState(100, dtype=np.float32, lr=1e-2)
    ref = w.copy()
    m = np.zeros(100, np.float32); v = np.zeros(100, np.float32)
    lib.oracle_adam_step.argtypes = [fp, fp, fp, fp, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, C.c_long]
    for t in (1, 2, 3):
        ref = ora.adam_step(ref, g, st)
        lib.oracle_adam_step(w.ctypes.data_as(fp), g.ctypes.data_as(fp), m.ctypes.data_as(fp), v.ctypes.data_as(fp),
                             100, 1e-2, 0.9, 0.999, 1e-7, t)
    np.testing.assert_allclose(w, ref, rtol=2e-6, atol=1e-7)


# ---- A13: variational latent layer (build-side extension; no reference arithmetic) -------
def test_gauss_eps_is_standard_normal_and_counter_based():
    e = ora.gauss_eps(seed=42, step=7, rows=4096, L=9)
    assert e.dtype == np.float32 and e.shape == (4096, 9)
    assert abs(float(e.mean())) < 0.02 and abs(float(e.std()) - 1.0) < 0.02
    # counter-based: a row block drawn with row0 equals the same rows of the big draw
    np.testing.assert_array_equal(ora.gauss_eps(42, 7, 100, 9, row0=1000), e[1000:1100])
    assert not np.array_equal(ora.gauss_eps(42, 8, 16, 9), e[:16])


def _vae_toy(seed=0):
    rng = np.random.default_rng(seed)
    shapes = [(12, 8), (8, 6), (3, 8), (8, 12)]  # layer 1 = (z_mean | z_log_var) head, latent 3
    Ws = [rng.normal(size=s) * 0.3 for s in shapes]
    bs = [rng.normal(size=s[1]) * 0.1 for s in shapes]
    x = rng.normal(size=(5, 12))
    return Ws, bs, x, np.full(5, 1.0 / 12)


def test_vae_gradients_match_finite_differences():
    Ws, bs, x, w = _vae_toy()
    eps = ora.gauss_eps(1, 0, 5, 3).astype(np.float64)
    loss, g = ora.vae_loss_and_grads(Ws, bs, 1, x, x, w, eps, 0.3)
    flat = ora.flatten_params(Ws, bs)

    def f(fl):
        o, W2, b2 = 0, [], []
        for W, b in zip(Ws, bs):
            W2.append(fl[o:o + W.size].reshape(W.shape)); o += W.size
            b2.append(fl[o:o + b.size]); o += b.size
        return ora.vae_loss_and_grads(W2, b2, 1, x, x, w, eps, 0.3)[0]
    idx = np.random.default_rng(1).choice(flat.size, 60, replace=False)
    for i in idx:
        d = np.zeros_like(flat); d[i] = 1e-6
        assert abs((f(flat + d) - f(flat - d)) / 2e-6 - g[i]) < 1e-7


def test_vae_without_noise_and_kl_is_the_plain_autoencoder():
    """kl_weight = 0, eps = 0 must reduce exactly to A7 (emulator.py:517): same loss, same
    gradients on the z_mean half, zero gradient on the z_log_var half."""
    Ws, bs, x, w = _vae_toy(3)
    loss, g = ora.vae_loss_and_grads(Ws, bs, 1, x, x, w, np.zeros((5, 3)), 0.0)
    Wp = [Ws[0], Ws[1][:, :3], Ws[2], Ws[3]]
    bp = [bs[0], bs[1][:3], bs[2], bs[3]]
    acts = [x]
    for l, (W, b) in enumerate(zip(Wp, bp)):
        z = acts[-1] @ W + b
        acts.append(z if l in (1, 3) else np.maximum(z, 0))
    lo, dz = ora.batch_loss_and_grad(acts[-1], x, w)
    assert abs(loss - lo) < 1e-14
    # backward of the plain stack (latent layer linear)
    dW1 = None
    for l in range(3, -1, -1):
        if l == 1:
            dW1 = acts[1].T @ dz
        dh = dz @ Wp[l].T
        dz = dh if l - 1 == 1 else dh * (acts[l] > 0)
    o = Ws[0].size + bs[0].size
    gW1 = g[o:o + Ws[1].size].reshape(8, 6)
    np.testing.assert_allclose(gW1[:, :3], dW1, atol=1e-14)
    assert np.all(gW1[:, 3:] == 0)


def test_kink_adjusted_oracle_recovers_a_relu_on_the_other_side_of_zero():
    """tests/helpers.py: the gradient check the GPU tests use for f32 accepts a difference from the float64 oracle only when it
    is the oracle's own gradient with d relu / dz taken the other way at pre-activations that are zero to rounding -- and only
    then.  Built here on the CPU: a stack whose unit (layer 0, row 3, unit 2) has z = +1e-9; the same stack with z = -1e-9 gives
    the 'device' gradient (everything else identical to ~1e-9)."""
    from helpers import oracle_step, per_layer_gradient_check
    rng = np.random.default_rng(0)
    dims, act = [5, 8, 6, 3], [1, 1, 0]
    Ws, bs = ora.init_mlp(dims, seed=4)
    bs = [rng.normal(scale=0.1, size=b.shape).astype(np.float64) for b in bs]
    Ws = [W.astype(np.float64) for W in Ws]
    x = rng.uniform(-1, 1, size=(20, 5))
    y = rng.normal(size=(20, 3))
    w = rng.uniform(0.5, 1.5, size=20) / 3
    r0, u0 = 3, 2

    def with_z(target):
        b0 = bs[0].copy()
        b0[u0] += target - (x[r0] @ Ws[0][:, u0] + b0[u0])
        return [b0] + bs[1:]
    bs_pos, bs_neg = with_z(1e-9), with_z(-1e-9)
    _, go = oracle_step(Ws, bs_pos, act, x, y, w)
    _, g_dev = oracle_step(Ws, bs_neg, act, x, y, w)
    assert np.abs(g_dev - go).max() > 1e-4 * np.abs(go).max()          # the flip is visible ...
    ok, note = per_layer_gradient_check(dims, act, Ws, bs_pos, x, g_dev, go, "f32", tgt=y, w=w)
    assert ok and "1 ReLU(s) at their kink" in note and "layer 0 row 3 unit 2" in note, note   # ... and explained
    ok, note = per_layer_gradient_check(dims, act, Ws, bs_pos, x, g_dev, go, "f32")            # without the data to show it: refused
    assert not ok, note
    bad = g_dev.copy(); bad[7] *= 1.01                                   # a wrong element is not a kink
    ok, note = per_layer_gradient_check(dims, act, Ws, bs_pos, x, bad, go, "f32", tgt=y, w=w)
    assert not ok, note
    g_row = go.copy()                                                    # one row's contribution missing from the first layer
    g_row[:5 * 8] -= (np.outer(x[7], np.ones(8)) * 1e-3).ravel()
    ok, note = per_layer_gradient_check(dims, act, Ws, bs_pos, x, g_row, go, "f32", tgt=y, w=w)
    assert not ok, note
