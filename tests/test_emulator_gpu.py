"""GPU: the reference's class surface (emulator.py) on the MI355X engine.  The tests mirror
the reference's own tests/test_emulator.py and tests/test_preprocess.py on synthetic data
of the same shapes (the real dataset is not redistributable/offline)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, pkg
from oracle import ref_numpy as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    synth = pkg("synth")
    return synth.make_dataset(n_train=3000, n_val=400, n_test=300, seed=1)


def test_gen_model():
    # reference tests/test_emulator.py:12-21
    emulator = pkg("emulator")
    hidden = [32, 64, 256]
    model = emulator._gen_model(7, hidden, 451, "relu")
    all_dims = hidden + [451]
    assert len(model.layers) == len(all_dims)
    for i, layer in enumerate(model.layers):
        assert layer.output_shape[-1] == all_dims[i]
    assert model.count_params() == 7 * 32 + 32 + 32 * 64 + 64 + 64 * 256 + 256 + 256 * 451 + 451


def test_z_nu_and_error():
    emulator = pkg("emulator")
    assert np.isclose(30, emulator.freq2redshift(emulator.redshift2freq(30)))  # test_emulator.py:36-39
    sig = pkg("synth").make_signals(20, seed=1)
    assert np.allclose(emulator.error(sig, sig), 0)                            # :42-47
    nu = emulator.redshift2freq(np.linspace(5, 50, 451))
    e_band = emulator.error(sig, sig + 1.0, relative=False, nu_arr=nu, flow=50, fhigh=100)
    assert e_band.shape == (20,) and np.allclose(e_band, 1.0)
    with pytest.raises(ValueError):
        emulator.error(sig, sig, flow=50)


def test_direct_emulator_predict_contract(data):
    """reference tests/test_emulator.py:55-69 (shape, batched == single to 5e-5) + oracle parity."""
    emulator = pkg("emulator")
    direm = emulator.DirectEmulator(**data)
    pars = direm.par_test[0]
    pred = direm.predict(pars)
    assert pred.shape == direm.signal_test[0].shape and pred.dtype == np.float32
    preds = direm.predict(direm.par_test[:10])
    assert preds.shape == (10, 451)
    assert np.allclose(preds[0], pred, atol=5e-5)
    assert direm.predict(direm.par_test[:1]).shape == (451,)        # any single row is squeezed (:404-405)
    assert direm.predict(list(map(float, pars))).shape == (451,)    # lists are accepted (sample notebook)
    Ws = [l.kernel for l in direm.emulator.layers]; bs = [l.bias for l in direm.emulator.layers]
    ref = ora.direct_predict(Ws, bs, direm.par_test[:50], direm.par_train, direm.signal_train, dtype=np.float64)
    np.testing.assert_allclose(direm.predict(direm.par_test[:50]), ref, atol=2e-5 * float(np.std(direm.signal_train)) + 1e-4, rtol=1e-5)
    err = direm.test_error()
    assert err.shape == (direm.signal_test.shape[0],)
    assert direm.test_error(relative=False, flow=50, fhigh=100).shape == err.shape
    with pytest.raises(NotImplementedError):
        direm.save()
    with pytest.raises(IOError):
        direm.load_model()  # the reference's models/emulator.h5 is not redistributed


def test_direct_emulator_train_returns_keras_history(data):
    emulator, cbm, optm, eng = pkg("emulator"), pkg("callbacks"), pkg("optimizers"), pkg("engine")
    eng.set_random_seed(0)
    direm = emulator.DirectEmulator(hidden_dims=[64, 128], **data)  # the sample notebook's toy model
    direm.emulator.compile(optimizer=optm.Adam(0.01), loss=emulator.relative_mse_loss(direm.signal_train))
    before = direm.test_error().mean()
    es = cbm.EarlyStopping(monitor="val_loss", patience=15, min_delta=1e-5, restore_best_weights=True)
    rl = cbm.ReduceLROnPlateau(monitor="val_loss", factor=0.7, min_delta=1e-4)
    loss, val_loss = direm.train(epochs=12, callbacks=[es, rl], verbose=0)
    assert isinstance(loss, list) and len(loss) == len(val_loss) == 12
    assert all(isinstance(v, float) for v in loss)
    assert loss[-1] < loss[0] and val_loss[-1] < val_loss[0]
    after = direm.test_error().mean()
    assert after < before
    assert direm.emulator.optimizer.iterations == 12 * int(np.ceil(3000 / 256))
    # weights trained on the device are visible through get_weights and drive predict
    Ws = direm.emulator.get_weights()[0::2]; bs = direm.emulator.get_weights()[1::2]
    ref = ora.direct_predict(Ws, bs, direm.par_test[:20], direm.par_train, direm.signal_train, dtype=np.float64)
    np.testing.assert_allclose(direm.predict(direm.par_test[:20]), ref, atol=1e-3, rtol=1e-4)
    with pytest.raises(RuntimeError):
        emulator.DirectEmulator(hidden_dims=[8], **data).train(epochs=1, verbose=0)  # not compiled


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_train_batch_size_4096_through_the_class_surface(precision):
    """``train(..., batch_size=)`` (the reference hard-codes 256, emulator.py:372): BASELINE configs[3]'s per-GPU
    batch of 4,096 rows through DirectEmulator.train -- two full batches and a partial one per epoch -- against the
    float64 oracle driven with the same initial weights and the same shuffles: epoch loss, validation loss,
    optimizer step count."""
    import copy
    emulator, optm, eng, synth = pkg("emulator"), pkg("optimizers"), pkg("engine"), pkg("synth")
    data = synth.make_dataset(n_train=9000, n_val=500, n_test=50, seed=4)
    eng.set_random_seed(11)
    direm = emulator.DirectEmulator(hidden_dims=[64, 128], precision=precision, **data)
    direm.emulator.compile(optimizer=optm.Adam(2e-3), loss=emulator.relative_mse_loss(direm.signal_train))
    W = [a.astype(np.float64) for a in direm.emulator.get_weights()[0::2]]
    b = [a.astype(np.float64) for a in direm.emulator.get_weights()[1::2]]
    rng = copy.deepcopy(eng._rng)  # the shuffles train() is about to draw
    loss, val_loss = direm.train(epochs=2, verbose=0, batch_size=4096)
    assert direm.emulator.optimizer.iterations == 2 * 3
    X = ora.par_transform(data["par_train"], data["par_train"])
    Xv = ora.par_transform(data["par_val"], data["par_train"])
    Y = ora.preproc(data["signal_train"], data["signal_train"]).astype(np.float64)
    Yv = ora.preproc(data["signal_val"], data["signal_train"]).astype(np.float64)
    w = ora.relative_mse_row_weight(Y, data["signal_train"]).astype(np.float64)
    wv = ora.relative_mse_row_weight(Yv, data["signal_train"]).astype(np.float64)
    X, Xv = X.astype(np.float32).astype(np.float64), Xv.astype(np.float32).astype(np.float64)  # Keras casts x to float32 [K]
    sto = ora.AdamState(sum(a.size for a in W) + sum(a.size for a in b), dtype=np.float64, lr=2e-3)
    tol = 5e-5 if precision == "f32" else 5e-3
    for ep in range(2):
        perm = rng.permutation(9000)
        tot = 0.0
        for s0 in range(0, 9000, 4096):
            idx = perm[s0:s0 + 4096]
            W, b, l, _ = ora.train_step(W, b, sto, X[idx], Y[idx], w[idx], np.float64)
            tot += l * len(idx)
        lo = tot / 9000
        vo = ora.evaluate(W, b, Xv, Yv, wv, 4096, np.float64)
        assert abs(loss[ep] - lo) / lo < tol, (ep, loss[ep], lo)
        assert abs(val_loss[ep] - vo) / vo < tol, (ep, val_loss[ep], vo)
    # the autoencoder-based emulator takes the same keyword on both recipes
    aee = emulator.AutoEncoderEmulator(precision="f16", enc_hidden_dims=[64], dec_hidden_dims=[32, 64], em_hidden_dims=[64], **data)
    aee.autoencoder.compile(optimizer=optm.Adam(1e-3), loss=emulator.relative_mse_loss(aee.signal_train))
    aee.emulator.compile(optimizer=optm.Adam(1e-3), loss=emulator.mean_squared_error)
    for joint in (False, True):
        out = aee.train(epochs=1, verbose=0, joint=joint, batch_size=4096)
        assert all(len(h) == 1 and np.isfinite(h[0]) for h in out)
    assert aee.autoencoder.optimizer.iterations == aee.emulator.optimizer.iterations == 2 * 3


def test_autoencoder_emulator_load_predict_shipped_weights(data, shipped):
    """AutoEncoderEmulator.load_model default paths (packaged conversions of the reference's
    files) + predict through the fused chain; reference tests/test_emulator.py:88-102."""
    emulator = pkg("emulator")
    ae_em = emulator.AutoEncoderEmulator(**data)
    ae_em.load_model()
    assert [l.units for l in ae_em.emulator.layers] == [352, 352, 352, 224, 9]
    assert [l.units for l in ae_em.autoencoder.layers] == [352, 9, 32, 352, 451]
    pars = ae_em.par_test[:10]
    preds = ae_em.predict(pars)
    assert preds.shape == (10, 451)
    assert np.allclose(ae_em.predict(pars[0]), preds[0], atol=5e-5)
    ref = ora.ae_predict(shipped["ae_emulator"], shipped["decoder"], pars, ae_em.par_train, ae_em.signal_train, dtype=np.float64)
    np.testing.assert_allclose(preds, ref, atol=2e-5 * float(np.std(ae_em.signal_train)) + 1e-4, rtol=1e-5)
    e1 = ae_em.test_error()
    e2 = ae_em.test_error(use_autoencoder=True)
    assert e1.shape == e2.shape == (ae_em.signal_test.shape[0],)
    # f16 operands: same predictions to well within the emulation-error budget
    ae16 = emulator.AutoEncoderEmulator(precision="f16", **data)
    ae16.load_model()
    d = emulator.error(preds, ae16.predict(pars))
    assert d.mean() < 0.05


@pytest.mark.parametrize("n", [1, 10, 300, 6000])
def test_predict_with_float32_and_float64_parameters_on_the_shipped_weights(shipped, n):
    """The reference floors and takes log10 in the dtype of the parameter array it is handed (preprocess.py:74-78); float32
    `parameters` and float32 `par_train` therefore give other transformed inputs than float64 ones.  Through the class
    surface (AutoEncoderEmulator.predict on the reference's trained weights; 1 / 10 / 300 rows: the few-row route, 6,000:
    the one-launch kernel with the device prologue) every dtype combination must meet oracle.ae_predict -- which follows
    the reference's dtype branch and is pinned to it by reference-generated goldens (tests/test_oracle.py) -- at the
    stated f32 tolerance: atol 2e-5 in pre-processed units."""
    emulator, synth = pkg("emulator"), pkg("synth")
    data = synth.make_dataset(n_train=3000, n_val=50, n_test=max(n, 2), seed=11)
    std = float(np.std(data["signal_train"]))
    for tr_dt in (np.float64, np.float32):
        d = dict(data)
        d["par_train"] = data["par_train"].astype(tr_dt)
        ae_em = emulator.AutoEncoderEmulator(**d)
        ae_em.load_model()
        for in_dt in (np.float64, np.float32):
            pars = data["par_test"][:n].astype(in_dt)
            ref = ora.ae_predict(shipped["ae_emulator"], shipped["decoder"], pars, d["par_train"], d["signal_train"], dtype=np.float64)
            got = ae_em.predict(pars)
            assert got.shape == ref.shape and got.dtype == np.float32
            np.testing.assert_allclose(got, ref, atol=2e-5 * std, rtol=1e-5, err_msg="par_train %s, parameters %s" % (tr_dt.__name__, in_dt.__name__))
    # integer parameters: the documented deviation (taken to float64; the reference would return -inf for fx == 0)
    ints = np.array([[1, 20, 0, 1, 1, 1, 30]] * 3)
    np.testing.assert_array_equal(ae_em.predict(ints), ae_em.predict(ints.astype(np.float64)))
    # float16 parameters: transformed by preprocess.par_transform on the host (the reference's branch for that dtype)
    p16 = data["par_test"][:4].astype(np.float16)
    ref = ora.ae_predict(shipped["ae_emulator"], shipped["decoder"], p16, d["par_train"], d["signal_train"], dtype=np.float64)
    np.testing.assert_allclose(ae_em.predict(p16), ref, atol=2e-5 * std, rtol=1e-5)


def test_autoencoder_emulator_two_phase_training(data):
    emulator, optm, eng = pkg("emulator"), pkg("optimizers"), pkg("engine")
    eng.set_random_seed(1)
    ae_em = emulator.AutoEncoderEmulator(latent_dim=6, enc_hidden_dims=[32], dec_hidden_dims=[16, 32],
                                         em_hidden_dims=[32, 32], **data)
    ae_em.autoencoder.compile(optimizer=optm.Adam(0.001), loss=emulator.relative_mse_loss(ae_em.signal_train))
    ae_em.emulator.compile(optimizer=optm.Adam(0.01), loss=emulator.mean_squared_error)
    out = ae_em.train(epochs=5, verbose=0)
    assert len(out) == 4 and all(len(o) == 5 for o in out)
    ae_loss, ae_val, em_loss, em_val = out
    assert ae_loss[-1] < ae_loss[0] and em_loss[-1] < em_loss[0]
    # decoder trained inside the autoencoder chain is the one predict() uses
    z = ae_em.emulator.predict(pkg("preprocess").par_transform(ae_em.par_test[:5], ae_em.par_train))
    dec = ae_em.autoencoder.decoder.predict(z)
    full = ae_em.predict(ae_em.par_test[:5])
    np.testing.assert_allclose(full, pkg("preprocess").unpreproc(dec, ae_em.signal_train), atol=1e-3, rtol=1e-4)


def test_load_keras_h5_and_alias_package(data):
    import VeryAccurateEmulator as VAE
    assert VAE.emulator.Emulator is VAE.emulator.DirectEmulator
    assert VAE.emulator.VAEEmulator is VAE.emulator.AutoEncoderEmulator
    exp = np.load(os.path.join(GOLDEN, "tiny_h5_expected.npz"))
    d = dict(data); d["signal_train"] = data["signal_train"][:, :5]; d["signal_val"] = data["signal_val"][:, :5]
    d["signal_test"] = data["signal_test"][:, :5]
    em = VAE.emulator.DirectEmulator(hidden_dims=[12], **d)
    em.load_model(os.path.join(GOLDEN, "tiny_keras_model.h5"))
    x = VAE.preprocess.par_transform(em.par_test[:7], em.par_train)
    ref = ora.mlp_forward([exp["W0"], exp["W1"]], [exp["b0"], exp["b1"]], x)
    np.testing.assert_allclose(em.emulator.predict(x), ref, atol=2e-5, rtol=1e-5)
    assert em.predict(em.par_test[:7]).shape == (7, 5)


def test_missing_dataset_is_an_error_not_a_download():
    emulator = pkg("emulator")
    if emulator.load_dataset() is not None:
        pytest.skip("a dataset file is present")
    with pytest.raises(ValueError, match="not found"):
        emulator.DirectEmulator()


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_variational_autoencoder_mode(data, precision):
    """A13 (build-side): AutoEncoder(variational=True, kl_weight=...) trains with sampled
    latents + KL; predict/encoder.predict are deterministic (z = z_mean)."""
    emulator, optm, eng = pkg("emulator"), pkg("optimizers"), pkg("engine")
    eng.set_random_seed(2)
    ae_em = emulator.AutoEncoderEmulator(latent_dim=6, enc_hidden_dims=[32], dec_hidden_dims=[16, 32],
                                         em_hidden_dims=[32, 32], variational=True, kl_weight=1e-4,
                                         precision=precision, **data)
    ae = ae_em.autoencoder
    head = ae.encoder.layers[-1]
    assert isinstance(head, eng.GaussianLatent) and head.kernel.shape == (32, 12) and head.output_shape[-1] == 6
    ae.compile(optimizer=optm.Adam(0.001), loss=emulator.relative_mse_loss(ae_em.signal_train))
    ae_em.emulator.compile(optimizer=optm.Adam(0.01), loss=emulator.mean_squared_error)
    ae_loss, ae_val, em_loss, em_val = ae_em.train(epochs=6, verbose=0)
    assert ae_loss[-1] < ae_loss[0] and ae_val[-1] < ae_val[0] and em_loss[-1] < em_loss[0]
    y = pkg("preprocess").preproc(ae_em.signal_test[:7], ae_em.signal_train)
    z = ae.encoder.predict(y)
    assert z.shape == (7, 6)
    np.testing.assert_array_equal(z, ae.encoder.predict(y))  # deterministic
    tol = 1e-4 if precision == "f32" else 3e-2
    np.testing.assert_allclose(ae.predict(y), ae.decoder.predict(z), atol=tol, rtol=tol)
    assert ae_em.predict(ae_em.par_test[:3]).shape == (3, 451)


def test_f16_training_early_stop_restores_best_weights(data):
    """The chain-kernel path behind the Keras callbacks: EarlyStopping stops a run whose validation loss
    rises (learning rate far too high), restores the best epoch's weights into the device stack (packed
    streams included), and a second fit() continues from them."""
    emulator, cbm, optm, eng = pkg("emulator"), pkg("callbacks"), pkg("optimizers"), pkg("engine")
    eng.set_random_seed(3)
    direm = emulator.DirectEmulator(hidden_dims=[64, 64], precision="f16", **data)
    direm.emulator.compile(optimizer=optm.Adam(0.003), loss=emulator.relative_mse_loss(direm.signal_train))
    l0, v0 = direm.train(epochs=4, verbose=0)
    good = [w.copy() for w in direm.emulator.get_weights()]
    # now wreck it: a huge learning rate makes the validation loss jump; patience 1 stops at once
    direm.emulator.optimizer.lr = 0.5
    es = cbm.EarlyStopping(monitor="val_loss", patience=1, restore_best_weights=True)
    l1, v1 = direm.train(epochs=8, callbacks=[es], verbose=0)
    assert len(l1) < 8 and es.stopped_epoch > 0
    best = int(np.argmin(v1))
    X = pkg("preprocess").par_transform(direm.par_val, direm.par_train)
    Y = pkg("preprocess").preproc(direm.signal_val, direm.signal_train)
    # the restored weights reproduce the best validation loss (f32 evaluation of f16-trained weights)
    got = direm.emulator.evaluate(X, Y, batch_size=256)
    assert abs(got - v1[best]) / v1[best] < 2e-2, (got, v1)
    # and training goes on from there at a sane rate
    direm.emulator.optimizer.lr = 0.001
    l2, v2 = direm.train(epochs=3, verbose=0)
    assert v2[-1] < 1.5 * v1[best]
    assert len(good) == len(direm.emulator.get_weights())


def test_save_keras_h5_and_resume_training(data, tmp_path):
    """Model.save('x.h5') writes a Keras legacy-H5 file (weights, layer names, activations, Adam state) that
    load_model reads back; compiling the loaded model with ITS optimizer resumes training exactly where the
    original would have continued (same weights after the same further epochs)."""
    emulator, optm, eng, h5 = pkg("emulator"), pkg("optimizers"), pkg("engine"), pkg("h5lite")
    eng.set_random_seed(5)
    a = emulator.DirectEmulator(hidden_dims=[48, 32], **data)
    a.emulator.compile(optimizer=optm.Adam(0.002), loss=emulator.relative_mse_loss(a.signal_train))
    a.train(epochs=3, verbose=0)
    p = str(tmp_path / "emulator.h5")
    a.emulator.save(p)
    b = emulator.DirectEmulator(hidden_dims=[48, 32], **data)
    b.load_model(p)
    for wa, wb in zip(a.emulator.get_weights(), b.emulator.get_weights()):
        np.testing.assert_array_equal(wa, wb)
    np.testing.assert_array_equal(a.predict(a.par_test[:9]), b.predict(b.par_test[:9]))
    assert [l.name for l in b.emulator.layers] == [l.name or n for l, n in zip(a.emulator.layers, ["dense", "dense_1", "dense_2"])]
    ob = b.emulator.optimizer
    assert ob is not None and ob.iterations == a.emulator.optimizer.iterations and float(ob.lr) == np.float32(0.002)
    # resume: same shuffles -> identical continuation
    b.emulator.compile(optimizer=ob, loss=emulator.relative_mse_loss(b.signal_train))
    eng.set_random_seed(9); la, _ = a.train(epochs=2, verbose=0)
    eng.set_random_seed(9); lb, _ = b.train(epochs=2, verbose=0)
    np.testing.assert_allclose(lb, la, rtol=1e-6)
    for wa, wb in zip(a.emulator.get_weights(), b.emulator.get_weights()):
        np.testing.assert_allclose(wb, wa, rtol=1e-5, atol=1e-7)
    # a variational encoder keeps its head through the file
    ae = emulator.AutoEncoder(a.signal_train, [16], [16], latent_dim=4, variational=True)
    ae.build((None, 451))
    pe = str(tmp_path / "encoder.h5")
    ae.encoder.save(pe)
    enc = h5.load_model(pe)
    assert isinstance(enc.layers[-1], eng.GaussianLatent) and enc.layers[-1].kernel.shape == (16, 8)
    y = pkg("preprocess").preproc(a.signal_test[:5], a.signal_train)
    np.testing.assert_array_equal(enc.predict(y), ae.encoder.predict(y))
