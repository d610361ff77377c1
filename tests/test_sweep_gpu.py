"""Sweep (BASELINE configs[4]): several models trained in lock step with grouped launches must
give what training each model on its own gives -- the reference trains one model per fit()
(emulator.py:739-747), so that IS the expected result."""
import numpy as np
import pytest

from conftest import pkg
from oracle import ref_numpy as ora

pytestmark = pytest.mark.gpu

CONFIGS = [dict(latent_dim=4, enc_hidden_dims=[32], dec_hidden_dims=[16, 32]),
           dict(latent_dim=9, enc_hidden_dims=[64], dec_hidden_dims=[32, 96]),
           dict(latent_dim=12, enc_hidden_dims=[96], dec_hidden_dims=[48, 64])]


def _models(sig, seed, precision="f32"):
    emulator, optm, eng = pkg("emulator"), pkg("optimizers"), pkg("engine")
    eng.set_random_seed(seed)
    out = []
    for c in CONFIGS:
        ae = emulator.AutoEncoder(sig, **c)
        ae.precision = precision
        ae.build((None, 451))
        ae.compile(optimizer=optm.Adam(1e-3), loss=emulator.relative_mse_loss(sig))
        out.append(ae)
    return out


@pytest.mark.parametrize("n,batch,precision", [(300, 128, "f32"), (700, 600, "f32"), (700, 200, "f32"), (300, 128, "f16"), (700, 600, "bf16")])
def test_sweep_equals_individual_fits(n, batch, precision):
    synth, eng, sweep = pkg("synth"), pkg("engine"), pkg("sweep")
    sig = synth.make_signals(n, seed=3)
    val = synth.make_signals(60, seed=4)
    y, yv = ora.preproc(sig, sig), ora.preproc(val, sig)
    solo = _models(sig, 11, precision)
    grouped = _models(sig, 11, precision)
    # f32: same kernels, same order (steps of <= 256 rows: the grouped launches of train_chain32s.h / dw_adam32.h, whose
    # row blocks may be 8 rows where a single model's are 4 -- the same sums per element, another grouping of the rows in
    # the batch loss; larger steps: the per-layer path).  f16/bf16 (chain kernel): the grouped weight-gradient launch may pick
    # another tile size than a single model's, i.e. another summation order of rounded products
    ltol, wtol = (2e-5, 2e-6) if precision == "f32" else (2e-3, 2e-3)
    for a, b in zip(solo, grouped):
        for wa, wb in zip(a.get_weights(), b.get_weights()):
            np.testing.assert_array_equal(wa, wb)
    hs = []
    for m in solo:
        eng.set_random_seed(5)  # same shuffle sequence for every run
        hs.append(m.fit(y, y, batch_size=batch, epochs=3, validation_data=(yv, yv)))
    eng.set_random_seed(5)
    hg = sweep.fit_models(grouped, y, y, batch_size=batch, epochs=3, validation_data=(yv, yv))
    for a, b, ha, hb in zip(solo, grouped, hs, hg):
        np.testing.assert_allclose(hb.history["loss"], ha.history["loss"], rtol=ltol)
        np.testing.assert_allclose(hb.history["val_loss"], ha.history["val_loss"], rtol=ltol)
        for wa, wb in zip(a.get_weights(), b.get_weights()):
            np.testing.assert_allclose(wb, wa, atol=wtol, rtol=1e-4)
        assert b.optimizer.iterations == a.optimizer.iterations == 3 * -(-n // batch)


def test_sweep_member_stops_early_others_continue():
    synth, cbm, sweep = pkg("synth"), pkg("callbacks"), pkg("sweep")
    sig = synth.make_signals(200, seed=8)
    y = ora.preproc(sig, sig)
    models = _models(sig, 2)

    class StopAt(cbm.Callback):
        def __init__(self, at):
            super().__init__(); self.at = at

        def on_epoch_end(self, epoch, logs=None):
            if epoch == self.at:
                self.model.stop_training = True
    hists = sweep.fit_models(models, y, y, batch_size=64, epochs=4, validation_data=(y, y),
                             callbacks=[[StopAt(0)], [], [StopAt(2)]])  # model 0 (the data holder) leaves first
    assert [len(h.history["loss"]) for h in hists] == [1, 4, 3]
    assert hists[1].history["loss"][-1] < hists[1].history["loss"][0]


def test_sweep_argument_errors(ctx):
    native = pkg("_native")
    a = native.Trainer(native.Stack(ctx, [8, 6, 8], [1, 0]), "f32", 32)
    b = native.Trainer(native.Stack(ctx, [8, 5, 4, 8], [1, 1, 0]), "f32", 32)   # other depth
    c = native.Trainer(native.Stack(ctx, [8, 7, 8], [1, 0]), "f32", 64)         # other max_batch
    d = native.Trainer(native.Stack(ctx, [8, 7, 8], [1, 0]), "f32", 32)
    for bad in ([a, b], [a, c], [a, a]):
        with pytest.raises(native.EngineError):
            native.Sweep(bad)
    sw = native.Sweep([a, d])
    with pytest.raises(native.EngineError):
        sw.run_epoch(None, 16)  # trainer 0 has no training set


def test_sweep_group_sizes(ctx):
    """One model is a valid group; 64 is the most one group holds (r5: the whole of BASELINE configs[4]; 16 until r4) -- in
    f16 (grouped chain launch with a model's row blocks on one XCD label + grouped gradient / Adam launch) and in f32
    (train_chain32s_group_kernel + dwadam32_group_kernel): model 0 and model 63 of the big group train as they do alone; 65
    are refused."""
    native = pkg("_native")
    rng = np.random.default_rng(0)
    x = rng.normal(size=(64, 12)).astype(np.float32)
    w = np.full(64, 1.0 / 12, np.float32)

    def make(k, prec="f16"):
        st = native.Stack(ctx, [12, 8 + k, 12], [1, 0])
        Ws, bs = ora.init_mlp([12, 8 + k, 12], seed=k)
        st.set_weights(ora.flatten_params(Ws, bs))
        tr = native.Trainer(st, prec, 64)
        tr.set_adam(lr=1e-3)
        return tr
    solo = make(0); solo.set_data(0, x, None, w)
    ref = [solo.run_epoch(None, 32) for _ in range(2)]
    one = make(0); one.set_data(0, x, None, w)
    sw1 = native.Sweep([one])
    got = [sw1.run_epoch(None, 32)[0] for _ in range(2)]
    np.testing.assert_allclose(got, ref, rtol=1e-6)
    for prec in ("f16", "f32"):
        solo0 = make(0, prec); solo0.set_data(0, x, None, w)
        solo63 = make(63, prec); solo63.set_data(0, x, None, w)
        r0, r63 = solo0.run_epoch(None, 32), solo63.run_epoch(None, 32)
        many = [make(k, prec) for k in range(64)]
        many[0].set_data(0, x, None, w)
        sw64 = native.Sweep(many)
        l64 = sw64.run_epoch(None, 32)
        assert len(l64) == 64 and np.all(np.isfinite(l64))
        tol = 1e-3 if prec == "f16" else 2e-5
        np.testing.assert_allclose(l64[0], r0, rtol=tol)    # members of the big group = the same models alone
        np.testing.assert_allclose(l64[63], r63, rtol=tol)
        np.testing.assert_allclose(many[63].stack.get_weights(), solo63.stack.get_weights(), atol=3e-3 if prec == "f16" else 5e-6)
    with pytest.raises(native.EngineError):
        native.Sweep(many + [make(64, "f32")])


@pytest.mark.parametrize("precision", ["f16", "f32"])
def test_sweep_of_variational_models_equals_individual_training(precision):
    """Variational autoencoders of different latent widths and KL weights in one sweep (chain path: f16, and the grouped
    launches of the small-batch f32 chain): the per-model noise streams (seed, step, row) make the grouped run reproduce
    the individual runs."""
    synth, eng, sweep, emulator, optm = pkg("synth"), pkg("engine"), pkg("sweep"), pkg("emulator"), pkg("optimizers")
    sig = synth.make_signals(260, seed=3)
    y = ora.preproc(sig, sig)
    cfgs = [dict(latent_dim=4, kl_weight=1e-3), dict(latent_dim=9, kl_weight=1e-4), dict(latent_dim=16, kl_weight=0.0)]

    def models():
        eng.set_random_seed(21)
        out = []
        for c in cfgs:
            ae = emulator.AutoEncoder(sig, enc_hidden_dims=[48], dec_hidden_dims=[32, 48], variational=True, **c)
            ae.precision = precision
            ae.build((None, 451))
            ae._vae_seed = 1000 + c["latent_dim"]  # same noise stream in both runs
            ae.compile(optimizer=optm.Adam(1e-3), loss=emulator.relative_mse_loss(sig))
            out.append(ae)
        return out
    solo, grouped = models(), models()
    hs = []
    for m in solo:
        eng.set_random_seed(5)
        hs.append(m.fit(y, y, batch_size=128, epochs=3))
    eng.set_random_seed(5)
    hg = sweep.fit_models(grouped, y, y, batch_size=128, epochs=3)
    for a, b, ha, hb in zip(solo, grouped, hs, hg):
        ltol, wtol = (3e-3, 3e-3) if precision == "f16" else (2e-5, 2e-6)
        np.testing.assert_allclose(hb.history["loss"], ha.history["loss"], rtol=ltol)
        for wa, wb in zip(a.get_weights(), b.get_weights()):
            np.testing.assert_allclose(wb, wa, atol=wtol, rtol=1e-3 if precision == "f16" else 1e-4)


@pytest.mark.parametrize("prec", ["f16", "f32"])
def test_sweep_on_two_streams_equals_one_stream_bit_for_bit(ctx, prec, monkeypatch):
    """r5: a sweep of >= 16 members on one rank runs as TWO half-groups on two streams, the second one launch behind the first
    (csrc/api_sweep.hip: one half's chain launch runs while the other half's gradient / Adam launch waits for HBM).  The
    members are independent models and every member sees the same kernels on the same data: losses and weights after three
    epochs (a permuted row table, a partial last batch) are IDENTICAL to the one-stream form (V21_SWEEP_STREAMS=1)."""
    native = pkg("_native")
    rng = np.random.default_rng(3)
    n, batch = 700, 256
    x = rng.normal(size=(n, 33)).astype(np.float32)
    w = np.full(n, 1.0 / 33, np.float32)
    perm = rng.permutation(n).astype(np.int32)

    def run(streams):
        monkeypatch.setenv("V21_SWEEP_STREAMS", streams)
        trs = []
        for k in range(17):
            dims = [33, 16 + 24 * k, 4 + k, 8 + 16 * k, 33]
            Ws, bs = ora.init_mlp(dims, seed=70 + k)
            st = native.Stack(ctx, dims, [1, 0, 1, 0])
            st.set_weights(ora.flatten_params(Ws, bs))
            tr = native.Trainer(st, prec, batch)
            tr.set_adam(lr=1e-3 * (1 + k % 3))
            trs.append(tr)
        trs[0].set_data(0, x, None, w)
        sw = native.Sweep(trs)
        losses = [sw.run_epoch(perm, batch) for _ in range(3)]
        return losses, [t.stack.get_weights() for t in trs], [t.get_state()[0] for t in trs]
    l1, w1, i1 = run("1")
    l2, w2, i2 = run("2")
    assert l1 == l2 and i1 == i2 == [9] * 17
    for a, b in zip(w1, w2):
        np.testing.assert_array_equal(a, b)
    assert np.all(np.isfinite(np.array(l2)))
