"""CPU: host-side training logic (callbacks, optimizer config, losses) against the
known answers recorded in the reference's notebooks."""
import numpy as np
import pytest

from conftest import pkg

cbm = pkg("callbacks")
optm = pkg("optimizers")


class FakeModel:
    def __init__(self, lr):
        self.optimizer = optm.Adam(lr)
        self.stop_training = False
        self.w = [np.zeros(3)]

    def get_weights(self):
        return [a.copy() for a in self.w]

    def set_weights(self, w):
        self.w = [a.copy() for a in w]


def _plateau_lrs(lr0, factor, min_lr, n):
    m = FakeModel(lr0)
    cb = cbm.ReduceLROnPlateau(monitor="val_loss", patience=1, factor=factor, min_delta=0.0, min_lr=min_lr)
    cb.set_model(m)
    cb.on_train_begin()
    out = []
    cb.on_epoch_end(0, {"val_loss": 1.0})
    for e in range(1, n + 1):  # never improves -> one reduction per epoch
        cb.on_epoch_end(e, {"val_loss": 1.0})
        out.append(float(np.float64(np.float32(m.optimizer.lr)) if False else float(m.optimizer.lr)))
    return m, out


def test_reduce_lr_sequence_matches_notebook_direct_emulator():
    """notebooks/Training.ipynb cell 5 stream: lr 0.01, factor 0.95 -> the printed values
    are float64(float32(lr)) * factor, then stored back as float32."""
    printed = [0.009499999787658453, 0.009024999709799886, 0.008573750033974648, 0.008145062532275914,
               0.0077378091402351854, 0.007350918860174715, 0.006983372895047068, 0.006634204206056892,
               0.006302493973635137, 0.005987369385547936]
    m = FakeModel(0.01)
    got = []
    for _ in printed:
        old = float(np.float32(m.optimizer.lr))
        new = max(old * 0.95, 1e-4)
        got.append(new)
        m.optimizer.lr = new
    assert got == printed
    # and through the callback itself
    m2, lrs = _plateau_lrs(0.01, 0.95, 1e-4, len(printed))
    np.testing.assert_array_equal(np.float32(lrs), np.float32(printed))


def test_reduce_lr_sequence_matches_notebook_autoencoder_and_floor():
    printed = [0.0009000000427477062, 0.0008100000384729356, 0.0007290000503417104, 0.0006561000715009868,
               0.0005904900433961303, 0.0005314410547725857, 0.00047829695977270604, 0.0004304672533180565,
               0.00038742052274756136, 0.0003486784757114947]
    m = FakeModel(0.001)
    got = []
    for _ in printed:
        new = max(float(np.float32(m.optimizer.lr)) * 0.9, 1e-4)
        got.append(new)
        m.optimizer.lr = new
    assert got == printed
    m2, lrs = _plateau_lrs(0.001, 0.9, 1e-4, 40)
    assert lrs[-1] == float(np.float32(1e-4))  # floored at min_lr, then no further change
    assert lrs[-1] == lrs[-2]


def test_reduce_lr_patience_and_min_delta():
    m = FakeModel(0.01)
    cb = cbm.ReduceLROnPlateau(monitor="val_loss", patience=5, factor=0.95, min_delta=5e-9, min_lr=1e-4)
    cb.set_model(m); cb.on_train_begin()
    logs = {"val_loss": 1.0}
    cb.on_epoch_end(0, logs)
    assert logs["lr"] == float(np.float32(0.01))
    for e in range(1, 5):
        cb.on_epoch_end(e, {"val_loss": 1.0 - 1e-9})  # inside min_delta: not an improvement
    assert float(m.optimizer.lr) == float(np.float32(0.01))
    cb.on_epoch_end(5, {"val_loss": 1.0})
    assert float(m.optimizer.lr) == float(np.float32(0.009499999787658453))
    cb.on_epoch_end(6, {"val_loss": 0.5})  # improvement resets the wait
    assert cb.wait == 0 and cb.best == 0.5


def test_early_stopping_restores_best_weights():
    """Training.ipynb: patience 15, stopped at epoch 265 with best epoch 250."""
    m = FakeModel(0.01)
    cb = cbm.EarlyStopping(monitor="val_loss", patience=15, min_delta=1e-10, restore_best_weights=True)
    cb.set_model(m); cb.on_train_begin()
    stopped = None
    for epoch in range(400):
        m.w = [np.full(3, float(epoch))]
        val = 1.0 / (1 + epoch) if epoch <= 249 else 1.0
        cb.on_epoch_end(epoch, {"val_loss": val})
        if m.stop_training:
            stopped = epoch
            break
    assert stopped == 264 and cb.best_epoch == 249  # 1-based: epochs 265 and 250
    np.testing.assert_array_equal(m.w[0], np.full(3, 249.0))


def test_history_and_callback_list():
    h = cbm.History()
    cl = cbm.CallbackList([h], model=None, params={"epochs": 2})
    cl.on_train_begin()
    cl.on_epoch_end(0, {"loss": 1.0, "val_loss": 2.0})
    cl.on_epoch_end(1, {"loss": 0.5, "val_loss": 1.5})
    assert h.history == {"loss": [1.0, 0.5], "val_loss": [2.0, 1.5]} and h.epoch == [0, 1]


def test_adam_config_and_float32_lr():
    o = optm.Adam(0.01)
    assert o.lr.dtype == np.float32 and float(o.lr) == 0.009999999776482582
    o.learning_rate = 0.5
    assert float(o.lr) == 0.5
    assert o.get_config()["epsilon"] == 1e-7 and o.get_config()["beta_2"] == 0.999
    with pytest.raises(ValueError):
        optm.get("sgd")


def test_losses_row_weights_and_identity():
    losses, synth, pp = pkg("losses"), pkg("synth"), pkg("preprocess")
    sig = synth.make_signals(64, seed=5)
    y_true = pp.preproc(sig[:10], sig); y_pred = pp.preproc(sig[-10:], sig)
    fn = losses.relative_mse_loss(sig)
    mse = losses.mean_squared_error(y_true, y_pred)
    amp = np.max(np.abs(sig[:10] / np.std(sig)), axis=1)
    np.testing.assert_allclose(fn(y_true, y_pred), mse / amp ** 2, rtol=2e-5)  # tests/test_emulator.py:24-33
    w = losses.row_weight_fn(fn)(y_true)
    np.testing.assert_allclose(w * np.sum((y_true - y_pred).astype(np.float64) ** 2, axis=1), fn(y_true, y_pred), rtol=2e-5)
    np.testing.assert_allclose(losses.row_weight_fn("mse")(y_true), 1 / 451)
    with pytest.raises(ValueError):
        losses.row_weight_fn(lambda a, b: a - b)


def test_shipped_models_learning_rates_are_members_of_the_float32_plateau_sequence():
    """A free pin of the [K] ReduceLROnPlateau state machine on the reference's own artefacts: the learning rates
    stored in the `training_config` of the shipped models/autoencoder_based_emulator/{ae_emulator,autoencoder}.h5
    (tests/golden/ae_path_weights.npz, written by make_weights_fixture.py) are, BIT FOR BIT as float32, what the
    callback produces from Adam(0.01) after 34 and 20 reductions by 0.9 -- i.e. new = float64(float32(lr)) * factor,
    stored back as float32 -- and are no member of the sequences a float64 learning rate would produce."""
    import os
    from conftest import GOLDEN
    d = np.load(os.path.join(GOLDEN, "ae_path_weights.npz"))
    for stem, k in (("ae_emulator", 34), ("autoencoder", 20)):
        shipped = np.float32(d[stem + "/adam_learning_rate"])
        _, lrs = _plateau_lrs(0.01, 0.9, 0.0, k)
        assert np.float32(lrs[-1]) == shipped, (stem, lrs[-1], float(shipped))
        assert shipped not in np.float32(lrs[:-1])
