"""Training callbacks with the Keras 2.7 protocol the reference's notebooks rely on
(notebooks/Training.ipynb cells 5 and 11: ``EarlyStopping(monitor, patience, min_delta,
restore_best_weights)``, ``ReduceLROnPlateau(monitor, patience, factor, min_delta, min_lr)``;
emulator.py:366-368 appends a ``TqdmCallback``).

The state machines are restated from the published Keras behaviour [K]:
  * "improved" for a loss-like monitor means ``current < best - abs(min_delta)``;
  * ReduceLROnPlateau multiplies the optimizer's float32 learning rate by ``factor`` in
    float64 and stores it back as float32, floored at ``min_lr`` -- this reproduces the
    learning-rate sequences printed in the reference's notebooks digit for digit
    (tests/test_callbacks.py);
  * EarlyStopping(restore_best_weights=True) keeps the best weights in host memory and
    puts them back when it stops the run.
"""
import numpy as np


class Callback:
    def __init__(self):
        self.model = None
        self.params = {}

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


class History(Callback):
    """``model.fit`` returns this; ``.history`` maps metric name -> list of floats."""

    def on_train_begin(self, logs=None):
        self.epoch = []
        self.history = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class CallbackList:
    def __init__(self, callbacks, model, params):
        self.callbacks = list(callbacks)
        for cb in self.callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(model)
            if hasattr(cb, "set_params"):
                cb.set_params(params)

    def _call(self, name, *a):
        for cb in self.callbacks:
            fn = getattr(cb, name, None)
            if fn is not None:
                fn(*a)

    def on_train_begin(self, logs=None):
        self._call("on_train_begin", logs)

    def on_train_end(self, logs=None):
        self._call("on_train_end", logs)

    def on_epoch_begin(self, epoch, logs=None):
        self._call("on_epoch_begin", epoch, logs)

    def on_epoch_end(self, epoch, logs=None):
        self._call("on_epoch_end", epoch, logs)


def _monitor_op(mode, monitor):
    if mode not in ("auto", "min", "max"):
        mode = "auto"
    if mode == "max" or (mode == "auto" and ("acc" in monitor or monitor.startswith("fmeasure"))):
        return np.greater, 1.0
    return np.less, -1.0


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto",
                 baseline=None, restore_best_weights=False):
        super().__init__()
        self.monitor, self.patience, self.verbose = monitor, patience, verbose
        self.baseline, self.restore_best_weights = baseline, restore_best_weights
        self.monitor_op, sign = _monitor_op(mode, monitor)
        self.min_delta = abs(min_delta) * sign
        self.wait = 0
        self.stopped_epoch = 0
        self.best_weights = None
        self.best_epoch = 0

    def on_train_begin(self, logs=None):
        self.wait = 0
        self.stopped_epoch = 0
        self.best = np.inf if self.monitor_op == np.less else -np.inf
        self.best_weights = None
        self.best_epoch = 0

    def on_epoch_end(self, epoch, logs=None):
        current = (logs or {}).get(self.monitor)
        if current is None:
            return
        if self.restore_best_weights and self.best_weights is None:
            self.best_weights = self.model.get_weights()
        self.wait += 1
        if self.monitor_op(current - self.min_delta, self.best):
            self.best = current
            self.best_epoch = epoch
            if self.restore_best_weights:
                self.best_weights = self.model.get_weights()
            if self.baseline is None or self.monitor_op(current, self.baseline):
                self.wait = 0
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.model.stop_training = True
            if self.restore_best_weights and self.best_weights is not None:
                if self.verbose > 0:
                    print("Restoring model weights from the end of the best epoch: %d." % (self.best_epoch + 1))
                self.model.set_weights(self.best_weights)

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose > 0:
            print("Epoch %d: early stopping" % (self.stopped_epoch + 1))


class ReduceLROnPlateau(Callback):
    def __init__(self, monitor="val_loss", factor=0.1, patience=10, verbose=0, mode="auto",
                 min_delta=1e-4, cooldown=0, min_lr=0, **kwargs):
        super().__init__()
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor >= 1.0.")
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr, self.mode = min_delta, cooldown, min_lr, mode
        self._reset()

    def _reset(self):
        op, _ = _monitor_op(self.mode, self.monitor)
        if op == np.less:
            self.monitor_op = lambda a, b: np.less(a, b - self.min_delta)
            self.best = np.inf
        else:
            self.monitor_op = lambda a, b: np.greater(a, b + self.min_delta)
            self.best = -np.inf
        self.cooldown_counter = 0
        self.wait = 0

    def on_train_begin(self, logs=None):
        self._reset()

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs["lr"] = float(np.float32(self.model.optimizer.lr))
        current = logs.get(self.monitor)
        if current is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self.monitor_op(current, self.best):
            self.best = current
            self.wait = 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old_lr = float(np.float32(self.model.optimizer.lr))  # float32 variable read back
                if old_lr > np.float32(self.min_lr):
                    new_lr = max(old_lr * self.factor, self.min_lr)
                    self.model.optimizer.lr = new_lr                # stored as float32
                    if self.verbose > 0:
                        print("\nEpoch %d: ReduceLROnPlateau reducing learning rate to %s."
                              % (epoch + 1, repr(new_lr)))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class TqdmCallback(Callback):
    """Progress bar per epoch (the reference uses tqdm.keras.TqdmCallback, which needs
    Keras; this one needs only tqdm and degrades to silence without it)."""

    def __init__(self, **tqdm_kwargs):
        super().__init__()
        self.tqdm_kwargs = tqdm_kwargs
        self.bar = None

    def on_train_begin(self, logs=None):
        try:
            from tqdm.auto import tqdm
            self.bar = tqdm(total=self.params.get("epochs"), unit="epoch", **self.tqdm_kwargs)
        except Exception:  # pragma: no cover
            self.bar = None

    def on_epoch_end(self, epoch, logs=None):
        if self.bar is not None:
            self.bar.update(1)
            self.bar.set_postfix({k: "%.4g" % v for k, v in (logs or {}).items() if isinstance(v, float)})

    def on_train_end(self, logs=None):
        if self.bar is not None:
            self.bar.close()
            self.bar = None
