"""Loss functions of the hot path.

Both losses the reference trains with reduce to a per-sample weight on the squared
error, ``loss_i = w_i * sum_j (y_ij - p_ij)^2``:
  * ``relative_mse_loss(signal_train)`` (emulator.py:51-83): w_i = 1 / (D * amp_i^2),
    amp_i = max_j |y_ij + mean_j/std| -- it depends on the target row only, so it is
    computed once per ``fit`` on the host and fused into dL/dpred on the device;
  * ``mean_squared_error`` (notebooks/Training.ipynb cell 10): w_i = 1 / D.
The callables below also evaluate the loss on numpy arrays, so they can be used the
way the reference's tests use the Keras ones (tests/test_emulator.py:24-33).
"""
import numpy as np

from . import preprocess as pp


def mean_squared_error(y_true, y_pred):
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    return np.mean((y_pred - y_true) ** 2, axis=-1)


mean_squared_error._v21_row_weight = lambda y_true: np.full(len(y_true), 1.0 / y_true.shape[1])
mse = mean_squared_error


def relative_mse_loss(signal_train):
    """The square of the paper's figure of merit in units of the training std: per-sample
    MSE divided by the squared amplitude of the (un-preprocessed) true signal."""
    stats = pp.SignalStats.of(signal_train)
    shift = stats.mean / stats.std

    def amplitude(y_true):
        y_true = np.asarray(y_true)
        return np.max(np.abs(y_true + shift.astype(y_true.dtype)), axis=1)

    def loss_function(y_true, y_pred):
        return mean_squared_error(y_true, y_pred) / amplitude(y_true) ** 2

    loss_function._v21_row_weight = lambda y_true: 1.0 / (y_true.shape[1] * amplitude(y_true).astype(np.float64) ** 2)
    loss_function._v21_kind = "relative_mse"
    return loss_function


def row_weight_fn(loss):
    """Resolve what ``Model.compile(loss=...)`` was given to a function y_true -> w."""
    if isinstance(loss, str):
        if loss.lower() in ("mse", "mean_squared_error"):
            return mean_squared_error._v21_row_weight
        raise ValueError("unknown loss %r" % loss)
    fn = getattr(loss, "_v21_row_weight", None)
    if fn is None:
        raise ValueError(
            "unsupported loss callable %r: the engine fuses losses of the form w(y_true) * sum((y-p)^2); use "
            "relative_mse_loss(signal_train) or mean_squared_error" % (loss,))
    return fn
