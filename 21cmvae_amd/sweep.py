"""Hyper-parameter sweeps: several models of the same depth trained in lock step on one
shared batch stream (BASELINE configs[4]: "64 concurrent latent-dim/hidden-width configs
packed as batched GEMM", 8 per GPU).

The reference has no sweep code: it trains one model per ``fit`` call
(emulator.py:369-378, :739-747) and its hyper-parameter tuner is not shipped.  ``fit_models``
runs exactly that ``fit`` for every model -- same shuffle, same batches, same Keras
bookkeeping, per-model optimizer/callbacks/History -- but issues each phase of the step
(layer k forward, loss, layer k backward, Adam) as ONE grouped device launch for all models
(include/v21.h: v21_sweep_*).  Models that stop early (``model.stop_training``) drop out of
the group; the others continue.
"""
import numpy as np

from . import _native, callbacks as cb_mod, engine

MAX_GROUP = 64  # include/v21.h: v21_sweep_create (r5: 64 = the whole of BASELINE configs[4] in one group; 16 until r4)


def fit_models(models, x, y, batch_size=256, epochs=1, validation_data=None, callbacks=None, shuffle=True,
               verbose=0, validation_batch_size=None):
    """Train ``models`` (compiled ``engine.Model`` objects of equal depth, activations and
    in/out width) on the same data.  ``callbacks``: None or one list per model.  Returns one
    ``History`` per model."""
    models = list(models)
    if not models:
        return []
    if len(models) > MAX_GROUP:
        raise ValueError("at most %d models per sweep group (run several groups, e.g. one per GPU)" % MAX_GROUP)
    callbacks = callbacks or [[] for _ in models]
    if len(callbacks) != len(models):
        raise ValueError("callbacks: one list per model")
    batch_size = int(batch_size)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    n = x.shape[0]
    same = y.shape == x.shape and (y is x or np.array_equal(x, y))
    for m in models:
        if m.optimizer is None or m.loss is None:
            raise RuntimeError("You must compile your model before training: model.compile(optimizer=, loss=)")
        if not m.built:
            m.build((None, x.shape[-1]))
    trainers = [m._ensure_trainer(batch_size) for m in models]
    rw = models[0]._row_weight(y)
    for m in models[1:]:
        if not np.array_equal(m._row_weight(y), rw):
            raise ValueError("all models of a sweep must be compiled with the same loss")
    vb = int(validation_batch_size or batch_size)
    if validation_data is not None:
        xv = np.ascontiguousarray(validation_data[0], dtype=np.float32)
        yv = np.ascontiguousarray(validation_data[1], dtype=np.float32)
        same_v = yv.shape == xv.shape and np.array_equal(xv, yv)
        rwv = models[0]._row_weight(yv)
        for tr in trainers:
            tr.set_data(1, xv, None if same_v else yv, rwv)
    hists = [cb_mod.History() for _ in models]
    cbs = [cb_mod.CallbackList([h] + list(c or []), m, {"epochs": epochs, "steps": -(-n // batch_size), "verbose": verbose})
           for h, c, m in zip(hists, callbacks, models)]
    active = list(range(len(models)))
    holder, group = None, None
    for m in models:
        m.stop_training = False
        m._dirty_host = True
    for c in cbs:
        c.on_train_begin()
    for epoch in range(epochs):
        if not active:
            break
        if holder != active[0]:  # the first active trainer holds the training set of the group
            holder = active[0]
            trainers[holder].set_data(0, x, None if same else y, rw)
            group = None
        if group is None:
            group = _native.Sweep([trainers[i] for i in active])
        for i in active:
            cbs[i].on_epoch_begin(epoch)
            trainers[i].set_lr(float(models[i].optimizer.lr))
            if getattr(models[i], "_vae_seed", None) is not None:  # a callback may anneal kl_weight (as Model.fit)
                trainers[i].set_vae(models[i].kl_weight, models[i].sample_latent, models[i]._vae_seed)
        perm = engine._rng.permutation(n).astype(np.int32) if shuffle else None
        losses = group.run_epoch(perm, batch_size)
        still = []
        for i, loss in zip(active, losses):
            logs = {"loss": loss}
            models[i]._dirty_host = True
            if validation_data is not None:
                logs["val_loss"] = trainers[i].evaluate(1, min(vb, trainers[i].max_batch))
            if verbose in (1, 2):
                print("model %d - epoch %d/%d - " % (i, epoch + 1, epochs) + " - ".join("%s: %.4e" % kv for kv in logs.items()))
            cbs[i].on_epoch_end(epoch, logs)
            if not models[i].stop_training:
                still.append(i)
        if still != active:
            active, group = still, None
    for i, m in enumerate(models):
        cbs[i].on_train_end()
        m.optimizer.iterations = trainers[i].get_state()[0]
        m._sync_host()
    return hists
