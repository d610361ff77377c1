"""Keras-``Model``-shaped front end of the MI355X engine.

The reference has no plugin boundary of its own: its seam is the subset of the
``tf.keras.Model`` API that ``emulator.py``, its notebooks and its tests touch (SURVEY
section 8b).  This module provides that subset -- ``Sequential`` (what
``emulator._gen_model`` returns, emulator.py:12-48), ``Model`` (base of the reference's
``AutoEncoder``, emulator.py:445), ``fit`` / ``predict`` / ``__call__`` / ``compile`` /
``build`` / ``summary`` / ``layers`` / ``get_weights`` / ``set_weights`` /
``stop_training`` -- with numpy in and out, and runs everything numeric through the
C ABI (``_native``) on the GPU.  There is no CPU execution path.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native, callbacks as cb_mod, losses as loss_mod, optimizers as opt_mod

_rng = np.random.default_rng()
DEFAULT_PRECISION = "f32"


def set_random_seed(seed):
    """Seed kernel initialisation and epoch shuffling (Keras: tf.random.set_seed)."""
    global _rng
    _rng = np.random.default_rng(seed)


class Dense:
    """One fully connected layer: ``activation(x @ kernel + bias)``; Glorot-uniform
    kernel, zero bias (Keras defaults, confirmed by the shipped files' model_config)."""

    def __init__(self, units, activation=None, name=None):
        act = activation if activation is not None else "linear"
        if callable(act):
            act = getattr(act, "__name__", str(act))
        if act not in ("relu", "linear"):
            raise NotImplementedError("activation %r: the engine implements 'relu' and linear (the reference "
                                      "uses ReLU hidden layers and a linear output)" % (activation,))
        self.units, self.activation, self.name = int(units), act, name
        self.kernel = None  # (in, out) float32
        self.bias = None
        self.input_dim = None
        self._version = 0  # bumped whenever kernel/bias change on the host

    @property
    def output_shape(self):
        return (None, self.units)

    def build(self, input_dim):
        if self.kernel is not None and self.input_dim == input_dim:
            return
        lim = np.sqrt(6.0 / (input_dim + self.units))
        self.kernel = _rng.uniform(-lim, lim, size=(input_dim, self.units)).astype(np.float32)
        self.bias = np.zeros(self.units, np.float32)
        self.input_dim = int(input_dim)
        self._version += 1

    @property
    def _act_code(self):
        return _native.ACT_RELU if self.activation == "relu" else _native.ACT_LINEAR

    def count_params(self):
        return 0 if self.kernel is None else self.kernel.size + self.bias.size

    def get_weights(self):
        return [self.kernel.copy(), self.bias.copy()]


class GaussianLatent(Dense):
    """The variational latent layer (build-side: the reference snapshot
    keeps only its ``z_mean`` layer name).  One Dense(in -> 2*units) whose columns are
    [z_mean | z_log_var]; the next layer sees z = z_mean + exp(z_log_var/2) * eps while
    training and z = z_mean in ``predict``.  ``units`` is the latent width."""

    def __init__(self, units, name=None):
        super().__init__(units, activation=None, name=name)
        self.activation = "gaussian"

    def build(self, input_dim):
        if self.kernel is not None and self.input_dim == input_dim:
            return
        lim = np.sqrt(6.0 / (input_dim + self.units))  # two Glorot blocks, as two Dense(units) heads would get
        self.kernel = _rng.uniform(-lim, lim, size=(input_dim, 2 * self.units)).astype(np.float32)
        self.bias = np.zeros(2 * self.units, np.float32)
        self.input_dim = int(input_dim)
        self._version += 1

    @property
    def _act_code(self):
        return _native.ACT_GAUSS


class Input:
    """Placeholder for ``tf.keras.Input(shape=(d,))`` (emulator.py:39)."""

    def __init__(self, shape):
        self.dim = int(shape[-1] if isinstance(shape, (tuple, list)) else shape)


class Model:
    """Base class: a chain of Dense layers evaluated and trained on the device.

    Subclasses (the reference's ``AutoEncoder``) expose sub-models; ``_chain()`` returns
    the list of ``Sequential`` blocks that make up the forward pass."""

    def __init__(self, name=None):
        self.name = name
        self.stop_training = False
        self.optimizer = None
        self.loss = None
        self.precision = DEFAULT_PRECISION
        self._stack = None
        self._stack_sig = None
        self._trainer = None
        self._trainer_sig = None
        self._dirty_host = False  # device weights newer than the layers' numpy copies
        # variational mode (stacks holding a GaussianLatent layer): loss_i = recon_i + kl_weight * KL_i
        self.kl_weight = 0.0
        self.sample_latent = True

    # -- structure ---------------------------------------------------------------
    def _chain(self):
        raise NotImplementedError

    def _dense_layers(self):
        return [l for m in self._chain() for l in m._layers]

    @property
    def layers(self):
        return self._dense_layers()

    def _input_dim(self):
        return self._chain()[0]._in_dim

    def build(self, input_shape):
        d = int(input_shape[-1])
        for m in self._chain():
            m._build(d)
            d = m._layers[-1].units
        return self

    @property
    def built(self):
        return all(l.kernel is not None for l in self._dense_layers())

    def count_params(self):
        return sum(l.count_params() for l in self._dense_layers())

    def summary(self, print_fn=print):
        name = self.name or self.__class__.__name__.lower()
        w = 65
        print_fn('Model: "%s"' % name)
        print_fn("_" * w)
        print_fn(" %-27s %-25s %-10s" % ("Layer (type)", "Output Shape", "Param #"))
        print_fn("=" * w)
        for i, l in enumerate(self._dense_layers()):
            lname = l.name or ("dense" if i == 0 else "dense_%d" % i)
            print_fn(" %-27s %-25s %-10d" % ("%s (Dense)" % lname, "(None, %d)" % l.units, l.count_params()))
            print_fn(" " * w)
        print_fn("=" * w)
        n = self.count_params()
        print_fn("Total params: {:,}".format(n))
        print_fn("Trainable params: {:,}".format(n))
        print_fn("Non-trainable params: 0")
        print_fn("_" * w)

    # -- device stack --------------------------------------------------------------
    def _signature(self):
        ls = self._dense_layers()
        return tuple((id(l), l._version, l.input_dim, l.units, l.activation) for l in ls)

    def _ensure_stack(self):
        if not self.built:
            raise ValueError("model is not built: call build((None, in_dim)) or pass in_dim to the constructor")
        sig = self._signature()
        if self._stack is None or self._stack_sig != sig:
            for m in self._chain():
                if m is not self:
                    m._sync_host()
            self._sync_host()
            sig = self._signature()
            ls = self._dense_layers()
            dims = [ls[0].input_dim] + [l.units for l in ls]
            act = [l._act_code for l in ls]
            ctx = _native.Context.default()
            self._stack = _native.Stack(ctx, dims, act)
            self._stack.set_weights(self._flat_host())
            self._stack_sig = sig
            self._trainer = None
        return self._stack

    def _flat_host(self):
        return np.concatenate([a.ravel() for l in self._dense_layers() for a in (l.kernel, l.bias)]).astype(np.float32)

    def _sync_host(self):
        """Pull trained weights back into the layers' numpy arrays."""
        if self._dirty_host and self._stack is not None:
            flat, o = self._stack.get_weights(), 0
            for l in self._dense_layers():
                k = l.kernel.size
                l.kernel = flat[o:o + k].reshape(l.kernel.shape).copy(); o += k
                nb = l.bias.size
                l.bias = flat[o:o + nb].copy(); o += nb
                l._version += 1
            self._dirty_host = False
            self._stack_sig = self._signature()  # the device copy IS these weights

    def get_weights(self):
        self._sync_host()
        return [a for l in self._dense_layers() for a in l.get_weights()]

    def set_weights(self, weights):
        ls = self._dense_layers()
        if len(weights) != 2 * len(ls):
            raise ValueError("expected %d arrays (kernel, bias per layer), got %d" % (2 * len(ls), len(weights)))
        for i, l in enumerate(ls):
            k, b = np.asarray(weights[2 * i], np.float32), np.asarray(weights[2 * i + 1], np.float32)
            if l.kernel is not None and (k.shape != l.kernel.shape or b.shape != l.bias.shape):
                raise ValueError("layer %d: weight shapes %r/%r do not match %r/%r" % (i, k.shape, b.shape, l.kernel.shape, l.bias.shape))
            l.kernel, l.bias, l.input_dim = k.copy(), b.copy(), k.shape[0]
        self._dirty_host = False
        old = self._stack_sig
        for l in ls:
            l._version += 1
        new = self._signature()
        same_shape = old is not None and [t[2:] for t in old] == [t[2:] for t in new] and \
            [t[0] for t in old] == [t[0] for t in new]
        if self._stack is not None and same_shape:
            self._stack.set_weights(self._flat_host())  # keep the stack (and the optimizer state)
            self._stack_sig = new
        else:
            self._stack = None

    # -- inference -----------------------------------------------------------------
    def predict(self, x, batch_size=None, verbose=0, precision=None, devices=None, flags=0, **_):
        """numpy (n, in) -> numpy float32 (n, out).  float64 input is cast to float32 on
        the way in, as Keras does [K]; rows are independent, so ``batch_size`` is
        accepted and ignored.  ``devices`` (not in the reference): a list of GPU ordinals
        -- the rows are cut into contiguous blocks, one per entry, evaluated concurrently on
        replicas of the stack (weights copied once per change) and put back in order; no
        collective is involved."""
        x = np.asarray(x)
        if x.ndim == 1:
            x = x[None, :]
        if not self.built:
            self.build((None, x.shape[-1]))
        st = self._ensure_stack()
        if not devices or len(devices) == 1 and int(devices[0]) == st.ctx.device:
            return st.forward(x, precision or self.precision, flags=flags)
        return self._predict_on_devices(st, x, precision or self.precision, [int(d) for d in devices], flags)

    def _predict_on_devices(self, st, x, precision, devices, flags):
        from concurrent.futures import ThreadPoolExecutor
        reps = self.__dict__.setdefault("_replicas", {})
        # trained weights newer than the host copies: pulled back ONCE (that bumps the layer versions, hence the
        # signature below), so the replicas are refreshed once per change and not on every call after a fit()
        self._sync_host()
        sig = (id(st), self._stack_sig, id(getattr(st, "_out_stats", None)), id(getattr(st, "_in_stats", None)))
        flat = None
        stacks = []
        for slot, d in enumerate(devices):  # one replica per LIST ENTRY (an ordinal may appear twice)
            key = (slot, d)
            ent = reps.get(key)
            if ent is None or ent[0] != sig:
                if flat is None:
                    flat = st.get_weights()
                same = ent is not None and ent[1].dims == st.dims and ent[1].act == st.act
                rs = ent[1] if same else _native.Stack(_native.Context(d), st.dims, st.act)
                rs.set_weights(flat)
                if getattr(st, "_out_stats", None) is not None:
                    rs.set_output_transform(st._out_stats.std, st._out_stats.mean)
                if getattr(st, "_in_stats", None) is not None:
                    ps = st._in_stats
                    rs.set_input_transform(ps.log_mask, ps.zero_floor, ps.lo, ps.hi)
                reps[key] = ent = (sig, rs)
            stacks.append(ent[1])
        n = x.shape[0]
        cuts = [n * i // len(devices) for i in range(len(devices) + 1)]
        with ThreadPoolExecutor(max_workers=len(devices)) as pool:  # ctypes releases the GIL during the calls
            parts = list(pool.map(lambda a: a[0].forward(x[a[1]:a[2]], precision, flags=flags),
                                  [(s_, cuts[i], cuts[i + 1]) for i, s_ in enumerate(stacks)]))
        return np.concatenate(parts, axis=0)

    def __call__(self, x, training=False):
        return self.predict(np.asarray(x))

    # -- training ------------------------------------------------------------------
    def compile(self, optimizer="adam", loss=None, **_):
        prev = self.optimizer
        self.optimizer = opt_mod.get(optimizer)
        if self.optimizer is not prev:
            self._restore_state = None  # a new optimizer starts from zero moments (Keras semantics)
        self.loss = loss
        self._row_weight = loss_mod.row_weight_fn(loss)
        self._trainer = None

    def _ensure_trainer(self, batch):
        stack = self._ensure_stack()
        sig = (id(stack), self.precision, id(self.optimizer))
        if self._trainer is None or self._trainer_sig != sig or self._trainer.max_batch < batch:
            old = self._trainer if self._trainer_sig == sig else None
            # a trainer that is only being replaced by a roomier one (evaluate / fit with a larger batch)
            # hands its optimizer state over: Keras keeps iterations, m and v across fit()/evaluate() calls [K]
            carried = old.get_state() if old is not None else None
            self._trainer = _native.Trainer(stack, self.precision, max(batch, 1))
            self._trainer_sig = sig
            mv = getattr(self, "_restore_state", None)  # Adam moments of a loaded file (h5lite.load_model)
            if carried is not None:
                self._trainer.set_state(*carried)
            elif mv is not None and mv[0] is not None and mv[0].size == stack.num_params:
                self._trainer.set_state(self.optimizer.iterations, mv[0], mv[1])
            else:
                self._trainer.set_state(self.optimizer.iterations)
            self._restore_state = None
        o = self.optimizer
        self._trainer.set_adam(float(o.lr), o.beta_1, o.beta_2, o.epsilon)
        if any(isinstance(l, GaussianLatent) for l in self._dense_layers()):
            if getattr(self, "_vae_seed", None) is None:
                self._vae_seed = int(_rng.integers(0, 2**63))
            self._trainer.set_vae(self.kl_weight, self.sample_latent, self._vae_seed)
        return self._trainer

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=0, callbacks=None,
            validation_data=None, shuffle=True, validation_batch_size=None, initial_epoch=0, **_):
        """Keras fit(): per epoch a fresh shuffle, batches of ``batch_size`` with the partial
        last batch kept, epoch loss = sample-weighted mean of batch losses, validation
        pass, ``callbacks.on_epoch_end(epoch, logs)``, ``stop_training`` honoured."""
        if self.optimizer is None or self.loss is None:
            raise RuntimeError("You must compile your model before training: model.compile(optimizer=, loss=)")
        batch_size = 32 if batch_size is None else int(batch_size)
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        if not self.built:
            self.build((None, x.shape[-1]))
        n = x.shape[0]
        tr = self._ensure_trainer(batch_size)
        same = y.shape == x.shape and (y is x or np.array_equal(x, y))
        tr.set_data(0, x, None if same else y, self._row_weight(y))
        vb = int(validation_batch_size or batch_size)
        if validation_data is not None:
            xv = np.ascontiguousarray(validation_data[0], dtype=np.float32)
            yv = np.ascontiguousarray(validation_data[1], dtype=np.float32)
            same_v = yv.shape == xv.shape and np.array_equal(xv, yv)
            tr.set_data(1, xv, None if same_v else yv, self._row_weight(yv))
        dp = tr.ctx.nranks > 1
        if dp:
            # Data parallel (parallel.init_engine_comm on this context): every rank holds the whole training set and
            # trains on its share of each global batch, so the replicas must START equal and SHUFFLE alike --
            # rank 0's weights, optimizer state and, per epoch, permutation are broadcast over the process group.
            from . import parallel

            def bcast(a):  # (the context's GPU, not torch's per-thread current device)
                return parallel.broadcast_array(a, device=tr.ctx.device)

            self._stack.set_weights(bcast(self._stack.get_weights()))
            it, mm, vv = tr.get_state()
            tr.set_state(int(bcast(np.array([it], np.int64))[0]), bcast(mm), bcast(vv))
            if getattr(self, "_vae_seed", None) is not None:
                self._vae_seed = int(bcast(np.array([self._vae_seed], np.uint64))[0])
        history = cb_mod.History()
        cbs = cb_mod.CallbackList([history] + list(callbacks or []), self,
                                  {"epochs": epochs, "steps": -(-n // batch_size), "verbose": verbose})
        self.stop_training = False
        self._dirty_host = True
        cbs.on_train_begin()

        def draw():
            return _rng.permutation(n).astype(np.int32)

        def draw_ahead(state):
            # a PRIVATE generator started from a snapshot of the shared one: the shared stream is not touched here
            g = np.random.Generator(type(_rng.bit_generator)())
            g.bit_generator.state = state
            return g.permutation(n).astype(np.int32), g.bit_generator.state

        # The next epoch's permutation is drawn while this epoch runs on the GPU (run_epoch blocks inside the library with
        # the GIL released; drawing 24,562 indices takes ~0.25 ms, 4 % of an f32 epoch of the reference recipe during
        # which the GPU sat idle) -- from a COPY of the shared generator.  The shared generator advances only when that
        # permutation is consumed, and it is consumed only if nothing (a callback, a nested fit(), set_random_seed)
        # has touched the shared generator in between; otherwise the look-ahead is dropped and the permutation is drawn
        # the ordinary way.  Either way the stream of random numbers is what it would be without the look-ahead.
        pool = ThreadPoolExecutor(1) if shuffle else None
        ahead = None  # (future -> (perm, generator state after the draw), the shared generator, its state at the snapshot)
        try:
            for epoch in range(initial_epoch, epochs):
                cbs.on_epoch_begin(epoch)
                tr.set_lr(float(self.optimizer.lr))
                if getattr(self, "_vae_seed", None) is not None:  # a callback may anneal kl_weight between epochs
                    tr.set_vae(self.kl_weight, self.sample_latent, self._vae_seed)
                perm = None
                if shuffle:
                    if ahead is not None and ahead[1] is _rng and _rng.bit_generator.state == ahead[2]:
                        perm, after = ahead[0].result()
                        _rng.bit_generator.state = after
                    else:
                        perm = draw()
                    ahead = None
                if dp and perm is not None:
                    perm = bcast(perm)
                if shuffle and epoch + 1 < epochs:
                    snap = _rng.bit_generator.state
                    ahead = (pool.submit(draw_ahead, snap), _rng, snap)
                logs = {"loss": tr.run_epoch(perm, batch_size)}
                self._dirty_host = True
                if validation_data is not None:
                    logs["val_loss"] = tr.evaluate(1, min(vb, tr.max_batch))
                if verbose in (1, 2):
                    print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4e" % kv for kv in logs.items()))
                cbs.on_epoch_end(epoch, logs)
                if self.stop_training:
                    break
        finally:
            if pool is not None:
                pool.shutdown(wait=True)  # (an unused look-ahead never touched the shared generator: nothing to undo)
        cbs.on_train_end()
        self.optimizer.iterations = tr.get_state()[0]
        self._sync_host()
        return history

    def evaluate(self, x, y, batch_size=None, verbose=0, **_):
        """Mean per-sample loss over (x, y).  The batch size does not change the result (rows are
        independent, the mean is sample-weighted), so an existing trainer is used at its own capacity."""
        b = int(batch_size or 32)
        if self._trainer is not None:
            b = min(b, self._trainer.max_batch)
        tr = self._ensure_trainer(b)
        x = np.ascontiguousarray(x, dtype=np.float32); y = np.ascontiguousarray(y, dtype=np.float32)
        tr.set_data(1, x, y, self._row_weight(y))
        return tr.evaluate(1, b)

    def save(self, path, include_optimizer=True):
        """``path`` ending in .h5/.hdf5: a Keras legacy-H5 model file (``h5write``: layer names, kernels, biases,
        activations, and the Adam state when the model was trained) that ``tf.keras.models.load_model``,
        h5py and ``h5lite.load_model`` read.  Anything else: a .npz of the same arrays."""
        self._sync_host()
        ls = self._dense_layers()
        if str(path).endswith((".h5", ".hdf5")):
            from . import h5write
            names, seen = [], set()
            for i, l in enumerate(ls):
                n = l.name or ("dense" if i == 0 else "dense_%d" % i)
                while n in seen:
                    n += "_"
                seen.add(n)
                names.append(n)
            opt = None
            if include_optimizer and self._trainer is not None and self.optimizer is not None:
                it, m, v = self._trainer.get_state()
                o = self.optimizer
                opt = {"iter": it, "m": m, "v": v,
                       "config": {"learning_rate": float(o.lr), "beta_1": o.beta_1, "beta_2": o.beta_2, "epsilon": o.epsilon}}
            h5write.write_keras_h5(path, names, [l.kernel for l in ls], [l.bias for l in ls],
                                   ["linear" if l.activation == "gaussian" else l.activation for l in ls],
                                   model_name=self.name, optimizer=opt,
                                   extra_attrs={"v21_layer_kinds": ",".join(l.activation for l in ls)})
            return
        blob = {"n_layers": np.array(len(ls))}
        for i, l in enumerate(ls):
            blob["W%d" % i], blob["b%d" % i] = l.kernel, l.bias
            blob["act%d" % i] = np.array(l.activation)
        np.savez(path, **blob)


class Sequential(Model):
    """What ``_gen_model`` builds: [Input?] + Dense... ; ``layers`` excludes the Input."""

    def __init__(self, layers=None, name=None):
        super().__init__(name=name)
        self._in_dim = None
        self._layers = []
        for l in layers or []:
            if isinstance(l, Input):
                self._in_dim = l.dim
            else:
                self._layers.append(l)
        if self._in_dim is not None:
            self._build(self._in_dim)

    def _chain(self):
        return [self]

    def _build(self, d):
        self._in_dim = int(d)
        for l in self._layers:
            l.build(d)
            d = l.units


def sequential_from_arrays(Ws, bs, acts=None, name=None):
    """A Sequential holding given kernels/biases (used by the loaders)."""
    acts = acts or (["relu"] * (len(Ws) - 1) + ["linear"])
    m = Sequential([Input((Ws[0].shape[0],))] + [Dense(W.shape[1], a) for W, a in zip(Ws, acts)], name=name)
    m.set_weights([a for W, b in zip(Ws, bs) for a in (W, b)])
    return m


class ChainedModel(Model):
    """Several Sequential blocks evaluated as ONE device stack (emulator -> decoder,
    emulator.py:789-790; encoder -> decoder, emulator.py:517).  The blocks keep their own
    layer objects, so weights trained through the chain are visible in the blocks."""

    def __init__(self, blocks, name=None):
        super().__init__(name=name)
        self._blocks = list(blocks)

    def _chain(self):
        return self._blocks

