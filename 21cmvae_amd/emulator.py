"""User-facing emulator classes: the reference's ``VeryAccurateEmulator.emulator`` surface
(``DirectEmulator`` emulator.py:207-442, ``AutoEncoder`` :445-518, ``AutoEncoderEmulator``
:528-842, ``_gen_model`` :12-48, ``relative_mse_loss`` :51-83, ``error`` :129-192 and the
redshift/frequency helpers) on top of the MI355X engine instead of TensorFlow.

Same names, arguments, return values and error behaviour; the differences a user can
observe are listed in INTEGRATION.md (no dataset download, ``precision=`` option,
training-set statistics cached, AE predict evaluated as one fused device stack).
"""
import os

import numpy as np

from . import preprocess as pp
from .callbacks import TqdmCallback
from .engine import ChainedModel, Dense, GaussianLatent, Input, Model, Sequential, sequential_from_arrays
from .losses import mean_squared_error, relative_mse_loss  # noqa: F401  (public, as in the reference)

PATH = os.path.dirname(os.path.abspath(__file__)) + "/"
_native = None


def _gen_model(in_dim, hidden_dims, out_dim, activation_func, name=None, variational=False):
    """Generate a new model: Dense(h, activation) per hidden dim, then a linear
    Dense(out_dim).  ``in_dim`` None defers weight creation until ``build``/first call
    (a decoder that follows another model).  ``variational`` (not in the reference
    snapshot) makes the last layer a ``GaussianLatent`` (z_mean | z_log_var) head."""
    layers = []
    if in_dim is not None:
        layers.append(Input(shape=(in_dim,)))
    for dim in hidden_dims or []:
        layers.append(Dense(dim, activation=activation_func))
    layers.append(GaussianLatent(out_dim, name="z_mean_log_var") if variational else Dense(out_dim))
    return Sequential(layers, name=name)


NU_0 = 1420405751.7667  # Hz, rest frequency of the 21-cm line

# Three places where this module deliberately does NOT reproduce an accident of the reference
# (INTEGRATION.md section 6).  ``set_strict_reference(True)`` switches all three back to the
# reference's observable behaviour, for callers that depend on it:
#   * ``error(..., flow=x)`` or ``error(..., fhigh=x)`` with ONE bound returns shape (N, 1), because the
#     reference indexes with an (k, 1) argwhere result (emulator.py:179-184); both bounds give (N,)
#   * ``freq2redshift`` multiplies an ndarray argument by 1e6 IN PLACE (emulator.py:124), so a
#     ``frequencies=`` array passed to a constructor is left in Hz (emulator.py:314-317)
#   * the latent emulator's model name keeps the reference's spelling "ae_emualtor" (emulator.py:662)
strict_reference = False


def set_strict_reference(on=True):
    """Reproduce the reference's three accidental behaviours listed above (default: off)."""
    global strict_reference
    strict_reference = bool(on)


def redshift2freq(z):
    """Redshift -> frequency in MHz."""
    nu = NU_0 / (1 + z)
    nu /= 1e6
    return nu


def freq2redshift(nu):
    """Frequency in MHz -> redshift.  The argument is left untouched unless
    ``strict_reference`` is set (the reference converts an ndarray to Hz in place,
    emulator.py:124)."""
    if strict_reference:
        nu *= 1e6  # to Hz -- in place for ndarrays, exactly as the reference does
        return NU_0 / nu - 1
    return NU_0 / (np.asarray(nu, dtype=float) * 1e6) - 1 if np.ndim(nu) else NU_0 / (nu * 1e6) - 1


def error(true_signal, pred_signal, relative=True, nu_arr=None, flow=None, fhigh=None):
    """Eq. 1 of Bye et al. (2022): rms over bins, optionally as % of the signal amplitude
    and restricted to a frequency band (emulator.py:129-192)."""
    if (flow or fhigh) and nu_arr is None:
        raise ValueError("No frequency array is given, cannot compute error in specified frequency band.")
    pred_signal, true_signal = np.asarray(pred_signal), np.asarray(true_signal)
    if pred_signal.ndim == 1:
        pred_signal = pred_signal[None, :]
        true_signal = true_signal[None, :]
    if flow or fhigh:
        sel = np.ones(len(nu_arr), bool)
        if flow:
            sel &= nu_arr >= flow
        if fhigh:
            sel &= nu_arr <= fhigh
        f = np.flatnonzero(sel)
        if strict_reference and not (flow and fhigh):
            f = f[:, None]  # the reference's (k, 1) index -> (N, k, 1) selection -> (N, 1) result
        pred_signal, true_signal = pred_signal[:, f], true_signal[:, f]
    err = np.sqrt(np.mean((pred_signal - true_signal) ** 2, axis=1))
    if relative:
        err = err / np.max(np.abs(true_signal), axis=1) * 100
    return err


# ---- default data (emulator.py:196-204) ---------------------------------------------
hidden_dims = [288, 352, 288, 224]
redshifts = np.linspace(5, 50, 451)
_DATASET_KEYS = ("par_train", "par_val", "par_test", "signal_train", "signal_val", "signal_test")
_dataset = None


def load_dataset(path=None):
    """The six arrays of ``dataset_21cmVAE.h5``.  The reference reads that file at import and
    downloads it when missing; this package NEVER downloads: the file is looked for at
    ``path``, ``$V21_DATASET`` or next to this module, and otherwise the constructors
    require the arrays to be passed explicitly.

    The arrays are shared by every emulator built without explicit data and are READ-ONLY
    (the reference's module-level arrays are writable; an in-place edit here raises numpy's
    "assignment destination is read-only" -- pass an edited copy to the constructor instead).
    That is what lets a default-constructed emulator skip the per-call checksum of the
    training set: nobody holds a writable handle on these buffers, so the cached statistics
    (``preprocess._cached``) are exact on identity alone."""
    global _dataset
    if _dataset is not None and path is None:
        return _dataset
    cands = [path, os.environ.get("V21_DATASET"), PATH + "dataset_21cmVAE.h5"]
    for c in cands:
        if c and os.path.exists(c):
            from . import h5lite
            with h5lite.File(c) as hf:
                _dataset = {k: pp.freeze(hf[k][:]) for k in _DATASET_KEYS}
            return _dataset
    return None


def _resolve_data(given):
    if all(v is not None for v in given.values()):
        return given
    ds = load_dataset()
    if ds is None:
        missing = [k for k, v in given.items() if v is None]
        raise ValueError("dataset_21cmVAE.h5 was not found (this package does not download it): pass %s, or "
                         "set V21_DATASET" % ", ".join(missing))
    return {k: (v if v is not None else ds[k]) for k, v in given.items()}


def _grid(redshifts_, frequencies):
    if frequencies is None:
        if redshifts_ is not None:
            frequencies = redshift2freq(redshifts_)
    elif redshifts_ is None:
        redshifts_ = freq2redshift(frequencies)
    return redshifts_, frequencies


class _EmulatorBase:
    par_labels = ["fstar", "Vc", "fx", "tau", "alpha", "nu_min", "Rmfp"]

    def _set_data(self, par_train, par_val, par_test, signal_train, signal_val, signal_test, freeze_data="auto"):
        """``freeze_data`` (not in the reference) says how ``self.par_train`` / ``self.signal_train`` -- the two arrays
        whose statistics every ``predict`` needs -- are held.  The reference keeps the caller's arrays and recomputes mean /
        std / min / max of the ~44 MB training set on every call (preprocess.py:22-23, 44-45, 89-101); caching those
        numbers is exact only while the buffer cannot change.

        ``"auto"`` (default since r5): THE CALLER'S OWN ARRAYS, as in the reference -- ``em.signal_train is
        signal_train``, nothing is held twice -- with their ``writeable`` flag switched off when they own their buffer:
        nobody can then edit them in place (numpy raises "assignment destination is read-only" instead of the edit being
        silently ignored), so the statistics are cached on identity alone and a call costs no checksum (~70 us for one
        parameter vector).  To change the training set, ASSIGN a new array (``em.signal_train = new``) -- or switch the
        flag back on (``arr.setflags(write=True)``): a writable array is re-hashed on every call, see ``False``.  Arrays that
        do not own their buffer (views: the base could still be written) are treated as ``False``.
        ``True`` (r4's default): private read-only COPIES; the caller's arrays are not touched, the training set is held twice.
        ``False``: the reference's semantics to the letter -- the caller's writable arrays, edits in place honoured -- at
        the price of a whole-buffer hash per call (``preprocess._cached``: ~1.2 ms on the 44 MB set)."""
        d = _resolve_data(dict(par_train=par_train, par_val=par_val, par_test=par_test,
                               signal_train=signal_train, signal_val=signal_val, signal_test=signal_test))
        for k, v in d.items():
            if k in ("par_train", "signal_train") and not pp._is_frozen(v):
                if freeze_data is True:
                    v = pp.freeze(v)
                elif freeze_data == "auto" and isinstance(v, np.ndarray) and v.base is None and v.flags.owndata:
                    v.flags.writeable = False   # (the caller's object, locked: pp._is_frozen(v) holds from here on)
            setattr(self, k, v)
        self.par_labels = list(_EmulatorBase.par_labels)

    def _predict_stack(self, model, params, devices=None):
        """par_transform -> device stack -> unpreproc, squeezing a single row (emulator.py:401-407 / :788-795).
        Both transforms run inside the library: float32 and float64 parameter arrays are handed over RAW and take the
        reference's branch for their dtype there (include/v21.h: v21_affine_in; few rows: on the host while they are
        padded, many: in float64 on the device, so a 65,536-row call is bound by the 118 MB of results over PCIe and not
        by a 4-ms numpy transform); the un-preprocessing rides in the last layer's epilogue.  Any other dtype is
        transformed by ``preprocess.par_transform`` first."""
        from . import _native as nat
        x = np.asarray(params)
        if x.ndim == 1:
            x = x[None, :]
        st = model._ensure_stack()
        ss = pp.SignalStats.of(self.signal_train)
        if getattr(st, "_out_stats", None) is not ss:  # a new record whenever the training set changed (even in place)
            st.set_output_transform(ss.std, ss.mean)
            st._out_stats = ss
        flags = nat.FWD_OUT_TRANSFORM
        if not np.issubdtype(x.dtype, np.floating):
            x = x.astype(np.float64)  # (documented deviation: the reference truncates the fx floor in integer arrays)
        if x.dtype in (np.float32, np.float64) and x.shape[-1] <= 8:
            ps = pp.ParamStats.of(self.par_train)
            if getattr(st, "_in_stats", None) is not ps:
                st.set_input_transform(ps.log_mask, ps.zero_floor, ps.lo, ps.hi)
                st._in_stats = ps
            flags |= nat.FWD_IN_TRANSFORM
        else:
            x = pp.par_transform(x, self.par_train)
        pred = model.predict(x, devices=devices, flags=flags)
        return pred[0, :] if pred.shape[0] == 1 else pred

    def save(self):
        raise NotImplementedError("Not implemented yet.")


class DirectEmulator(_EmulatorBase):
    """The direct emulator (21cmVAE proper): 7 parameters -> 451-bin global signal."""

    def __init__(self, par_train=None, par_val=None, par_test=None, signal_train=None, signal_val=None,
                 signal_test=None, hidden_dims=hidden_dims, activation_func="relu", redshifts=redshifts,
                 frequencies=None, precision="f32", freeze_data="auto"):
        self._set_data(par_train, par_val, par_test, signal_train, signal_val, signal_test, freeze_data)
        self.emulator = _gen_model(self.par_train.shape[-1], hidden_dims, self.signal_train.shape[-1],
                                   activation_func, name="emulator")
        self.emulator.precision = precision
        self.redshifts, self.frequencies = _grid(redshifts, frequencies)

    def load_model(self, model_path=PATH + "models/emulator.h5"):
        """Load a saved model (Keras legacy .h5 or the engine's .npz).  Raises IOError if the
        path does not hold one (the reference's default file is not redistributed)."""
        from . import h5lite
        prec = self.emulator.precision
        self.emulator = h5lite.load_model(model_path)
        self.emulator.precision = prec

    def train(self, epochs, callbacks=[], verbose="tqdm", batch_size=256):
        """Train the emulator; returns (loss, val_loss) lists, one entry per epoch.
        ``batch_size`` (not in the reference, which hard-codes 256 at emulator.py:372): rows per optimizer
        step, e.g. 4,096 per GPU for BASELINE configs[3]; the default is the reference's."""
        X_train = pp.par_transform(self.par_train, self.par_train)
        X_val = pp.par_transform(self.par_val, self.par_train)
        y_train = pp.preproc(self.signal_train, self.signal_train)
        y_val = pp.preproc(self.signal_val, self.signal_train)
        callbacks = list(callbacks)  # (the reference appends to its default-argument list)
        if verbose == "tqdm":
            callbacks.append(TqdmCallback())
            verbose = 0
        hist = self.emulator.fit(x=X_train, y=y_train, batch_size=batch_size, epochs=epochs,
                                 validation_data=(X_val, y_val), validation_batch_size=batch_size,
                                 callbacks=callbacks, verbose=verbose)
        return hist.history["loss"], hist.history["val_loss"]

    def predict(self, params, devices=None):
        """Global signal(s) for one parameter vector (-> 1-D) or an (N, 7) array (-> (N, 451)).
        ``devices`` (not in the reference): GPU ordinals to spread the rows over."""
        return self._predict_stack(self.emulator, params, devices)

    def test_error(self, relative=True, flow=None, fhigh=None):
        return error(self.signal_test, self.predict(self.par_test), relative=relative,
                     nu_arr=self.frequencies, flow=flow, fhigh=fhigh)


class AutoEncoder(ChainedModel):
    """Encoder + decoder pair of the autoencoder-based emulator; ``call`` reconstructs."""

    def __init__(self, signal_train=None, enc_hidden_dims=[], dec_hidden_dims=[], latent_dim=9,
                 activation_func="relu", variational=False, kl_weight=0.0):
        """``variational`` / ``kl_weight`` (build-side): the encoder ends in a
        (z_mean | z_log_var) head, training samples z and adds kl_weight * KL to each row's
        loss; ``predict``/``encoder.predict`` use z = z_mean.  variational=False is the
        reference's deterministic autoencoder (emulator.py:445-518)."""
        if signal_train is None:
            signal_train = _resolve_data(dict(signal_train=None))["signal_train"]
        super().__init__([], name="auto_encoder")
        d = signal_train.shape[-1]
        self.encoder = _gen_model(d, enc_hidden_dims, latent_dim, activation_func, name="encoder",
                                  variational=variational)
        self.decoder = _gen_model(None, dec_hidden_dims, d, activation_func, name="decoder")
        self.kl_weight = float(kl_weight)

    def _chain(self):  # plain attribute assignment of .encoder/.decoder re-wires the chain
        return [self.encoder, self.decoder]

    def call(self, signals):
        return self.predict(signals)


# default parameters (emulator.py:522-525)
latent_dim = 9
enc_hidden_dims = [352]
dec_hidden_dims = [32, 352]
em_hidden_dims = [352, 352, 352, 224]


class AutoEncoderEmulator(_EmulatorBase):
    """The autoencoder-based emulator of Appendix A: parameters -> latent -> decoder."""

    AE_PATH = PATH + "models/autoencoder_based_emulator/"

    def __init__(self, par_train=None, par_val=None, par_test=None, signal_train=None, signal_val=None,
                 signal_test=None, latent_dim=latent_dim, enc_hidden_dims=enc_hidden_dims,
                 dec_hidden_dims=dec_hidden_dims, em_hidden_dims=em_hidden_dims, activation_func="relu",
                 redshifts=redshifts, frequencies=None, precision="f32", variational=False, kl_weight=0.0,
                 freeze_data="auto"):
        self._set_data(par_train, par_val, par_test, signal_train, signal_val, signal_test, freeze_data)
        self.redshifts, self.frequencies = _grid(redshifts, frequencies)
        autoencoder = AutoEncoder(self.signal_train, enc_hidden_dims, dec_hidden_dims, latent_dim, activation_func,
                                  variational=variational, kl_weight=kl_weight)
        autoencoder.build((None, self.signal_train.shape[-1]))
        autoencoder.precision = precision
        self.autoencoder = autoencoder
        self.emulator = _gen_model(self.par_train.shape[-1], em_hidden_dims, latent_dim, activation_func,
                                   name="ae_emualtor" if strict_reference else "ae_emulator")
        self.emulator.precision = precision
        self.precision = precision
        self._chain_model = None

    def load_model(self, emulator_path=AE_PATH + "ae_emulator.h5", encoder_path=AE_PATH + "encoder.h5",
                   decoder_path=AE_PATH + "decoder.h5"):
        from . import h5lite
        self.emulator = h5lite.load_model(emulator_path)
        encoder = h5lite.load_model(encoder_path)
        decoder = h5lite.load_model(decoder_path)
        autoencoder = AutoEncoder(signal_train=self.signal_train)
        autoencoder.encoder = encoder
        autoencoder.decoder = decoder
        for m in (self.emulator, autoencoder):
            m.precision = self.precision
        self.autoencoder = autoencoder
        self._chain_model = None

    def train(self, epochs, ae_callbacks=[], em_callbacks=[], verbose="tqdm", joint=False, batch_size=256):
        """Sequential two-phase recipe of the reference (emulator.py:701-768): fit the
        autoencoder x -> x, encode the signals with the frozen encoder, fit the emulator
        parameters -> latent.  Returns (ae_loss, ae_val_loss, loss, val_loss).

        ``joint=True`` (not in the reference; BASELINE configs[2]): both models take one optimizer step
        per batch on the SAME rows, the emulator's targets being the latents the encoder produces for those
        rows in that step -- see ``_train_joint``.
        ``batch_size`` (not in the reference, which hard-codes 256 at emulator.py:742,759): rows per optimizer step."""
        batch_size = int(batch_size)
        if joint:
            return self._train_joint(epochs, ae_callbacks, em_callbacks, verbose, batch_size)
        y_train = pp.preproc(self.signal_train, self.signal_train)
        y_val = pp.preproc(self.signal_val, self.signal_train)
        ae_callbacks, em_callbacks = list(ae_callbacks), list(em_callbacks)
        if verbose == "tqdm":
            ae_callbacks.append(TqdmCallback())
            em_callbacks.append(TqdmCallback())
            verbose = 0
        hist = self.autoencoder.fit(x=y_train, y=y_train, batch_size=batch_size, epochs=epochs,
                                    validation_data=(y_val, y_val), callbacks=ae_callbacks, verbose=verbose)
        ae_loss, ae_val_loss = hist.history["loss"], hist.history["val_loss"]
        X_train = pp.par_transform(self.par_train, self.par_train)
        X_val = pp.par_transform(self.par_val, self.par_train)
        z_train = self.autoencoder.encoder.predict(y_train)
        z_val = self.autoencoder.encoder.predict(y_val)
        hist = self.emulator.fit(x=X_train, y=z_train, batch_size=batch_size, epochs=epochs,
                                 validation_data=(X_val, z_val), callbacks=em_callbacks, verbose=verbose)
        self._chain_model = None
        return ae_loss, ae_val_loss, hist.history["loss"], hist.history["val_loss"]

    def _train_joint(self, epochs, ae_callbacks, em_callbacks, verbose, batch_size=256):
        """One pass over the data trains BOTH models (v21_joint_*): per batch, an autoencoder step (x -> x) and
        an emulator step (parameters -> the latents the encoder has just produced for these rows; no gradient
        into the encoder).  The reference's recipe waits for the autoencoder to finish first
        (emulator.py:739-764); here the emulator tracks the encoder while it still moves, and when the
        autoencoder's callbacks stop it, it is frozen (learning rate 0) and the remaining epochs ARE the
        reference's second phase.  Keras bookkeeping per model (History, callbacks, epoch losses, validation
        after every epoch -- the emulator's against the current encoder's latents of the validation set).
        Needs mean_squared_error on the emulator; any precision (f32: the reference's arithmetic, batches of up to
        2,048 rows).  A variational autoencoder (``AutoEncoder(variational=True)``) is fine: the emulator learns
        z_mean, what ``encoder.predict`` returns.
        With a data-parallel communicator on the context every rank trains on its share of every batch."""
        from . import _native as nat, callbacks as cb_mod, engine
        ae, em = self.autoencoder, self.emulator
        for m in (ae, em):
            if m.optimizer is None or m.loss is None:
                raise RuntimeError("You must compile your model before training: model.compile(optimizer=, loss=)")
        lat = em.layers[-1].units
        if not np.array_equal(em._row_weight(np.zeros((2, lat), np.float32)), em._row_weight(np.ones((2, lat), np.float32))):
            raise ValueError("joint training needs a target-independent emulator loss (mean_squared_error)")
        y_train = np.ascontiguousarray(pp.preproc(self.signal_train, self.signal_train), dtype=np.float32)
        y_val = np.ascontiguousarray(pp.preproc(self.signal_val, self.signal_train), dtype=np.float32)
        X_train = np.ascontiguousarray(pp.par_transform(self.par_train, self.par_train), dtype=np.float32)
        X_val = np.ascontiguousarray(pp.par_transform(self.par_val, self.par_train), dtype=np.float32)
        n = y_train.shape[0]
        ae_callbacks, em_callbacks = list(ae_callbacks), list(em_callbacks)
        if verbose == "tqdm":
            ae_callbacks.append(TqdmCallback())
            em_callbacks.append(TqdmCallback())
            verbose = 0
        tra, tre = ae._ensure_trainer(batch_size), em._ensure_trainer(batch_size)
        tra.set_data(0, y_train, None, ae._row_weight(y_train))
        tra.set_data(1, y_val, None, ae._row_weight(y_val))
        zdummy = np.zeros((n, lat), np.float32)
        tre.set_data(0, X_train, zdummy, em._row_weight(zdummy))
        zdv = np.zeros((X_val.shape[0], lat), np.float32)
        tre.set_data(1, X_val, zdv, em._row_weight(zdv))  # (targets: the encoder's latents of y_val, formed on the device)
        joint = nat.Joint(tra, tre, latent_layer=len(ae.encoder.layers) - 1)
        dp = tra.ctx.nranks > 1
        if dp:  # data parallel: the replicas must start equal and shuffle alike (as engine.Model.fit does)
            from . import parallel

            def bcast(a):
                return parallel.broadcast_array(a, device=tra.ctx.device)
            for m, tr in ((ae, tra), (em, tre)):
                m._stack.set_weights(bcast(m._stack.get_weights()))
                it, mm, vv = tr.get_state()
                tr.set_state(int(bcast(np.array([it], np.int64))[0]), bcast(mm), bcast(vv))
            if getattr(ae, "_vae_seed", None) is not None:
                ae._vae_seed = int(bcast(np.array([ae._vae_seed], np.uint64))[0])
                tra.set_vae(ae.kl_weight, ae.sample_latent, ae._vae_seed)
        hists = [cb_mod.History(), cb_mod.History()]
        params = {"epochs": epochs, "steps": -(-n // batch_size), "verbose": verbose}
        cbs = [cb_mod.CallbackList([hists[0]] + ae_callbacks, ae, params), cb_mod.CallbackList([hists[1]] + em_callbacks, em, params)]
        for m in (ae, em):
            m.stop_training = False
            m._dirty_host = True
        for c in cbs:
            c.on_train_begin()
        ae_running = em_running = True
        for epoch in range(epochs):
            if not em_running and not ae_running:
                break
            if ae_running:
                cbs[0].on_epoch_begin(epoch)
            if em_running:
                cbs[1].on_epoch_begin(epoch)
            tra.set_lr(float(ae.optimizer.lr) if ae_running else 0.0)
            tre.set_lr(float(em.optimizer.lr))
            perm = engine._rng.permutation(n).astype(np.int32)
            if dp:
                perm = bcast(perm)
            if em_running:
                la, le = joint.run_epoch(perm, batch_size)
            else:  # the emulator has stopped: the autoencoder goes on alone
                la, le = tra.run_epoch(perm, batch_size), None
            ae._dirty_host = em._dirty_host = True
            # both validation passes in one launch; the emulator's against the current encoder's latents of the
            # validation signals (the reference's encoder.predict(signal_val), emulator.py:754, without leaving the device)
            va, ve = joint.evaluate() if em_running else (tra.evaluate(1, batch_size), None)
            if ae_running:
                logs = {"loss": la, "val_loss": va}
                cbs[0].on_epoch_end(epoch, logs)
                if ae.stop_training:
                    ae_running = False  # frozen from here on: the reference's phase 2
            if em_running:
                logs = {"loss": le, "val_loss": ve}
                cbs[1].on_epoch_end(epoch, logs)
                if em.stop_training:
                    em_running = False
            if verbose in (1, 2):
                print("Epoch %d/%d - ae loss %.4e - emulator loss %s" % (epoch + 1, epochs, la, "%.4e" % le if le is not None else "-"))
        for c in cbs:
            c.on_train_end()
        ae.optimizer.iterations = tra.get_state()[0]
        em.optimizer.iterations = tre.get_state()[0]
        ae._sync_host(); em._sync_host()
        self._chain_model = None
        return hists[0].history["loss"], hists[0].history["val_loss"], hists[1].history["loss"], hists[1].history["val_loss"]

    def _predict_chain(self):
        blocks = [self.emulator, self.autoencoder.decoder]
        cm = self._chain_model
        if cm is None or cm._blocks[0] is not blocks[0] or cm._blocks[1] is not blocks[1]:
            cm = ChainedModel(blocks, name="ae_emulator_decoder")
            self._chain_model = cm
        cm.precision = self.emulator.precision
        return cm

    def predict(self, params, devices=None):
        """emulator.predict then decoder.predict then unpreproc (emulator.py:788-795),
        evaluated as ONE fused device stack 7 -> ... -> 9 -> 32 -> 352 -> 451."""
        return self._predict_stack(self._predict_chain(), params, devices)

    def test_error(self, use_autoencoder=False, relative=True, flow=None, fhigh=None):
        if use_autoencoder:
            pred = pp.unpreproc(self.autoencoder(pp.preproc(self.signal_test, self.signal_train)), self.signal_train)
        else:
            pred = self.predict(self.par_test)
        return error(self.signal_test, pred, relative=relative, nu_arr=self.frequencies, flow=flow, fhigh=fhigh)


# names used by BASELINE.json's north_star
Emulator = DirectEmulator
VAEEmulator = AutoEncoderEmulator
