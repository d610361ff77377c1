"""Host-side behaviour of the class surface that needs no GPU: the statistics cache must never
answer with numbers the reference would not compute (it recomputes them on every call:
/root/reference/VeryAccurateEmulator/preprocess.py:22-23, 44-45, 89-101), and the
``strict_reference`` switch restores the reference's three accidental behaviours
(INTEGRATION.md section 6)."""
import numpy as np
import pytest

from conftest import pkg
from oracle import ref_numpy as ora


def test_statistics_cache_notices_any_in_place_edit():
    pp = pkg("preprocess")
    rng = np.random.default_rng(0)
    sig = rng.normal(size=(3001, 451)).astype(np.float32)
    par = pkg("synth").make_params(500, seed=1, corners=True)
    s0 = pp.SignalStats.of(sig)
    assert pp.SignalStats.of(sig) is s0  # unchanged buffer: served from the cache
    # one element at a time, at positions a strided sample would miss
    for pos in [(0, 1), (17, 3), (1500, 225), (3000, 450)]:
        sig[pos] += 1.0
        s1 = pp.SignalStats.of(sig)
        assert s1 is not s0
        np.testing.assert_array_equal(s1.mean, np.mean(sig, axis=0))
        assert s1.std == np.std(sig)
        s0 = s1
    # a shift leaves std unchanged and must still re-centre preproc/unpreproc
    before = pp.unpreproc(np.zeros((1, 451), np.float32), sig)
    sig += 5.0
    after = pp.unpreproc(np.zeros((1, 451), np.float32), sig)
    np.testing.assert_allclose(after - before, 5.0, rtol=0, atol=1e-4)
    np.testing.assert_array_equal(after, ora.unpreproc(np.zeros((1, 451), np.float32), sig))
    p0 = pp.ParamStats.of(par)
    par[3, 5] = 1e3  # a new column maximum
    p1 = pp.ParamStats.of(par)
    assert p1 is not p0 and p1.hi[5] == 1e3
    np.testing.assert_array_equal(pp.par_transform(par[:7], par), ora.par_transform(par[:7], par))


def test_frozen_copies_are_cached_on_identity_and_cannot_change():
    pp = pkg("preprocess")
    sig = np.random.default_rng(1).normal(size=(100, 451)).astype(np.float32)
    f = pp.freeze(sig)
    assert f is not sig and not f.flags.writeable
    s0 = pp.SignalStats.of(f)
    assert pp.SignalStats.of(f) is s0
    with pytest.raises(ValueError):
        f[0, 0] = 1.0
    sig[0, 0] += 100.0  # the caller's array is no longer connected to the frozen copy
    assert pp.SignalStats.of(f) is s0
    # a read-only VIEW of a writable array is not "frozen": its base can still be written
    v = sig.view()
    v.flags.writeable = False
    a = pp.SignalStats.of(v)
    sig[1, 1] += 3.0
    assert pp.SignalStats.of(v) is not a


def test_emulators_hold_the_callers_training_arrays_locked_by_default(monkeypatch):
    """r5 contract of the class surface (INTEGRATION.md sections 1 and 7): an emulator holds THE CALLER'S OWN
    par_train / signal_train, as the reference does (`em.signal_train is signal_train`, nothing held twice), with their
    writeable flag switched off -- so the statistics cache is exact WITHOUT a per-call checksum: an in-place edit raises,
    assigning a new array is seen, and an array the user unlocks again is re-hashed on every call.  freeze_data=True:
    private read-only copies (r4's default); freeze_data=False: the caller's writable arrays and the whole-buffer hash."""
    emu, pp, synth = pkg("emulator"), pkg("preprocess"), pkg("synth")
    data = synth.make_dataset(n_train=400, n_val=20, n_test=20, seed=2)
    em = emu.DirectEmulator(hidden_dims=[8], **data)
    for k in ("par_train", "signal_train"):
        a = getattr(em, k)
        assert a is data[k] and not a.flags.writeable          # the caller's object, locked
    assert em.par_val is data["par_val"] and data["par_val"].flags.writeable   # (only the two arrays with cached statistics)
    s0, p0 = pp.SignalStats.of(em.signal_train), pp.ParamStats.of(em.par_train)
    calls = []
    monkeypatch.setattr(pp, "_digest", lambda buf: calls.append(1) or b"x")
    assert pp.SignalStats.of(em.signal_train) is s0 and pp.ParamStats.of(em.par_train) is p0 and not calls   # no hash
    with pytest.raises(ValueError, match="read-only"):
        em.signal_train[0, 0] = 1.0
    with pytest.raises(ValueError, match="read-only"):
        data["signal_train"][0, 0] += 50.0                    # ... through the caller's name as well: it IS that array
    monkeypatch.undo()
    # the user unlocks the array: from then on it is re-hashed per call, and an edit in place is seen
    data["signal_train"].setflags(write=True)
    data["signal_train"][0, 0] += 50.0
    s_edit = pp.SignalStats.of(em.signal_train)
    assert s_edit is not s0 and s_edit.mean[0] == np.mean(data["signal_train"], axis=0)[0]
    new = data["signal_train"].copy()
    new[1, 1] -= 7.0
    em.signal_train = new                                     # assignment: a new identity, new statistics
    s1 = pp.SignalStats.of(em.signal_train)
    assert s1 is not s_edit and s1.mean[1] == np.mean(new, axis=0)[1]
    # a view does not own its buffer (the base could still be written): held by reference, hashed per call
    base = synth.make_dataset(n_train=420, n_val=20, n_test=20, seed=3)
    view = base["signal_train"][:400]
    em_v = emu.DirectEmulator(hidden_dims=[8], **dict(data, signal_train=view, par_train=base["par_train"][:400]))
    assert em_v.signal_train is view and view.flags.writeable
    a = pp.SignalStats.of(em_v.signal_train)
    base["signal_train"][3, 3] += 1.0
    assert pp.SignalStats.of(em_v.signal_train) is not a
    # r4's default on request: private read-only copies, the caller's arrays untouched
    data2 = synth.make_dataset(n_train=400, n_val=20, n_test=20, seed=4)
    em_c = emu.DirectEmulator(hidden_dims=[8], freeze_data=True, **data2)
    assert em_c.signal_train is not data2["signal_train"] and not em_c.signal_train.flags.writeable and data2["signal_train"].flags.writeable
    # the reference's by-reference semantics to the letter
    data3 = synth.make_dataset(n_train=400, n_val=20, n_test=20, seed=5)
    em2 = emu.AutoEncoderEmulator(freeze_data=False, enc_hidden_dims=[8], dec_hidden_dims=[8], em_hidden_dims=[8], **data3)
    assert em2.signal_train is data3["signal_train"] and em2.par_train is data3["par_train"] and data3["signal_train"].flags.writeable
    a = pp.SignalStats.of(em2.signal_train)
    data3["signal_train"][3, 3] += 1.0
    assert pp.SignalStats.of(em2.signal_train) is not a       # the in-place edit is seen (whole-buffer hash)


def test_strict_reference_switch():
    emu = pkg("emulator")
    nu = np.linspace(50.0, 200.0, 451)
    rng = np.random.default_rng(0)
    t, p = rng.normal(size=(6, 451)), rng.normal(size=(6, 451))
    try:
        emu.set_strict_reference(False)
        assert emu.error(t, p, nu_arr=nu, flow=60.0).shape == (6,)
        assert emu.error(t, p, nu_arr=nu, fhigh=90.0).shape == (6,)
        fr = nu.copy()
        z = emu.freq2redshift(fr)
        np.testing.assert_array_equal(fr, nu)  # argument untouched
        emu.set_strict_reference(True)
        # emulator.py:179-184: a single bound indexes with an (k, 1) array -> (N, 1); both bounds -> (N,)
        e1 = emu.error(t, p, nu_arr=nu, flow=60.0)
        assert e1.shape == (6, 1)
        assert emu.error(t, p, nu_arr=nu, fhigh=90.0).shape == (6, 1)
        assert emu.error(t, p, nu_arr=nu, flow=60.0, fhigh=90.0).shape == (6,)
        emu.set_strict_reference(False)
        np.testing.assert_allclose(e1[:, 0], emu.error(t, p, nu_arr=nu, flow=60.0))
        emu.set_strict_reference(True)
        # emulator.py:124: ndarray argument converted to Hz in place
        z2 = emu.freq2redshift(fr)
        np.testing.assert_allclose(fr, nu * 1e6)
        np.testing.assert_allclose(z2, z)
        assert abs(emu.freq2redshift(emu.redshift2freq(30.0)) - 30.0) < 1e-9  # tests/test_emulator.py:36-39
    finally:
        emu.set_strict_reference(False)


class _FakeTrainer:
    """Stands in for _native.Trainer in the fit() loop (no GPU): records the permutation of every epoch."""
    max_batch = 1 << 20

    class ctx:
        nranks, rank, device = 1, 0, 0

    def __init__(self, fail_at=None):
        self.perms, self.fail_at = [], fail_at

    def set_data(self, *a): pass
    def set_lr(self, lr): pass
    def get_state(self): return len(self.perms), None, None

    def run_epoch(self, perm, batch):
        if self.fail_at is not None and len(self.perms) == self.fail_at:
            raise RuntimeError("device lost")
        self.perms.append(None if perm is None else perm.copy())
        return 1.0 / (1 + len(self.perms))

    def evaluate(self, which, batch): return 0.5


def _fit_with_fake_trainer(eng, monkeypatch, epochs, callbacks=(), fail_at=None, seed=None):
    m = eng.Sequential([eng.Input((3,)), eng.Dense(4, "relu"), eng.Dense(2)])  # (draws its Glorot kernels from the shared stream)
    m.compile(optimizer="adam", loss="mse")
    if seed is not None:
        eng.set_random_seed(seed)
    fake = _FakeTrainer(fail_at)
    monkeypatch.setattr(m, "_ensure_trainer", lambda b: fake)
    x = np.zeros((50, 3), np.float32); y = np.zeros((50, 2), np.float32)
    m.fit(x, y, batch_size=16, epochs=epochs, callbacks=list(callbacks))
    return fake


def test_fit_shuffle_stream_is_the_sequential_one_despite_the_look_ahead(monkeypatch):
    """engine.Model.fit draws the next epoch's permutation while the current one runs -- from a COPY of the shared
    generator.  The permutations, and the shared generator's state afterwards, must be exactly those of
    drawing one permutation at the start of every epoch: (a) plain run; (b) a callback that draws from the shared
    generator between epochs; (c) early stopping (the unused look-ahead leaves no trace); (d) set_random_seed from a
    callback; (e) run_epoch raising (the worker pool is shut down, the generator is not ahead)."""
    eng = pkg("engine")

    # (a)
    ref = np.random.default_rng(5)
    fake = _fit_with_fake_trainer(eng, monkeypatch, 4, seed=5)
    for p in fake.perms:
        np.testing.assert_array_equal(p, ref.permutation(50).astype(np.int32))
    assert eng._rng.bit_generator.state == ref.bit_generator.state

    # (b) + (c): a callback uses the shared stream after every epoch and stops the fit after the third
    class Draws:
        def __init__(self): self.vals = []
        def set_model(self, m): self.model = m
        def on_epoch_end(self, epoch, logs=None):
            self.vals.append(eng._rng.integers(0, 1 << 30))
            if epoch == 2:
                self.model.stop_training = True
    ref = np.random.default_rng(6)
    cb = Draws()
    fake = _fit_with_fake_trainer(eng, monkeypatch, 10, [cb], seed=6)
    assert len(fake.perms) == 3
    for p, v in zip(fake.perms, cb.vals):
        np.testing.assert_array_equal(p, ref.permutation(50).astype(np.int32))
        assert v == ref.integers(0, 1 << 30)
    assert eng._rng.bit_generator.state == ref.bit_generator.state

    # (d) a callback re-seeds: the look-ahead drawn from the old generator must not be used
    class Reseed:
        def set_model(self, m): pass
        def on_epoch_end(self, epoch, logs=None):
            if epoch == 0:
                eng.set_random_seed(99)
    fake = _fit_with_fake_trainer(eng, monkeypatch, 3, [Reseed()], seed=7)
    ref = np.random.default_rng(99)
    for p in fake.perms[1:]:
        np.testing.assert_array_equal(p, ref.permutation(50).astype(np.int32))

    # (e) run_epoch raises in the second epoch
    ref = np.random.default_rng(8)
    with pytest.raises(RuntimeError, match="device lost"):
        _fit_with_fake_trainer(eng, monkeypatch, 5, fail_at=1, seed=8)
    ref.permutation(50); ref.permutation(50)  # two epochs began, two permutations were consumed
    assert eng._rng.bit_generator.state == ref.bit_generator.state


def test_native_handles_are_destroyed_children_first_whatever_order_python_finalizes_them():
    """`_native._Owned`: a Stack outlives its Trainers, a Trainer the Joint / Sweep built on it.  At interpreter exit the
    collector finalizes a script's objects in any order (they hang in one cycle through the module namespace); a Trainer
    destroyed before its Joint was a use-after-free in the library (r4: segmentation fault at exit, found by
    scripts/diag/joint_fuzz.py).  No GPU: the destroy callbacks only record their order."""
    native = pkg("_native")
    order = []

    class H(native._Owned):
        def __init__(self, name, parents=()):
            self.h, self.name = name, name
            self._own(lambda h: order.append(h), parents)

    for finalize in (["stack", "ae", "em", "joint"], ["joint", "em", "ae", "stack"], ["ae", "stack", "joint", "em"]):
        order.clear()
        stack = H("stack")
        ae, em = H("ae", [stack]), H("em", [stack])
        joint = H("joint", [ae, em])
        objs = {"stack": stack, "ae": ae, "em": em, "joint": joint}
        for name in finalize:
            objs[name].__del__()          # what the collector does, in this order
        assert sorted(order) == ["ae", "em", "joint", "stack"], order
        assert order[0] == "joint" and order[-1] == "stack", (finalize, order)
        for o in objs.values():
            o.__del__()                   # a second finalization (resurrection, explicit call) destroys nothing twice
        assert len(order) == 4
    # a handle that is still in use is kept: the stack lives while one trainer does
    order.clear()
    stack = H("stack"); t1, t2 = H("t1", [stack]), H("t2", [stack])
    stack.__del__(); t1.__del__()
    assert order == ["t1"]
    t2.__del__()
    assert order == ["t1", "t2", "stack"]
