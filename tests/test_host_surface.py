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
