"""Seeded synthetic stand-ins for the reference's dataset.

The real ``dataset_21cmVAE.h5`` (emulator.py:198-204 of the reference) is not available
offline, so benchmarks and tests use arrays of the same shapes and dtypes:
seven astrophysical parameters per row (the 21cmGEM box, order = ``par_labels``,
emulator.py:293-301) and smooth 451-bin float32 signals on z = linspace(5, 50, 451)
(emulator.py:197).  Values are synthetic; they affect loss values, not throughput.
"""
import numpy as np

N_TRAIN, N_VAL, N_TEST = 24562, 2730, 1704
N_BINS, N_PAR = 451, 7

# (low, high, log-uniform?) per parameter: fstar, Vc, fx, tau, alpha, nu_min, Rmfp
_BOX = [(1e-4, 0.5, True), (4.2, 100.0, True), (1e-6, 1e3, True), (0.04, 0.2, False),
        (1.0, 1.5, False), (0.1, 3.0, False), (10.0, 50.0, False)]


def make_params(n, seed, zero_fx_frac=0.01, corners=False, dtype=np.float64):
    """(n,7) parameter rows.  1 % of rows get fx == 0 exactly (exercises the
    fx==0 -> 1e-6 branch of preprocess.par_transform); ``corners`` forces rows 0 and 1
    onto the box corners so train min/max are exact."""
    rng = np.random.default_rng(seed)
    p = np.empty((n, N_PAR))
    for j, (lo, hi, logu) in enumerate(_BOX):
        u = rng.uniform(size=n)
        p[:, j] = 10 ** (np.log10(lo) + u * (np.log10(hi) - np.log10(lo))) if logu else lo + u * (hi - lo)
    if zero_fx_frac > 0 and n >= 8:
        k = max(1, int(round(zero_fx_frac * n)))
        p[rng.choice(np.arange(2, n), size=min(k, n - 2), replace=False), 2] = 0.0
    if corners and n >= 2:
        p[0] = [b[0] for b in _BOX]
        p[1] = [b[1] for b in _BOX]
    return p.astype(dtype)


def signal_basis(seed=2, n_comp=9):
    """Fixed smooth (n_comp, 451) basis: Gaussians in redshift of varying centre/width."""
    rng = np.random.default_rng(seed)
    z = np.linspace(5, 50, N_BINS)
    c = rng.uniform(8, 40, size=n_comp)
    s = rng.uniform(2, 9, size=n_comp)
    return np.exp(-0.5 * ((z[None, :] - c[:, None]) / s[:, None]) ** 2)


def make_signals(n, seed, basis_seed=2):
    """(n,451) float32 smooth curves, mK-like scale (troughs of tens to ~200 mK)."""
    rng = np.random.default_rng(seed)
    B = signal_basis(basis_seed)
    Z = rng.normal(size=(n, B.shape[0]))
    return (Z @ B * 50.0 - 40.0).astype(np.float32)


def signals_from_params(par, basis_seed=2, map_seed=4):
    """(n,451) float32 signals that are a smooth deterministic function of the parameters
    (so that an emulator has something to learn): box-normalised log-parameters ->
    tanh of a fixed random linear map -> the smooth basis."""
    par = np.asarray(par, dtype=np.float64)
    u = np.empty_like(par)
    for j, (lo, hi, logu) in enumerate(_BOX):
        col = par[:, j].copy()
        if logu:
            col[col == 0] = lo
            u[:, j] = (np.log10(col) - np.log10(lo)) / (np.log10(hi) - np.log10(lo))
        else:
            u[:, j] = (col - lo) / (hi - lo)
    B = signal_basis(basis_seed)
    A = np.random.default_rng(map_seed).normal(size=(N_PAR, B.shape[0]))
    Z = np.tanh((2 * u - 1) @ A)
    return (Z @ B * 50.0 - 40.0).astype(np.float32)


def make_dataset(n_train=N_TRAIN, n_val=N_VAL, n_test=N_TEST, seed=1):
    """The six arrays the reference reads at import, same names (emulator.py:198-204);
    signals are a smooth function of the parameters."""
    pt, pv, pe = make_params(n_train, seed, corners=True), make_params(n_val, seed + 100), make_params(n_test, seed + 200)
    return dict(par_train=pt, par_val=pv, par_test=pe, signal_train=signals_from_params(pt),
                signal_val=signals_from_params(pv), signal_test=signals_from_params(pe))


def save_dataset(path, data=None):
    """Write the six arrays in the layout of ``dataset_21cmVAE.h5`` (emulator.py:198-204) with the
    built-in HDF5 writer, so the no-argument constructors (``$V21_DATASET``) can be tried on synthetic data."""
    from . import h5write
    data = make_dataset() if data is None else data
    f = h5write.FileW()
    for k in ("par_train", "par_val", "par_test", "signal_train", "signal_val", "signal_test"):
        f.create_dataset(k, np.ascontiguousarray(data[k]))
    f.write(path)
    return path
