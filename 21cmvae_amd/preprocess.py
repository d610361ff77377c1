"""Pre/post-processing of parameters and signals -- host side of the hot path.

Same public functions and results as the reference's ``VeryAccurateEmulator/preprocess.py``
(``preproc`` :4-24, ``unpreproc`` :27-46, ``par_transform`` :49-110), restated around two
small statistics records so that the training-set reductions (a per-bin mean + one
global std over ~11 M floats, a log/min/max over the parameter table) are computed
once and cached instead of on every ``predict`` call, and so that the same numbers can
be handed to the fused HIP prologue/epilogue (``v21_affine_in`` / ``v21_affine_out``).
"""
import weakref

import numpy as np

LOG_COLUMNS = (0, 1, 2)    # fstar, Vc, fx are emulated in log10 (preprocess.py:77-78)
ZERO_FLOOR = {2: 1e-6}     # fx == 0 -> 1e-6 before the log (preprocess.py:76)

_cache = {}

try:  # a fast non-cryptographic hash (offline wheelhouse); any 64+-bit hash of the whole buffer will do
    import xxhash

    def _digest(buf):
        return xxhash.xxh3_128_digest(buf)
except ImportError:  # pragma: no cover
    import hashlib

    def _digest(buf):
        return hashlib.blake2b(buf, digest_size=16).digest()


def freeze(arr):
    """A private read-only copy of `arr`.  Nobody holds a writable handle on it, so its
    statistics are cached on identity alone (no per-call checksum): what a sampler that calls
    ``predict`` once per parameter vector wants (``DirectEmulator(..., freeze_data=True)``)."""
    out = np.array(arr, copy=True)
    out.flags.writeable = False
    return out


def _is_frozen(a):
    return isinstance(a, np.ndarray) and not a.flags.writeable and a.base is None and a.flags.owndata


def _fingerprint(arr):
    """(shape, dtype, 128-bit hash of EVERY byte): the reference recomputes the training-set
    statistics on each call (preprocess.py:22-23, 44-45, 89-101); the cache may only answer when
    the buffer is bit-for-bit the one the statistics were computed from."""
    a = np.asarray(arr)
    if _is_frozen(a):
        return (a.shape, str(a.dtype), "frozen")
    c = a if a.flags.c_contiguous else np.ascontiguousarray(a)
    return (a.shape, str(a.dtype), _digest(c.reshape(-1).view(np.uint8).data))


def _cached(kind, arr, build):
    """Statistics keyed on the identity of the training array plus a checksum of the whole
    buffer, so an array edited in place -- anywhere -- is noticed.  One O(N) pass of a fast hash
    replaces the reference's mean/std/log/min/max passes per call."""
    key = (kind, id(arr))
    hit = _cache.get(key)
    fp = _fingerprint(arr)
    if hit is not None and hit[0]() is arr and hit[2] == fp:
        return hit[1]
    val = build(arr)
    try:
        _cache[key] = (weakref.ref(arr, lambda _r, k=key: _cache.pop(k, None)), val, fp)
    except TypeError:  # not weak-referenceable (e.g. a list): compute every time
        pass
    return val


class SignalStats:
    """mean over the training set per bin and the scalar std over all entries."""

    def __init__(self, signal_train):
        signal_train = np.asarray(signal_train)
        self.mean = np.mean(signal_train, axis=0)
        self.std = np.std(signal_train)

    @classmethod
    def of(cls, signal_train):
        return _cached("sig", signal_train, cls)


class ParamStats:
    """Column minima/maxima of the log-transformed training parameters (float64)."""

    def __init__(self, params_train):
        t = _to_log_space(np.asarray(params_train))
        self.lo = t.min(axis=0)
        self.hi = t.max(axis=0)
        n = t.shape[1]
        self.log_mask = [j in LOG_COLUMNS for j in range(n)]
        self.zero_floor = [ZERO_FLOOR.get(j, 0.0) for j in range(n)]

    @classmethod
    def of(cls, params_train):
        return _cached("par", params_train, cls)


def _to_log_space(p):
    """log10 of the LOG_COLUMNS, taken IN THE DTYPE OF THE INPUT for floating arrays, as the reference
    does (preprocess.py:74-78 / 89-93: `.copy()` of the columns, the floor written into that copy,
    `np.log10` of it, and only then the cast into a float64 result, :81-85): float32 parameters get a
    float32 floor (1e-6 rounded to float32) and a float32 logarithm.  Documented deviation: a
    non-floating array is taken to float64 first (the reference would truncate the floor to 0 and
    return -inf)."""
    out = np.empty(p.shape)  # float64 whatever the input dtype, like the reference
    src = p if np.issubdtype(p.dtype, np.floating) else p.astype(np.float64)
    for j in range(p.shape[1]):
        col = src[:, j]
        if j in LOG_COLUMNS:
            col = np.array(col, copy=True, order="C")  # contiguous, input dtype
            if j in ZERO_FLOOR:
                col[col == 0] = ZERO_FLOOR[j]
            col = np.log10(col)
        out[:, j] = col
    return out


def preproc(signal, signal_train):
    """Subtract the per-bin training mean, divide by the global training std."""
    st = SignalStats.of(signal_train)
    out = np.array(signal, copy=True)
    out -= st.mean
    out /= st.std
    return out


def unpreproc(signal, signal_train):
    """Inverse of :func:`preproc`."""
    st = SignalStats.of(signal_train)
    out = signal * st.std
    out += st.mean
    return out


def par_transform(parameters, params_train):
    """log10 of the first three columns (fx == 0 -> 1e-6), then the affine map that sends
    the training box to [-1, 1].  1-D input becomes one row; float64 out."""
    p = np.asarray(parameters)
    if p.ndim == 1:
        p = p[None, :]
    st = ParamStats.of(params_train)
    q = _to_log_space(p)
    q -= st.lo
    q /= st.hi - st.lo
    q *= 2
    q -= 1
    return q
