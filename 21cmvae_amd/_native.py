"""ctypes binding of libv21.so (C ABI: include/v21.h).

The product path has NO CPU fallback: if the HIP library is missing or no GPU is
visible, the first call that needs the engine raises ``EngineUnavailable``.  Loading
the library and listing its symbols works without a GPU (used by the CPU test suite).
"""
import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("V21_LIB") or os.path.join(HERE, "libv21.so")

PREC_F32, PREC_F16, PREC_BF16 = 0, 1, 2
PRECISIONS = {"f32": PREC_F32, "fp32": PREC_F32, "float32": PREC_F32,
              "f16": PREC_F16, "fp16": PREC_F16, "float16": PREC_F16,
              "bf16": PREC_BF16, "bfloat16": PREC_BF16}
ACT_LINEAR, ACT_RELU, ACT_GAUSS = 0, 1, 2
FWD_IN_TRANSFORM, FWD_OUT_TRANSFORM, FWD_FORCE_GENERIC, FWD_NO_SMALL, FWD_FORCE_CHAIN, FWD_FORCE_JIT = 1, 2, 4, 8, 16, 32
COMM_ID_BYTES = 128


class EngineUnavailable(RuntimeError):
    """libv21.so cannot be loaded or no MI355X is visible."""


class EngineError(RuntimeError):
    """A v21_* call returned a negative status."""


class AffineIn(C.Structure):
    _fields_ = [("n", C.c_int32), ("log_mask", C.c_int32 * 8), ("zero_floor", C.c_double * 8),
                ("lo", C.c_double * 8), ("span", C.c_double * 8)]


class AffineOut(C.Structure):
    _fields_ = [("std", C.c_float), ("mean", C.POINTER(C.c_float)), ("n", C.c_int32)]


_HOST_AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_size_t)


class CommHostOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allreduce_sum_f32", _HOST_AR), ("reduce_scatter_sum_f32", _HOST_AR),
                ("allgather_f32", _HOST_AR)]


class Adam(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]


_P = C.c_void_p
_F = C.POINTER(C.c_float)
# name -> (restype, argtypes); every symbol include/v21.h declares
SIGNATURES = {
    "v21_last_error": (C.c_char_p, []),
    "v21_version": (C.c_int, []),
    "v21_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "v21_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "v21_ctx_destroy": (C.c_int, [_P]),
    "v21_ctx_sync": (C.c_int, [_P]),
    "v21_ctx_set_stream": (C.c_int, [_P, _P]),
    "v21_ctx_get_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "v21_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "v21_free": (C.c_int, [_P, _P]),
    "v21_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "v21_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "v21_memset": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "v21_event_create": (C.c_int, [_P, C.POINTER(_P)]),
    "v21_event_destroy": (C.c_int, [_P, _P]),
    "v21_event_record": (C.c_int, [_P, _P]),
    "v21_event_elapsed_ms": (C.c_int, [_P, _P, _P, C.POINTER(C.c_float)]),
    "v21_mlp_create": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_P)]),
    "v21_mlp_destroy": (C.c_int, [_P]),
    "v21_mlp_num_params": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "v21_mlp_set_weights": (C.c_int, [_P, _F, C.c_size_t]),
    "v21_mlp_get_weights": (C.c_int, [_P, _F, C.c_size_t]),
    "v21_mlp_set_input_transform": (C.c_int, [_P, C.POINTER(AffineIn)]),
    "v21_mlp_set_output_transform": (C.c_int, [_P, C.POINTER(AffineOut)]),
    "v21_mlp_has_fused": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "v21_mlp_jit": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "v21_jit_prebuild": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_char_p]),
    "v21_mlp_forward": (C.c_int, [_P, _P, C.c_int, C.c_int64, _F, C.c_int, C.c_int]),
    "v21_mlp_forward_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int, C.c_int]),
    "v21_trainer_create": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "v21_trainer_destroy": (C.c_int, [_P]),
    "v21_trainer_set_adam": (C.c_int, [_P, C.POINTER(Adam)]),
    "v21_trainer_set_lr": (C.c_int, [_P, C.c_float]),
    "v21_trainer_get_lr": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "v21_trainer_set_data": (C.c_int, [_P, C.c_int, _F, _F, _F, C.c_int64]),
    "v21_trainer_run_epoch": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double)]),
    "v21_trainer_eval": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "v21_trainer_step_dev": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int]),
    "v21_trainer_last_step_loss": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "v21_trainer_get_state": (C.c_int, [_P, C.POINTER(C.c_int64), _F, _F, C.c_size_t]),
    "v21_trainer_set_state": (C.c_int, [_P, C.c_int64, _F, _F, C.c_size_t]),
    "v21_trainer_get_grad": (C.c_int, [_P, _F, C.c_size_t]),
    "v21_trainer_use_graph": (C.c_int, [_P, C.c_int]),
    "v21_debug_poison_lds": (C.c_int, [_P, C.c_uint32]),
    "v21_debug_check_chain_jobs": (C.c_int, [_P, C.c_longlong, C.c_longlong]),
    "v21_debug_trainer_counters": (C.c_int, [_P, C.POINTER(C.c_longlong)]),
    "v21_route_forward": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "v21_route_train": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "v21_trainer_jit": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "v21_mlp_last_route": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "v21_trainer_last_route": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "v21_route_name": (C.c_char_p, [C.c_int, C.c_int]),
    "v21_trainer_get_data_dev": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64)]),
    "v21_debug_clock_probe_start": (C.c_int, [_P, C.c_double, C.c_double]),
    "v21_debug_forward_clocked": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "v21_debug_clock_probe_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "v21_trainer_set_vae": (C.c_int, [_P, C.c_float, C.c_int, C.c_uint64]),
    "v21_trainer_enable_stamps": (C.c_int, [_P, C.c_int]),
    "v21_trainer_chain_stamps": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int]),
    "v21_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "v21_host_free": (C.c_int, [_P, _P]),
    "v21_sweep_create": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(_P)]),
    "v21_sweep_destroy": (C.c_int, [_P]),
    "v21_sweep_run_epoch": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double)]),
    "v21_joint_create": (C.c_int, [_P, _P, C.c_int, C.POINTER(_P)]),
    "v21_joint_destroy": (C.c_int, [_P]),
    "v21_joint_run_epoch": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double)]),
    "v21_joint_eval": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "v21_comm_get_unique_id": (C.c_int, [_P, _P]),
    "v21_comm_init": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "v21_comm_init_host": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(CommHostOps)]),
    "v21_comm_destroy": (C.c_int, [_P]),
    "v21_comm_init_null": (C.c_int, [_P, C.c_int, C.c_int]),
    "v21_comm_set_buckets": (C.c_int, [_P, C.c_int]),
    "v21_trainer_phase_timing": (C.c_int, [_P, C.c_int, C.c_int]),
    "v21_trainer_phase_times": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "v21_comm_set_sharded": (C.c_int, [_P, C.c_int]),
    "v21_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "v21_comm_allreduce_f32": (C.c_int, [_P, _P, C.c_size_t]),
    "v21_comm_reduce_scatter_f32": (C.c_int, [_P, _P, C.c_size_t]),
    "v21_comm_allgather_f32": (C.c_int, [_P, _P, C.c_size_t]),
}

# ---- routes (include/v21.h: v21_route_*; csrc/routes.h): the names the tests and INTEGRATION.md section 6 use
FWD_ROUTES = {1: "small", 2: "fused", 3: "fused_rt", 4: "table", 5: "generic"}
TRAIN_FWD_ROUTES = {1: "per_layer", 2: "chain16", 3: "fused128", 4: "fused64", 5: "chain32", 6: "chain32s_8", 7: "chain32s_4"}
TRAIN_UPD_ROUTES = {1: "per_layer", 2: "dw16_adam", 3: "dw16_splitk", 4: "dwadam32", 5: "nt_dwadam", 6: "nt_sliced"}


def route_forward(dims, act, precision, n, flags=0, rt_ready=False):
    """The route v21_mlp_forward_dev takes for n rows of this stack (pure host logic: no GPU).  -> name of FWD_ROUTES."""
    L = len(act)
    r = C.c_int(0)
    check(load_library().v21_route_forward(L, (C.c_int * (L + 1))(*[int(d) for d in dims]), (C.c_int * L)(*[int(a) for a in act]),
                                           precision_id(precision), int(n), int(flags), 1 if rt_ready else 0, C.byref(r)))
    return FWD_ROUTES[r.value]


def route_train(dims, act, precision, max_batch, rows, nranks=1, rt_ready=False):
    """The kernels one optimizer step of `rows` rows takes for a trainer created with max_batch on nranks ranks (pure host
    logic: no GPU; rt_ready: the run-time instantiated fused training kernel of a stack outside archs.h has arrived).
    -> (name of TRAIN_FWD_ROUTES, name of TRAIN_UPD_ROUTES)."""
    L = len(act)
    f, u = C.c_int(0), C.c_int(0)
    check(load_library().v21_route_train(L, (C.c_int * (L + 1))(*[int(d) for d in dims]), (C.c_int * L)(*[int(a) for a in act]),
                                         precision_id(precision), int(max_batch), int(rows), int(nranks), 1 if rt_ready else 0, C.byref(f), C.byref(u)))
    return TRAIN_FWD_ROUTES[f.value], TRAIN_UPD_ROUTES[u.value]


_lib = None
_lock = threading.Lock()


def load_library():
    """dlopen libv21.so and attach prototypes.  Needs no GPU."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise EngineUnavailable(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C 21cmvae_amd/csrc)" % LIB_PATH)
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # missing libamdhip64 etc.
            raise EngineUnavailable("cannot load %s: %s" % (LIB_PATH, e)) from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
        return lib


def check(status):
    if status != 0:
        msg = load_library().v21_last_error()
        raise EngineError("v21 error %d: %s" % (status, msg.decode() if msg else "?"))


def precision_id(p):
    if isinstance(p, int):
        return p
    try:
        return PRECISIONS[str(p).lower()]
    except KeyError:
        raise ValueError("unknown precision %r (use f32, f16 or bf16)" % (p,)) from None


def _row_table(perm, n_rows, who):
    """An epoch's row table as the C ABI takes it: int32, contiguous, one entry per row of the training set (the library
    reads exactly that many entries from the pointer: a shorter array would be read past its end)."""
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    if n_rows is not None and perm.shape != (n_rows,):
        raise ValueError("%s: the row table has shape %s, the training set has %d rows" % (who, perm.shape, n_rows))
    return perm


def _fptr(a):
    return a.ctypes.data_as(_F)


class _Owned:
    """A C handle that must outlive the handles built on it (a Stack its Trainers, a Trainer the Joint / Sweep it is
    part of).  Python gives no such order: when a script ends, everything reachable from its module namespace is one
    unreachable cycle (every function defined there refers to it) and the collector calls the finalizers of such a set
    in ANY order -- a Trainer destroyed before its Joint was a use-after-free in v21_joint_destroy (r4: found by
    scripts/diag/joint_fuzz.py as a segmentation fault at interpreter exit).  So the wrappers count: finalizing a
    wrapper only marks it; its handle is destroyed once nothing built on it is left, children first."""

    def _own(self, destroy, parents=()):
        self._destroy, self._parents, self._children, self._finalized = destroy, list(parents), 0, False
        for p in self._parents:
            p._children += 1

    def _release(self):
        self._finalized = True
        self._try_destroy()

    def _try_destroy(self):
        if not self._finalized or self._children > 0 or not getattr(self, "h", None):
            return
        h, self.h = self.h, None
        try:
            self._destroy(h)
        except Exception:
            pass
        parents, self._parents = self._parents, []
        for p in parents:
            p._children -= 1
            p._try_destroy()

    def __del__(self):
        try:
            if hasattr(self, "_finalized"):
                self._release()
        except Exception:
            pass


class Context:
    """One per device.  Owns a HIP stream; calls on one context are serialised."""
    _default = {}

    def __init__(self, device=0):
        self.lib = load_library()
        n = C.c_int(0)
        st = self.lib.v21_device_count(C.byref(n))
        if st != 0 or n.value < 1:
            raise EngineUnavailable("no HIP device visible (v21_device_count -> %d, n=%d): the MI355X "
                                    "engine has no CPU fallback" % (st, n.value))
        h = _P()
        check(self.lib.v21_ctx_create(device, C.byref(h)))
        self.h, self.device = h, device
        self.lock = threading.Lock()

    @classmethod
    def default(cls, device=None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("V21_DEVICE_FROM_RANK") else 0
        if device not in cls._default:
            cls._default[device] = cls(device)
        return cls._default[device]

    def sync(self):
        check(self.lib.v21_ctx_sync(self.h))

    def set_stream(self, hip_stream):
        check(self.lib.v21_ctx_set_stream(self.h, _P(hip_stream)))

    def poison_lds(self, pattern=0xFFFFFFFF):
        """Diagnostics: fill every CU's LDS with ``pattern`` (NaN by default) before the next launch."""
        check(self.lib.v21_debug_poison_lds(self.h, pattern))

    def clock_probe_start(self, duration_ms, period_us=50.0):
        """Diagnostics: sample the shader clock for `duration_ms` beside the kernels launched meanwhile (include/v21.h)."""
        check(self.lib.v21_debug_clock_probe_start(self.h, float(duration_ms), float(period_us)))

    def clock_probe_read(self):
        """-> {"ghz_mean", "ghz_min", "ghz_max", "samples"} of the probe started last (waits for it)."""
        a, b, c_, n = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
        check(self.lib.v21_debug_clock_probe_read(self.h, C.byref(a), C.byref(b), C.byref(c_), C.byref(n)))
        return {"ghz_mean": a.value, "ghz_min": b.value, "ghz_max": c_.value, "samples": n.value}

    def memset(self, dptr, byte, nbytes):
        check(self.lib.v21_memset(self.h, _P(dptr), int(byte), int(nbytes)))

    def malloc(self, nbytes):
        p = _P()
        check(self.lib.v21_malloc(self.h, nbytes, C.byref(p)))
        return p.value

    # ---- page-locked result buffers -----------------------------------------------------------
    # A large predict() result lands in host memory over PCIe; into an ordinary numpy array the
    # runtime goes through an internal staging buffer (~10 GB/s), into page-locked memory it goes
    # directly.  Pinning is expensive, so buffers are pooled: an array handed out owns its buffer
    # until it is garbage-collected, then the buffer returns to the pool (at most PIN_POOL_MAX
    # buffers are alive; beyond that, and for small results, plain numpy arrays are used).
    PIN_MIN_BYTES = 8 << 20
    PIN_POOL_MAX = 4

    def pinned_empty(self, shape, dtype=np.float32):
        """A numpy array in page-locked memory, or None (pool exhausted / small / unavailable)."""
        import weakref
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if nbytes < self.PIN_MIN_BYTES:
            return None
        import threading
        pool = self.__dict__.setdefault("_pin_pool", {"free": [], "live": 0, "lock": threading.Lock()})
        ptr = None
        with pool["lock"]:  # finalizers of released arrays may run on any thread
            for i, (p, cap) in enumerate(pool["free"]):
                if cap >= nbytes:
                    ptr, cap_ = pool["free"].pop(i)
                    break
            if ptr is None:
                if pool["live"] + len(pool["free"]) >= self.PIN_POOL_MAX:
                    if pool["free"]:  # replace the smallest idle buffer by one that fits
                        pool["free"].sort(key=lambda t: t[1])
                        q, _ = pool["free"].pop(0)
                        self.lib.v21_host_free(self.h, _P(q))
                    else:
                        return None
                hp = _P()
                if self.lib.v21_host_alloc(self.h, nbytes, C.byref(hp)) != 0:
                    return None
                ptr, cap_ = hp.value, nbytes
            pool["live"] += 1
        raw = (C.c_char * nbytes).from_address(ptr)
        arr = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

        def _back(pool=pool, ptr=ptr, cap=cap_):
            with pool["lock"]:
                pool["live"] -= 1
                pool["free"].append((ptr, cap))
        weakref.finalize(raw, _back)  # `raw` lives exactly as long as any view of the array
        return arr

    def free(self, dptr):
        check(self.lib.v21_free(self.h, _P(dptr)))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        check(self.lib.v21_memcpy_h2d(self.h, _P(dptr), arr.ctypes.data_as(_P), arr.nbytes))

    def d2h(self, arr, dptr):
        assert arr.flags.c_contiguous
        check(self.lib.v21_memcpy_d2h(self.h, arr.ctypes.data_as(_P), _P(dptr), arr.nbytes))

    def event(self):
        e = _P()
        check(self.lib.v21_event_create(self.h, C.byref(e)))
        return e

    def record(self, ev):
        check(self.lib.v21_event_record(self.h, ev))

    def elapsed_ms(self, a, b):
        ms = C.c_float(0)
        check(self.lib.v21_event_elapsed_ms(self.h, a, b, C.byref(ms)))
        return ms.value

    # data-parallel communicator (RCCL inside the library)
    def comm_unique_id(self):
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        check(self.lib.v21_comm_get_unique_id(self.h, buf))
        return bytes(buf)

    nranks, rank = 1, 0

    def comm_init(self, nranks, rank, uid):
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(uid)
        check(self.lib.v21_comm_init(self.h, nranks, rank, buf))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_init_host(self, nranks, rank, allreduce, reduce_scatter, allgather):
        """Collectives supplied by the host: three callables ``f(array_view, n_or_n_per) -> None`` that work IN
        PLACE on a float32 numpy view of the library's page-locked staging buffer (see include/v21.h)."""
        def wrap(fn, per_rank):
            def cb(_user, ptr, n):
                try:
                    total = n * nranks if per_rank else n
                    fn(np.ctypeslib.as_array(ptr, shape=(total,)), int(n))
                    return 0
                except Exception:  # an exception must not unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            return _HOST_AR(cb)
        self._host_ops = CommHostOps(None, wrap(allreduce, False), wrap(reduce_scatter, True), wrap(allgather, True))
        check(self.lib.v21_comm_init_host(self.h, nranks, rank, C.byref(self._host_ops)))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_set_sharded(self, on=True):
        check(self.lib.v21_comm_set_sharded(self.h, 1 if on else 0))

    def comm_init_null(self, nranks, rank=0):
        """A communicator without a transport (include/v21.h: v21_comm_init_null): the N > 1 step structure on one GPU,
        nothing exchanged -- for timing the compute side of a data-parallel step, never for training."""
        check(self.lib.v21_comm_init_null(self.h, int(nranks), int(rank)))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_set_buckets(self, buckets):
        """1: one all-reduce per step (default); 2: two, the output-side half overlapping the second weight-gradient launch."""
        check(self.lib.v21_comm_set_buckets(self.h, int(buckets)))

    def comm_info(self):
        """(nranks, rank, transport) as the attached communicator reports them; transport "none" / "rccl" / "host"."""
        n, r, t = C.c_int(0), C.c_int(0), C.c_int(0)
        check(self.lib.v21_comm_info(self.h, C.byref(n), C.byref(r), C.byref(t)))
        return n.value, r.value, ("none", "rccl", "host", "null")[t.value]

    def ranks_seen(self):
        """Sum of 1.0 over the communicator (one all-reduce through the library's transport): how many ranks really
        take part in an exchange.  A collective call: every rank must make it."""
        d = self.malloc(4)
        try:
            self.h2d(d, np.ones(1, np.float32))
            self.allreduce(d, 1)
            out = np.zeros(1, np.float32)
            self.d2h(out, d)
        finally:
            self.free(d)
        return int(round(float(out[0])))

    def comm_destroy(self):
        check(self.lib.v21_comm_destroy(self.h))
        self.nranks, self.rank = 1, 0

    def reduce_scatter(self, dptr, n_per):
        check(self.lib.v21_comm_reduce_scatter_f32(self.h, _P(dptr), n_per))

    def allgather(self, dptr, n_per):
        check(self.lib.v21_comm_allgather_f32(self.h, _P(dptr), n_per))

    def allreduce(self, dptr, n):
        check(self.lib.v21_comm_allreduce_f32(self.h, _P(dptr), n))


class Stack(_Owned):
    """A dense stack (v21_mlp): dims[0] -> ... -> dims[-1], per-layer activation."""

    def __init__(self, ctx, dims, act):
        self.ctx, self.lib = ctx, ctx.lib
        self.dims, self.act = [int(d) for d in dims], [int(a) for a in act]
        assert len(self.act) == len(self.dims) - 1
        L = len(self.act)
        h = _P()
        check(self.lib.v21_mlp_create(ctx.h, L, (C.c_int * (L + 1))(*self.dims), (C.c_int * L)(*self.act), C.byref(h)))
        self.h = h
        self._own(self.lib.v21_mlp_destroy)
        n = C.c_size_t(0)
        check(self.lib.v21_mlp_num_params(h, C.byref(n)))
        self.num_params = n.value
        self._mean_keep = None

    def set_weights(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32).ravel()
        check(self.lib.v21_mlp_set_weights(self.h, _fptr(flat), flat.size))

    def get_weights(self):
        out = np.empty(self.num_params, np.float32)
        check(self.lib.v21_mlp_get_weights(self.h, _fptr(out), out.size))
        return out

    def set_input_transform(self, log_mask, zero_floor, lo, hi):
        if log_mask is None:
            check(self.lib.v21_mlp_set_input_transform(self.h, None))
            return
        t = AffineIn()
        n = len(lo)
        t.n = n
        for j in range(n):
            t.log_mask[j] = int(bool(log_mask[j]))
            t.zero_floor[j] = float(zero_floor[j])
            t.lo[j] = float(lo[j])
            t.span[j] = float(hi[j] - lo[j])  # float64, formed as the reference forms `maximum - minimum` (preprocess.py:106)
        check(self.lib.v21_mlp_set_input_transform(self.h, C.byref(t)))

    def set_output_transform(self, std, mean):
        if mean is None:
            check(self.lib.v21_mlp_set_output_transform(self.h, None))
            return
        mean = np.ascontiguousarray(mean, dtype=np.float32)
        t = AffineOut(float(std), _fptr(mean), mean.size)
        check(self.lib.v21_mlp_set_output_transform(self.h, C.byref(t)))

    def has_fused(self, precision="f32"):
        y = C.c_int(0)
        check(self.lib.v21_mlp_has_fused(self.h, precision_id(precision), C.byref(y)))
        return bool(y.value)

    def last_route(self):
        """(route name of the last device forward call, {route name: calls since creation}) -- written where the kernels
        are launched (include/v21.h: v21_mlp_last_route)."""
        r = C.c_int(0)
        cnt = (C.c_longlong * 8)()
        check(self.lib.v21_mlp_last_route(self.h, C.byref(r), cnt))
        return FWD_ROUTES.get(r.value, "none"), {FWD_ROUTES[i]: int(cnt[i]) for i in FWD_ROUTES if cnt[i]}

    def jit(self, precision="f32", wait_ms=-1):
        """Ask for the fused kernel of THIS stack (run-time instantiation, include/v21.h: v21_mlp_jit) and wait up to
        `wait_ms` (< 0: until compiled).  -> "ready" / "compiling"; raises EngineError when the stack cannot have one."""
        s = C.c_int(0)
        check(self.lib.v21_mlp_jit(self.h, precision_id(precision), int(wait_ms), C.byref(s)))
        return "ready" if s.value == 1 else "compiling"

    def forward(self, x, precision="f32", flags=0):
        """host (n, in) float32/float64 -> host (n, out) float32"""
        x = np.asarray(x)
        if x.dtype == np.float64:
            dt = 1
        else:
            x = x.astype(np.float32, copy=False)
            dt = 0
        x = np.ascontiguousarray(x)
        if x.ndim != 2 or x.shape[1] != self.dims[0]:
            raise ValueError("expected input of shape (n, %d), got %r" % (self.dims[0], x.shape))
        y = self.ctx.pinned_empty((x.shape[0], self.dims[-1]))
        if y is None:
            y = np.empty((x.shape[0], self.dims[-1]), np.float32)
        with self.ctx.lock:
            check(self.lib.v21_mlp_forward(self.h, x.ctypes.data_as(_P), dt, x.shape[0], _fptr(y),
                                           precision_id(precision), flags))
        return y

    def forward_clocked(self, d_x, ldx, n, d_y, ldy, d_stamps, precision="f16", flags=0):
        """forward_dev through the clock-stamped instantiation of the headline stack's kernel (include/v21.h:
        v21_debug_forward_clocked); d_stamps: device buffer of 5 uint64 per 128-row workgroup."""
        check(self.lib.v21_debug_forward_clocked(self.h, _P(d_x), ldx, n, _P(d_y), ldy, precision_id(precision), flags, _P(d_stamps)))

    def forward_dev(self, d_x, ldx, n, d_y, ldy, precision="f32", flags=0):
        check(self.lib.v21_mlp_forward_dev(self.h, _P(d_x), ldx, n, _P(d_y), ldy, precision_id(precision), flags))


def jit_prebuild_train(dims, act, precision, directory=None):
    """jit_prebuild for the fused TRAINING kernel of (dims, act, f16 | bf16) (include/v21.h: v21_trainer_jit)."""
    lib = load_library()
    L = len(act)
    check(lib.v21_jit_prebuild(L, (C.c_int * (L + 1))(*[int(d) for d in dims]), (C.c_int * L)(*[int(a) for a in act]),
                               precision_id(precision) | 16, directory.encode() if directory else None))


def jit_prebuild(dims, act, precision, directory=None):
    """Compile the fused kernel of (dims, act, precision) into `directory` (None: kernel_cache/ next to libv21.so).
    Needs hiprtc but no GPU.  Call it from a process that has not loaded ANOTHER LLVM (importing torch does: hiprtc then
    finds that copy's option table, which lacks the AMDGPU flags, and LLVM ends the process) -- a build step, as
    __graft_entry__.build() uses it; at run time the library compiles in a child process of its own (csrc/jitc_main.cpp)."""
    lib = load_library()
    L = len(act)
    check(lib.v21_jit_prebuild(L, (C.c_int * (L + 1))(*[int(d) for d in dims]), (C.c_int * L)(*[int(a) for a in act]),
                               precision_id(precision), directory.encode() if directory else None))


class Trainer(_Owned):
    """Adam trainer bound to a Stack (v21_trainer)."""

    def __init__(self, stack, precision="f32", max_batch=256):
        self.stack, self.lib, self.ctx = stack, stack.lib, stack.ctx
        h = _P()
        check(self.lib.v21_trainer_create(stack.h, precision_id(precision), int(max_batch), C.byref(h)))
        self.h = h
        self._own(self.lib.v21_trainer_destroy, [stack])
        self.max_batch = int(max_batch)

    def set_adam(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        cfg = Adam(lr, beta1, beta2, eps)
        check(self.lib.v21_trainer_set_adam(self.h, C.byref(cfg)))

    def set_lr(self, lr):
        check(self.lib.v21_trainer_set_lr(self.h, float(lr)))

    def get_lr(self):
        v = C.c_float(0)
        check(self.lib.v21_trainer_get_lr(self.h, C.byref(v)))
        return v.value

    def set_data(self, which, x, y, row_weight):
        x = np.ascontiguousarray(x, dtype=np.float32)
        rw = np.ascontiguousarray(row_weight, dtype=np.float32)
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float32)
            assert y.shape[0] == x.shape[0]
        assert rw.shape == (x.shape[0],)
        # A second train() on the same arrays (the reference's recipe: emulator.py:739-764 re-feeds its numpy arrays
        # on every call) finds its split resident: (shape, 128-bit hash of every byte) of the three buffers decides.
        # One pass of the hash is ~2 ms for the reference's 44 MB against ~6.5 ms of pageable upload.
        from .preprocess import _digest
        key = tuple((a.shape, _digest(a.reshape(-1).view(np.uint8).data)) if a is not None else None for a in (x, y, rw))
        if getattr(self, "_resident", None) is None:
            self._resident = {}
        if self._resident.get(which) == key:
            return
        self._resident.pop(which, None)
        check(self.lib.v21_trainer_set_data(self.h, which, _fptr(x), _fptr(y) if y is not None else None,
                                            _fptr(rw), x.shape[0]))
        self._resident[which] = key
        if which == 0:
            self.n_train = int(x.shape[0])

    def run_epoch(self, perm, batch):
        loss = C.c_double(0)
        pp = None
        if perm is not None:
            perm = _row_table(perm, getattr(self, "n_train", None), "Trainer.run_epoch")
            pp = perm.ctypes.data_as(C.POINTER(C.c_int32))
        with self.ctx.lock:
            check(self.lib.v21_trainer_run_epoch(self.h, pp, int(batch), C.byref(loss)))
        return loss.value

    def evaluate(self, which, batch):
        loss = C.c_double(0)
        with self.ctx.lock:
            check(self.lib.v21_trainer_eval(self.h, which, int(batch), C.byref(loss)))
        return loss.value

    def data_dev(self, which=0):
        """-> (d_x, d_y, d_rw, n): device pointers of the resident split (d_y == d_x when it was set with y=None), for custom
        loops that step on slices of it with step_dev (include/v21.h: v21_trainer_get_data_dev)."""
        x, y, rw, n = _P(), _P(), _P(), C.c_int64(0)
        check(self.lib.v21_trainer_get_data_dev(self.h, int(which), C.byref(x), C.byref(y), C.byref(rw), C.byref(n)))
        return x.value, y.value, rw.value, n.value

    def step_dev(self, d_x, d_y, d_rw, n_rows, global_rows=None):
        check(self.lib.v21_trainer_step_dev(self.h, _P(d_x), _P(d_y) if d_y else None, _P(d_rw), int(n_rows),
                                            int(global_rows if global_rows is not None else n_rows)))

    def last_step_loss(self):
        v = C.c_double(0)
        check(self.lib.v21_trainer_last_step_loss(self.h, C.byref(v)))
        return v.value

    def get_state(self):
        n = self.stack.num_params
        m, v, it = np.empty(n, np.float32), np.empty(n, np.float32), C.c_int64(0)
        check(self.lib.v21_trainer_get_state(self.h, C.byref(it), _fptr(m), _fptr(v), n))
        return it.value, m, v

    def set_state(self, it, m=None, v=None):
        n = self.stack.num_params
        m = None if m is None else np.ascontiguousarray(m, np.float32)
        v = None if v is None else np.ascontiguousarray(v, np.float32)
        check(self.lib.v21_trainer_set_state(self.h, int(it), _fptr(m) if m is not None else None,
                                             _fptr(v) if v is not None else None, n))

    def set_vae(self, kl_weight, sample=True, seed=0):
        """Variational mode of a stack with an ACT_GAUSS layer (include/v21.h: v21_trainer_set_vae)."""
        check(self.lib.v21_trainer_set_vae(self.h, float(kl_weight), 1 if sample else 0, int(seed) & (2**64 - 1)))

    def use_graph(self, enable=True):
        """Captured-step replay (hipGraph), opt-in: one rank, no variational layer.  Same kernels, same
        arithmetic, bit-identical results; the host enqueues one graph launch per step instead of 3-14 kernels
        (measured r2: the steps are GPU-bound, so this frees the host thread but does not shorten a step)."""
        check(self.lib.v21_trainer_use_graph(self.h, 1 if enable else 0))

    def check_chain_jobs(self, fw_bytes=-1, bw_bytes=-1):
        """Diagnostics: validate the small-batch f32 chain's job table against packed streams of the given sizes
        (bytes; -1 = the allocated ones).  Raises EngineError where a row points outside."""
        check(self.lib.v21_debug_check_chain_jobs(self.h, int(fw_bytes), int(bw_bytes)))

    def route_counters(self):
        """Diagnostics: eager 16-bit steps so far by route: dict(chain=, fused=, stream_packs=, stream_adam=) -- steps through
        the 32-row chain, through the fused training kernel, fused steps that launched the stream pack first, Adam passes
        that wrote the fused kernel's stream (include/v21.h: v21_debug_trainer_counters)."""
        out = (C.c_longlong * 4)()
        check(self.lib.v21_debug_trainer_counters(self.h, out))
        return dict(zip(("chain", "fused", "stream_packs", "stream_adam"), (int(v) for v in out)))

    def jit(self, wait_ms=-1):
        """The fused training kernel of THIS trainer's stack (include/v21.h: v21_trainer_jit): wait up to `wait_ms` (< 0: until
        compiled) for its run-time instantiation.  -> "ready" / "compiling"; raises EngineError when the trainer cannot have one."""
        s = C.c_int(0)
        check(self.lib.v21_trainer_jit(self.h, int(wait_ms), C.byref(s)))
        return "ready" if s.value == 1 else "compiling"

    def phase_timing(self, steps, cut=4):
        """Stamp the next `steps` eager steps with two HIP events: the step's start and cut point `cut` (include/v21.h:
        v21_trainer_phase_timing); steps = 0: off."""
        check(self.lib.v21_trainer_phase_timing(self.h, int(steps), int(cut)))

    def phase_times(self):
        """-> (median microseconds from a step's start to the cut point, stamped steps) since the last call."""
        ms, n = C.c_double(0.0), C.c_int(0)
        check(self.lib.v21_trainer_phase_times(self.h, C.byref(ms), C.byref(n)))
        return 1e3 * ms.value, n.value

    def phase_profile(self, step, steps=50):
        """Where a step's time goes: `step()` (a callable that takes ONE optimizer step on this trainer) is run `steps`
        times per cut point with two HIP events per step, and the phases are the differences of the four cumulative times
        (the marker's own cost -- an event is a packet in the stream, ~5 us when nothing separates two -- cancels in them;
        the first phase still carries one marker: `marker_us` = the stamped whole step minus `unstamped_step_us` when the
        caller supplies the latter says how much that is).  -> dict of microseconds."""
        cum = []
        for cut in (1, 2, 3, 4):
            self.phase_timing(steps, cut)
            for _ in range(steps):
                step()
            us, n = self.phase_times()
            cum.append(us if n else float("nan"))
        self.phase_timing(0)
        return {"forward_and_activation_gradients_us": cum[0], "weight_gradients_us": cum[1] - cum[0], "exchange_exposed_us": cum[2] - cum[1],
                "adam_and_repack_us": cum[3] - cum[2], "stamped_step_us": cum[3], "steps_per_cut": steps,
                "note": "differences of cumulative start-to-cut times of four separate stamped runs (two HIP events per step); the first "
                        "phase and stamped_step_us carry one marker's cost (stamped_step_us minus the leg's unstamped ms_per_step); "
                        "exchange_exposed = the part of the gradient exchange no weight-gradient launch covers; single-rank steps whose "
                        "gradients and Adam are ONE launch report it under adam_and_repack"}

    def last_route(self):
        """((forward route, update route) of the last eager step, {(fwd or upd) route name: steps since creation}) -- written
        where the kernels are launched (include/v21.h: v21_trainer_last_route)."""
        f, u = C.c_int(0), C.c_int(0)
        fc, uc = (C.c_longlong * 8)(), (C.c_longlong * 8)()
        check(self.lib.v21_trainer_last_route(self.h, C.byref(f), C.byref(u), fc, uc))
        counts = {TRAIN_FWD_ROUTES[i]: int(fc[i]) for i in TRAIN_FWD_ROUTES if fc[i]}
        counts.update({"upd:" + TRAIN_UPD_ROUTES[i]: int(uc[i]) for i in TRAIN_UPD_ROUTES if uc[i]})
        return (TRAIN_FWD_ROUTES.get(f.value, "none"), TRAIN_UPD_ROUTES.get(u.value, "none")), counts

    def enable_stamps(self, on=True):
        """Cycle stamps of the chain kernel's phases (diagnostics; off by default: they cost 2-3 us per step)."""
        check(self.lib.v21_trainer_enable_stamps(self.h, 1 if on else 0))

    def chain_stamps(self, n=40):
        out = (C.c_uint64 * n)()
        check(self.lib.v21_trainer_chain_stamps(self.h, out, n))
        return np.array(out[:], dtype=np.uint64)

    def get_grad(self):
        g = np.empty(self.stack.num_params, np.float32)
        check(self.lib.v21_trainer_get_grad(self.h, _fptr(g), g.size))
        return g


class Joint(_Owned):
    """Autoencoder + latent emulator stepping together on the same rows (v21_joint_*; BASELINE configs[2])."""

    def __init__(self, ae_trainer, em_trainer, latent_layer):
        self.lib = ae_trainer.lib
        self.trainers = (ae_trainer, em_trainer)  # keep them alive
        h = _P()
        check(self.lib.v21_joint_create(ae_trainer.h, em_trainer.h, int(latent_layer), C.byref(h)))
        self.h = h
        self._own(self.lib.v21_joint_destroy, self.trainers)

    def run_epoch(self, perm, batch):
        """-> (autoencoder epoch loss, emulator epoch loss)"""
        out = (C.c_double * 2)()
        pp = None
        if perm is not None:
            perm = _row_table(perm, getattr(self.trainers[0], "n_train", None), "Joint.run_epoch")
            pp = perm.ctypes.data_as(C.POINTER(C.c_int32))
        check(self.lib.v21_joint_run_epoch(self.h, pp, int(batch), out))
        return float(out[0]), float(out[1])

    def evaluate(self):
        """-> (autoencoder validation loss, emulator validation loss against the current encoder's latents)"""
        out = (C.c_double * 2)()
        check(self.lib.v21_joint_eval(self.h, out))
        return float(out[0]), float(out[1])


class Sweep(_Owned):
    """Several Trainers stepped in lock step on one shared batch stream (v21_sweep);
    trainer 0 holds the training set."""

    def __init__(self, trainers):
        self.trainers = list(trainers)
        self.lib, self.ctx = self.trainers[0].lib, self.trainers[0].ctx
        arr = (_P * len(self.trainers))(*[t.h.value for t in self.trainers])
        h = _P()
        check(self.lib.v21_sweep_create(arr, len(self.trainers), C.byref(h)))
        self.h = h
        self._own(self.lib.v21_sweep_destroy, self.trainers)

    def run_epoch(self, perm, batch):
        losses = (C.c_double * len(self.trainers))()
        pp = None
        if perm is not None:
            perm = _row_table(perm, getattr(self.trainers[0], "n_train", None), "Sweep.run_epoch")
            pp = perm.ctypes.data_as(C.POINTER(C.c_int32))
        with self.ctx.lock:
            check(self.lib.v21_sweep_run_epoch(self.h, pp, int(batch), losses))
        return [float(v) for v in losses]
