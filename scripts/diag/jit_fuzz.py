"""Diagnostics (r4): random stacks through the run-time-instantiated fused kernel (csrc/jit.hip) against the float64 oracle and
the table-driven kernel.  A stack the fused kernel cannot hold must be refused or rejected, never answer wrongly.
  python jit_fuzz.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
from oracle import ref_numpy as ora
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = native.Context.default()
bad = refused = ok = 0
for c in range(cases):
    L = int(rng.integers(1, 7))
    widths = [1, 2, 7, 9, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 96, 100, 128, 144, 145, 160, 200, 224, 256, 288, 300, 352, 400, 448, 451, 460]
    dims = [int(rng.choice(widths)) for _ in range(L + 1)]
    act = [int(rng.integers(0, 2)) for _ in range(L - 1)] + [0]
    prec = ["f16", "f32", "bf16"][int(rng.integers(0, 3))]
    n = int(rng.choice([1, 5, 77, 128, 129, 4097, 9000]))
    Ws, bs = ora.init_mlp(dims, seed=c)
    bs = [rng.normal(scale=0.1, size=b.shape).astype(np.float32) for b in bs]
    st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
    x = rng.normal(size=(n, dims[0])).astype(np.float32)
    h = x.astype(np.float64)
    for W_, b_, a_ in zip(Ws, bs, act):
        h = h @ W_.astype(np.float64) + b_.astype(np.float64)
        h = np.maximum(h, 0) if a_ else h
    t0 = time.time()
    try:
        st.jit(prec)
        y = st.forward(x, prec, flags=native.FWD_FORCE_JIT)
    except native.EngineError as e:
        refused += 1
        print("case %2d %-40s %-5s n=%-5d refused: %s" % (c, dims, prec, n, str(e)[:90]), flush=True)
        continue
    tol = {"f32": 3e-5, "f16": 3e-3, "bf16": 3e-2}[prec] * max(1.0, np.abs(h).max())
    err = np.abs(y - h).max()
    flag = "OK " if err <= tol and np.isfinite(y).all() else "BAD"
    if flag == "BAD":
        bad += 1
        rows = np.flatnonzero((np.abs(y - h) > tol).any(1))[:6]; cols = np.flatnonzero((np.abs(y - h) > tol).any(0))[:8]
        print("case %2d %-40s act %s %-5s n=%-5d %s err %.3g tol %.3g rows %s cols %s" % (c, dims, act, prec, n, flag, err, tol, rows, cols), flush=True)
    else:
        ok += 1
        print("case %2d %-40s %-5s n=%-5d %s err %.3g (%.1f s)" % (c, dims, prec, n, flag, err, time.time() - t0), flush=True)
print("cases %d: ok %d, refused %d, BAD %d" % (cases, ok, refused, bad))
