"""Diagnostics (r4): the run-time-instantiated fused kernel against the float64 oracle, per stack / precision / row count."""
import os
import importlib, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
from oracle import ref_numpy as ora
ctx = native.Context.default()
for dims in ([7, 64, 128, 451], [451, 100, 451], [7, 32, 128, 256, 451], [7, 288, 352, 288, 224, 9], [451, 352, 9], [7, 352, 352, 352, 224, 451]):
    act = [1] * (len(dims) - 2) + [0]
    Ws, bs = ora.init_mlp(dims, seed=11)
    rng = np.random.default_rng(12)
    bs = [rng.normal(scale=0.1, size=b.shape).astype(np.float32) for b in bs]
    st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
    for prec in ("f32", "f16"):
        st.jit(prec)
        for n in (77, 5000):
            x = rng.normal(size=(n, dims[0])).astype(np.float32)
            ref = ora.mlp_forward(Ws, bs, x, dtype=np.float64)
            y = st.forward(x, prec, flags=native.FWD_FORCE_JIT)
            yc = st.forward(x, prec, flags=native.FWD_FORCE_CHAIN)
            d = np.abs(y - ref)
            bad = d > (1e-4 if prec == "f32" else 3e-2) * max(1, np.abs(ref).max())
            print(dims, prec, n, "jit max err %.3g  chain max err %.3g  bad %d" % (d.max(), np.abs(yc - ref).max(), bad.sum()),
                  "bad cols", np.flatnonzero(bad.any(0))[:12], "bad rows", np.flatnonzero(bad.any(1))[:8], flush=True)
