import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module("21cmvae_amd._native")
import bench
ctx = native.Context.default()
st = native.Stack(ctx, bench.DIMS, bench.ACT); st.set_weights(bench.glorot(bench.DIMS, seed=3))
for n in (1, 32, 256):
    x = np.random.default_rng(0).normal(size=(n, 7)).astype(np.float32)
    for prec in ("f32", "f16"):
        for flags, name in ((0, "fused"), (native.FWD_FORCE_GENERIC, "generic")):
            for _ in range(5): st.forward(x, prec, flags)
            t0 = time.perf_counter()
            for _ in range(300): y = st.forward(x, prec, flags)
            print("n=%4d %s %-8s %7.1f us/call" % (n, prec, name, (time.perf_counter() - t0) / 300 * 1e6))
