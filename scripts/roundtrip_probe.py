import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module("21cmvae_amd._native")
import bench
ctx = native.Context.default()
st = native.Stack(ctx, bench.DIMS, bench.ACT); st.set_weights(bench.glorot(bench.DIMS, seed=3))
x = np.random.default_rng(0).normal(size=(65536, 7)).astype(np.float32)
for prec in ("f16", "f32", "f16", "f32"):
    for _ in range(4): y = st.forward(x, prec)
    t0 = time.perf_counter()
    for _ in range(10): y = st.forward(x, prec)
    print(prec, "%.2f ms/call" % ((time.perf_counter() - t0) / 10 * 1e3), ctx._pin_pool["live"], len(ctx._pin_pool["free"]))
