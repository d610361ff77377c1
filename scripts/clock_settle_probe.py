"""Probe: how the headline kernel's time per launch evolves over a long back-to-back run (clock / power settling)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
native = importlib.import_module("21cmvae_amd._native")
synth = importlib.import_module("21cmvae_amd.synth")
B = 65536
ctx = native.Context(0)
st = native.Stack(ctx, bench.DIMS, bench.ACT)
st.set_weights(bench.glorot(bench.DIMS, seed=3))
params = synth.make_params(B, seed=1000, dtype=np.float32)
dx, dy = ctx.malloc(params.nbytes), ctx.malloc(B * 451 * 4)
ctx.h2d(dx, params)
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
out = []
t00 = time.perf_counter()
for blk in range(24):
    n = 50 if blk < 8 else 500
    a, b = ctx.event(), ctx.event()
    ctx.record(a)
    for _ in range(n):
        st.forward_dev(dx, 7, B, dy, 451, prec, 0)
    ctx.record(b); ctx.sync()
    out.append((time.perf_counter() - t00, n, ctx.elapsed_ms(a, b) / n * 1e3))
for t, n, us in out:
    print("t=%.3fs  %4d launches  %.2f us/launch" % (t, n, us))
