"""Probe: is the headline kernel clock/power-bound?  Same launches on random vs all-zero operands (zero data toggles
fewer MFMA / LDS / register bits: a large difference means the shader clock, not the schedule, sets the time)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
native = importlib.import_module("21cmvae_amd._native")
synth = importlib.import_module("21cmvae_amd.synth")
B = 65536
ctx = native.Context(0)
st = native.Stack(ctx, bench.DIMS, bench.ACT)
w = bench.glorot(bench.DIMS, seed=3)
params = synth.make_params(B, seed=1000, dtype=np.float32)
dx, dy = ctx.malloc(params.nbytes), ctx.malloc(B * 451 * 4)
def run(prec, n=300):
    for _ in range(100):
        st.forward_dev(dx, 7, B, dy, 451, prec, 0)
    ctx.sync()
    a, b = ctx.event(), ctx.event()
    ctx.record(a)
    for _ in range(n):
        st.forward_dev(dx, 7, B, dy, 451, prec, 0)
    ctx.record(b); ctx.sync()
    return ctx.elapsed_ms(a, b) / n * 1e3
for prec in ("f16", "bf16"):
    for name, ww, xx in (("random", w, params), ("zero weights", np.zeros_like(w), params), ("random", w, params),
                         ("weights 1/64 const", np.full_like(w, 1 / 64), np.ones_like(params))):
        st.set_weights(ww); ctx.h2d(dx, xx)
        print("%-5s %-20s %.2f us/launch" % (prec, name, run(prec)), flush=True)
