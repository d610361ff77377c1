"""A/B of fused-forward variants in ONE process, interleaved rounds (methodology: compare
variants on the same device, same clocks).  Usage: python scripts/ab_fused.py f16 "X2=0" "X2=1" "X2=1,PRIO=1" ..."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module("21cmvae_amd._native")
prec = sys.argv[1]
variants = sys.argv[2:]
DIMS = [7, 352, 352, 352, 224, 451]
ctx = native.Context(0)
st = native.Stack(ctx, DIMS, [1, 1, 1, 1, 0])
rng = np.random.default_rng(0)
st.set_weights(rng.normal(scale=0.05, size=st.num_params).astype(np.float32))
B = 65536
x = rng.uniform(-1, 1, size=(B, 7)).astype(np.float32)
d_x, d_y = ctx.malloc(x.nbytes), ctx.malloc(B * 451 * 4)
ctx.h2d(d_x, x)
KEYS = ("X2", "PRIO", "DELAY", "PIN", "S16", "W8", "SP")
def setv(v):
    for k in KEYS:
        os.environ.pop("V21_FUSED_" + k, None)
    for kv in v.split(","):
        if kv:
            k, val = kv.split("="); os.environ["V21_FUSED_" + k] = val
res = {v: [] for v in variants}
for rnd in range(12):
    for v in variants:
        setv(v)
        for _ in range(3):
            st.forward_dev(d_x, 7, B, d_y, 451, prec, 0)
        ctx.sync()
        a, b = ctx.event(), ctx.event()
        ctx.record(a)
        for _ in range(20):
            st.forward_dev(d_x, 7, B, d_y, 451, prec, 0)
        ctx.record(b); ctx.sync()
        res[v].append(ctx.elapsed_ms(a, b) / 20 * 1e3)
for v in variants:
    r = np.array(res[v][2:])
    print("%-28s median %.2f us  min %.2f  max %.2f" % (v or "(default)", np.median(r), r.min(), r.max()))
