"""Per-wave cycle stamps of workgroup 0 of the chain kernels (diagnostic build only: make the library with
-DV21_CHAIN_FINE and point V21_LIB at it).  For every forward layer: when each of the 16 waves reached the layer, started
its contraction, finished its unit(s) and left the layer's barrier -- relative to the end of the previous layer.
    python scripts/diag/chain_wave_stamps.py [batch] [f16|bf16|f32]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
ctx = native.Context(0)
st = native.Stack(ctx, dims, act)
rng = np.random.default_rng(0)
st.set_weights(rng.normal(scale=0.05, size=st.num_params).astype(np.float32))
tr = native.Trainer(st, prec, B)
tr.enable_stamps()
x = rng.normal(size=(B, 451)).astype(np.float32); w = np.full(B, 1 / 451, np.float32)
d_x, d_w = ctx.malloc(x.nbytes), ctx.malloc(w.nbytes)
ctx.h2d(d_x, x); ctx.h2d(d_w, w)
for _ in range(20):
    tr.step_dev(d_x, None, d_w, B, B)
ctx.sync()
s = tr.chain_stamps(64 + 16 * 24).astype(np.int64)
print("layer ends:", np.diff(s[:12]).tolist())
pw = s[64:].reshape(-1, 16)
for l in range(5):
    base = s[1 + l]
    print("fwd layer", l, "(end %d)" % (s[2 + l] - base))
    for i, name in enumerate(["layer start ", "contract go ", "unit done   ", "past barrier"]):
        print("   ", name, [int(v - base) if v else -1 for v in pw[4 * l + i]])
