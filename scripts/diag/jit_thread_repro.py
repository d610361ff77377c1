"""Diagnostics (r4): a run-time compilation (hiprtc thread, csrc/jit.hip) running beside kernel launches of the main thread."""
import os
import importlib, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
from oracle import ref_numpy as ora
ctx = native.Context.default()
dims, act = [451, 72, 9], [1, 0]      # (not in kernel_cache/: compiled now)
Ws, bs = ora.init_mlp(dims, seed=1)
st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
x = np.random.default_rng(0).normal(size=(9000, 451)).astype(np.float32)
y = st.forward(x, "f16")             # asks for the kernel, takes the chain route meanwhile
print("first forward done", flush=True)
tdims, tact = [451, 64, 9, 32, 451], [1, 0, 1, 0]
Wt, bt = ora.init_mlp(tdims, seed=2)
stt = native.Stack(ctx, tdims, tact); stt.set_weights(ora.flatten_params(Wt, bt))
tr = native.Trainer(stt, "f16", 256); tr.set_adam(lr=1e-3)
sig = (np.random.default_rng(1).normal(size=(4096, 451)) * 20 - 30).astype(np.float32)
yb = ora.preproc(sig, sig); w = ora.relative_mse_row_weight(yb, sig).astype(np.float32)
tr.set_data(0, yb, None, w)
t0 = time.time()
while time.time() - t0 < 25:
    tr.run_epoch(None, 256)
    s = st.jit("f16", wait_ms=0)
    if s == "ready":
        print("kernel ready after %.1f s" % (time.time() - t0), flush=True)
        break
y2 = st.forward(x, "f16")
print("max diff chain vs jit", np.abs(y - y2).max(), flush=True)
