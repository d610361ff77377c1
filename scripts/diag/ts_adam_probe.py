"""Diagnostics: where does the gradient of a fused step fed by the Adam-written stream differ from one fed by the packed stream?"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("V21_FUSED_TRAIN_ROWS", "1")
native = importlib.import_module("21cmvae_amd._native"); synth = importlib.import_module("21cmvae_amd.synth")
from oracle import ref_numpy as ora
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
ctx = native.Context(0)
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
n = int(os.environ.get("PROBE_ROWS", "777"))
rng = np.random.default_rng(5)
Ws, bs = ora.init_mlp(dims, seed=77)
bs = [rng.normal(scale=0.05, size=b.shape).astype(np.float32) for b in bs]
flat = ora.flatten_params(Ws, bs)
sig = synth.make_signals(n, seed=11); x = ora.preproc(sig, sig)
w = ora.relative_mse_row_weight(x, sig).astype(np.float32)
st = native.Stack(ctx, dims, act); st.set_weights(flat)
tr = native.Trainer(st, prec, n); tr.set_adam(lr=1e-3); tr.set_data(0, x, None, w)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3): tr.run_epoch(None, n)
w3 = st.get_weights()
l4 = tr.run_epoch(None, n); g4 = tr.get_grad()
st2 = native.Stack(ctx, dims, act); st2.set_weights(w3)
tr2 = native.Trainer(st2, prec, n); tr2.set_adam(lr=1e-3); tr2.set_data(0, x, None, w)
l4p = tr2.run_epoch(None, n); g4p = tr2.get_grad()
print("loss", l4, l4p, tr.route_counters(), tr2.route_counters())
off = 0
for l in range(len(act)):
    K, N = dims[l], dims[l + 1]
    dW = np.abs(g4[off:off + K * N] - g4p[off:off + K * N]).reshape(K, N); off += K * N
    db = np.abs(g4[off:off + N] - g4p[off:off + N]); off += N
    bad = np.argwhere(dW > 0)
    print(f"layer {l}: dW differs in {len(bad)} of {K*N} (max {dW.max():.3e}); db differs in {(db>0).sum()} (max {db.max():.3e})")
    if len(bad):
        print("   rows (k) with differences:", np.unique(bad[:, 0])[:40], " cols (n):", np.unique(bad[:, 1])[:40])
