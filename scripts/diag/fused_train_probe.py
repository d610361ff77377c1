"""Diagnostics (r4): one optimizer step of the fused training kernel (csrc/fused_train.h) against the 32-row chain kernel and
the float64 oracle -- loss, gradient -- and the step time of both routes.   python fused_train_probe.py [rows] [prec]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
synth = importlib.import_module("21cmvae_amd.synth")
from oracle import ref_numpy as ora
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
ctx = native.Context.default()
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
Ws, bs = ora.init_mlp(dims, seed=4)
flat = ora.flatten_params(Ws, bs)
sig = synth.make_signals(rows, seed=2000)
y = ora.preproc(sig, sig)
w = ora.relative_mse_row_weight(y, sig).astype(np.float32)
d_x, d_rw = ctx.malloc(y.nbytes), ctx.malloc(w.nbytes)
ctx.h2d(d_x, y); ctx.h2d(d_rw, w)
res = {}
for name, env in (("chain", "0"), ("fused", "1")):
    os.environ["V21_FUSED_TRAIN"] = env
    os.environ["V21_FUSED_TRAIN_ROWS"] = "1"   # (the default threshold is 16,384 rows: the probe compares the routes at any size)
    st = native.Stack(ctx, dims, act); st.set_weights(flat)
    tr = native.Trainer(st, prec, rows); tr.set_adam(lr=1e-3)
    if os.environ.get("PROBE_RESIDENT", "1") == "1":   # step on the trainer's resident training set (what Model.fit does): the fused kernels gather its 16-bit copy
        tr.set_data(0, y, None, w)
        d_x, _, d_rw, _ = tr.data_dev(0)
    tr.step_dev(d_x, None, d_rw, rows, rows)
    loss = tr.last_step_loss() / rows
    g = tr.get_grad()
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < 0.3:      # (the clock needs ~50 ms of load to come up from idle)
        for _ in range(10): tr.step_dev(d_x, None, d_rw, rows, rows)
        ctx.sync()
    ctx.sync(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): tr.step_dev(d_x, None, d_rw, rows, rows)
    ctx.sync(); us = (time.perf_counter() - t0) / n * 1e6
    res[name] = (loss, g, us)
    # the shader clock beside the same steps (one sampling wave: v21_debug_clock_probe_*)
    ctx.clock_probe_start(n * us * 1e-3, 25.0)
    for _ in range(n): tr.step_dev(d_x, None, d_rw, rows, rows)
    ctx.sync(); ck = ctx.clock_probe_read()
    print("%s: loss %.6e  |g| %.4e  step %.1f us  (%.1f TFLOP/s)  clock %.2f GHz (%.2f .. %.2f)" % (name, loss, np.linalg.norm(g), us, rows * 1675840 / us / 1e6, ck["ghz_mean"], ck["ghz_min"], ck["ghz_max"]), flush=True)
if rows <= 20000:
    W = [a.astype(np.float64) for a in Ws]; b = [a.astype(np.float64) for a in bs]
    h = y.astype(np.float64); acts = [h]
    for W_, b_, a_ in zip(W, b, act):
        h = h @ W_ + b_
        h = np.maximum(h, 0) if a_ else h
        acts.append(h)
    lo, go = ora.batch_loss_and_grad(acts[-1], y.astype(np.float64), w.astype(np.float64))
    print("oracle loss %.6e" % lo)
gc, gf = res["chain"][1], res["fused"][1]
print("loss rel diff fused vs chain %.3e" % (abs(res["fused"][0] - res["chain"][0]) / res["chain"][0]))
print("gradient: cos %.6f  norm ratio %.5f  max abs diff %.3e (|g|max %.3e)" % (float(gc @ gf / np.linalg.norm(gc) / np.linalg.norm(gf)), np.linalg.norm(gf) / np.linalg.norm(gc), np.abs(gc - gf).max(), np.abs(gc).max()))
o = 0
for l, (W_, b_) in enumerate(zip(Ws, bs)):
    for nm, n_ in (("W", W_.size), ("b", b_.size)):
        a_, b2 = gc[o:o + n_], gf[o:o + n_]; o += n_
        print("  layer %d %s: cos %.6f ratio %.5f" % (l, nm, float(a_ @ b2 / (np.linalg.norm(a_) * np.linalg.norm(b2) + 1e-30)), np.linalg.norm(b2) / (np.linalg.norm(a_) + 1e-30)))
