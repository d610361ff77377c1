"""Gradient fidelity of the reduced-precision training modes vs the f32 mode, by batch size."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module("21cmvae_amd._native")
synth = importlib.import_module("21cmvae_amd.synth")
pp = importlib.import_module("21cmvae_amd.preprocess")
losses = importlib.import_module("21cmvae_amd.losses")
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
ctx = native.Context(0)
rng = np.random.default_rng(0)
w0 = None
for B in (256, 4096):
    sig = synth.make_signals(B, seed=5)
    y = pp.preproc(sig, sig)
    rw = losses.relative_mse_loss(sig)._v21_row_weight(y).astype(np.float32)
    for stage in ("init", "trained"):
      if stage == "trained":  # a converged model: residuals (and gradients) ~100x smaller
        st = native.Stack(ctx, dims, act); st.set_weights(w0)
        tr = native.Trainer(st, "f32", B); tr.set_adam(lr=1e-3); tr.set_data(0, y, None, rw)
        for _ in range(600):
            last = tr.run_epoch(None, B)
        wcur = st.get_weights()
        print("   trained: loss %.3e" % last)
      else:
        import bench
        w0 = bench.glorot(dims, seed=4); wcur = w0
      g = {}
      for prec in ("f32", "f16", "bf16"):
        st = native.Stack(ctx, dims, act)
        st.set_weights(wcur)
        tr = native.Trainer(st, prec, B)
        tr.set_data(0, y, None, rw)
        tr.run_epoch(None, B)
        g[prec] = tr.get_grad().astype(np.float64)
      for prec in ("f16", "bf16"):
        a, b = g[prec], g["f32"]
        o, cs = 0, []
        for l in range(5):
            n = dims[l] * dims[l + 1] + dims[l + 1]
            cs.append(float(a[o:o + n] @ b[o:o + n] / (np.linalg.norm(a[o:o + n]) * np.linalg.norm(b[o:o + n]) + 1e-300)))
            o += n
        print("B=%d %s %s: |g|/|g32| = %.4f  cos per layer = %s" % (B, stage, prec, np.linalg.norm(a) / np.linalg.norm(b), ["%.5f" % c for c in cs]))
