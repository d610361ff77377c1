"""Debug: per-layer gradient differences between the f32 chain and the per-layer f32 path."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
from oracle import ref_numpy as ora
ctx = native.Context(0)
dims, act, n = [7, 352, 352, 352, 224, 9], [1, 1, 1, 1, 0], int(sys.argv[1]) if len(sys.argv) > 1 else 257
rng = np.random.default_rng(3)
x = rng.uniform(-1, 1, size=(n, 7)).astype(np.float32)
y = rng.normal(size=(n, 9)).astype(np.float32)
w = ora.mse_row_weight(y).astype(np.float32)
res = {}
for chain in (True, False):
    os.environ["V21_TRAIN_CHAIN"] = "1" if chain else "0"
    Ws, bs = ora.init_mlp(dims, seed=31)
    st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
    tr = native.Trainer(st, "f32", max(n, 2))
    tr.set_adam(lr=1e-3); tr.set_data(0, x, y, w)
    loss = tr.run_epoch(None, max(n, 2))
    res[chain] = (loss, tr.get_grad())
print("loss", res[True][0], res[False][0])
gc, gn = res[True][1], res[False][1]
o = 0
for l, (k, nn) in enumerate(zip(dims[:-1], dims[1:])):
    for name, sz, shape in (("W", k * nn, (k, nn)), ("b", nn, (nn,))):
        a, b = gc[o:o + sz].reshape(shape), gn[o:o + sz].reshape(shape)
        d = np.abs(a - b)
        print("layer %d %s: max|diff| %.3e  scale %.3e  argmax %s" % (l, name, d.max(), np.abs(b).max(), np.unravel_index(d.argmax(), shape)))
        o += sz
cos = float(gc.astype(np.float64) @ gn.astype(np.float64) / (np.linalg.norm(gc.astype(np.float64)) * np.linalg.norm(gn.astype(np.float64))))
err = np.abs(gc - gn) / np.abs(gn).max()
print("1 - cos %.3e; err quantiles 0.99 %.2e 0.999 %.2e 0.9999 %.2e max %.2e; count > 2e-4: %d of %d" % (1 - cos, np.quantile(err, 0.99), np.quantile(err, 0.999), np.quantile(err, 0.9999), err.max(), int((err > 2e-4).sum()), err.size))
