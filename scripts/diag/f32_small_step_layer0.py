"""Layer-0 gradient of ONE f32 step against the float64 oracle, by row count and rows per workgroup of the small-batch chain
(diagnosis of sweep_fuzz seed 204 case 9, r5).   python f32_small_step_layer0.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_numpy as ora
from helpers import oracle_step
native = importlib.import_module("21cmvae_amd._native")
ctx = native.Context.default()
dims, act = [451, 288, 256, 128, 256, 451], [1, 0, 0, 0, 0]
if len(sys.argv) > 1:
    dims = [int(v) for v in sys.argv[1].split(",")]; act = [1] * (len(dims) - 2) + [0]
offs = np.cumsum([0] + [a * b + b for a, b in zip(dims[:-1], dims[1:])])
rng = np.random.default_rng(5)
for rows in (255, 256, 257, 258, 259, 260, 261, 300, 513, 1025, 1029):
    x = rng.uniform(-1, 1, size=(rows, dims[0])).astype(np.float32)
    y = None if dims[0] == dims[-1] else rng.normal(size=(rows, dims[-1])).astype(np.float32)
    w = (rng.uniform(0.5, 1.5, size=rows) / dims[-1]).astype(np.float32)
    Ws, bs = ora.init_mlp(dims, seed=3)
    lo, go = oracle_step(Ws, bs, act, x, x if y is None else y, w)
    for rpw in ("4", "8"):
        os.environ["V21_C32S_ROWS"] = rpw
        st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
        tr = native.Trainer(st, "f32", rows); tr.set_adam(lr=1e-3)
        tr.set_data(0, x, y, w)
        l = tr.run_epoch(None, rows)
        g = tr.get_grad()
        per = ["%.1e" % float(np.abs(g[offs[i]:offs[i + 1]] - go[offs[i]:offs[i + 1]]).max()) for i in range(len(dims) - 1)]
        print("rows %5d rows/wg %s route %s: loss rel %.1e  max |g - g64| per layer %s  (largest |g64| %.2e)" % (rows, rpw, tr.last_route()[0], abs(l - lo) / abs(lo), per, float(np.abs(go).max())), flush=True)
