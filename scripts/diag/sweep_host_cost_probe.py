"""Diagnostics (r5): the HOST's share of a sweep step -- groups of tiny members (the kernels take ~20 us whatever the group
size) at 8 / 32 / 64 members: the growth of the step time with the member count is host work per member and step
(ensure_copies, Adam step sizes, launch arguments), not GPU time.   python scripts/diag/sweep_host_cost_probe.py [f16|f32]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
native = importlib.import_module("21cmvae_amd._native")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
ctx = native.Context(0)
rng = np.random.default_rng(0)
batch, spe = 32, 200
x = rng.normal(size=(batch * spe, 33)).astype(np.float32); w = np.full(batch * spe, 1 / 33, np.float32)
for G in (8, 32, 64):
    trs = []
    for k in range(G):
        dims = [33, 16, 4, 16, 33]
        st = native.Stack(ctx, dims, [1, 0, 1, 0]); st.set_weights(bench.glorot(dims, seed=k))
        tr = native.Trainer(st, prec, batch); tr.set_adam(lr=1e-3); trs.append(tr)
    trs[0].set_data(0, x, None, w)
    sw = native.Sweep(trs)
    sw.run_epoch(None, batch); ctx.sync()
    t0 = time.perf_counter()
    sw.run_epoch(None, batch); ctx.sync()
    print("%s %2d tiny members: %.1f us per group step" % (prec, G, (time.perf_counter() - t0) / spe * 1e6))
    del sw, trs
