"""Diagnostic: where wave 0 of workgroup 0 spends a forward layer of the training chain (needs a build of train_chain.h with
extra chain_stamp() calls at slots 16 + 8 l + 0..6: after layer start, flush, bias + job, contraction, epilogue, tile loop,
zero-fill; not part of the shipped library -- see DESIGN.md K3'' for the numbers this produced)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
ctx = native.Context(0); st = native.Stack(ctx, dims, act)
rng = np.random.default_rng(0); st.set_weights(rng.normal(scale=0.05, size=st.num_params).astype(np.float32))
tr = native.Trainer(st, "f16", B)
tr.enable_stamps()
x = rng.normal(size=(B, 451)).astype(np.float32); w = np.full(B, 1 / 451, np.float32)
d_x, d_w = ctx.malloc(x.nbytes), ctx.malloc(w.nbytes); ctx.h2d(d_x, x); ctx.h2d(d_w, w)
for _ in range(50): tr.step_dev(d_x, None, d_w, B, B)
ctx.sync()
s = tr.chain_stamps(64).astype(np.int64)
for l in range(5):
    v = s[16 + 8 * l: 16 + 8 * l + 7]; prev = s[1 + l]
    print("L%d: start+%d flush/rwl %d  bias+job %d  contract %d  epilogue %d  loopexit %d  zero %d  barrier %d" % (
        l, v[0] - prev, v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], v[5] - v[4], v[6] - v[5], s[2 + l] - v[6]))
