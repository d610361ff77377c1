"""Probe: one epoch of the joint step (v21_joint_*) against the two models stepped one after the other, 30,000 rows, batch 256."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module("21cmvae_amd._native")
ctx = native.Context(0)
rng = np.random.default_rng(0)
n, B = 30000, 256
ae = native.Stack(ctx, [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]); ae.set_weights((rng.normal(size=ae.num_params) * 0.05).astype(np.float32))
em = native.Stack(ctx, [7, 352, 352, 352, 224, 9], [1, 1, 1, 1, 0]); em.set_weights((rng.normal(size=em.num_params) * 0.05).astype(np.float32))
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
ta, te = native.Trainer(ae, prec, B), native.Trainer(em, prec, B)
x = rng.normal(size=(n, 451)).astype(np.float32); p = rng.normal(size=(n, 7)).astype(np.float32)
w = np.full(n, 1 / 451, np.float32); w2 = np.full(n, 1 / 9, np.float32)
ta.set_data(0, x, None, w); te.set_data(0, p, np.zeros((n, 9), np.float32), w2)
j = native.Joint(ta, te, 1)
perm = rng.permutation(n).astype(np.int32)
j.run_epoch(perm, B)
t0 = time.perf_counter(); j.run_epoch(perm, B); t1 = time.perf_counter()
ta.run_epoch(perm, B); te.run_epoch(perm, B)
t2 = time.perf_counter(); ta.run_epoch(perm, B); te.run_epoch(perm, B); t3 = time.perf_counter()
steps = -(-n // B)
print(prec, "joint epoch %.2f ms (%.1f us per step pair); sequential %.2f ms (%.1f us per pair)" % ((t1 - t0) * 1e3, (t1 - t0) / steps * 1e6, (t3 - t2) * 1e3, (t3 - t2) / steps * 1e6))
