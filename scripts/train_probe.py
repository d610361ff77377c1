"""Probe: time per optimizer step of the AE stack (used under rocprofv3)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module("21cmvae_amd._native")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
ctx = native.Context(0)
st = native.Stack(ctx, dims, act)
rng = np.random.default_rng(0)
st.set_weights(rng.normal(scale=0.05, size=st.num_params).astype(np.float32))
tr = native.Trainer(st, prec, B)
x = rng.normal(size=(B, 451)).astype(np.float32); w = np.full(B, 1 / 451, np.float32)
tr.set_data(0, x, None, w)              # the batch as the trainer's resident training set (what Model.fit steps on)
d_x, _, d_w, _ = tr.data_dev(0)
for _ in range(5):
    tr.step_dev(d_x, None, d_w, B, B)
ctx.sync()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step_dev(d_x, None, d_w, B, B)
t1 = time.perf_counter()
ctx.sync()
t2 = time.perf_counter()
print("B=%d %s: host enqueue %.1f us/step, total %.1f us/step" % (B, prec, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
try:
    tr.enable_stamps()   # (off while timing: eleven stamps cost 2-3 us per step)
    tr.step_dev(d_x, None, d_w, B, B); ctx.sync()
    s = tr.chain_stamps(2 + 5 + 1 + 4).astype(np.int64)
    d = np.diff(s)
    print("chain stamps (ticks):", d.tolist(), "total", int(s[-1] - s[0]))
except Exception as e:
    print("no stamps:", e)
