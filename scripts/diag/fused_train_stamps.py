"""Diagnostics: cycles of workgroup 0 / wave 0 per virtual layer of the fused training kernel (needs a library built with
-DV21_T_STAMPS: V21_LIB=.../libv21_tstamp.so).  python fused_train_stamps.py [rows]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["V21_FUSED_TRAIN_ROWS"] = "1"
native = importlib.import_module("21cmvae_amd._native")
from oracle import ref_numpy as ora
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = native.Context(0)
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
L = 5
rng = np.random.default_rng(0)
Ws, bs = ora.init_mlp(dims, seed=1)
st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
tr = native.Trainer(st, "f16", rows); tr.set_adam(lr=1e-4)
x = rng.uniform(-1, 1, size=(rows, 451)).astype(np.float32)
w = np.full(rows, 1.0 / 451, np.float32)
d_x, d_rw = ctx.malloc(x.nbytes), ctx.malloc(w.nbytes)
ctx.h2d(d_x, x); ctx.h2d(d_rw, w)
tr.enable_stamps(True)
for _ in range(30):
    tr.step_dev(d_x, None, d_rw, rows, rows)
ctx.sync()
s = tr.chain_stamps(32).astype(np.int64)
vd = dims + dims[-2:0:-1]          # virtual stack: forward layers, then activation-gradient layers
print("rows %d: kernel start -> input gathered+flushed %d cycles (100 MHz ticks x clock ratio: s_memtime counts shader cycles)" % (rows, s[1] - s[0]))
tot = s[2 + 2 * L - 1] - s[0]
for v in range(2 * L - 1):
    K, N = vd[v], vd[v + 1]
    mf = ((N + 31) // 32) * ((K + 15) // 16)
    cyc = s[3 + v] - s[2 + v]
    print("  virtual layer %d  %3d -> %3d  %4d MFMAs  %7d cycles  %6.1f cycles per MFMA (32 = pipe busy)" % (v, K, N, mf, cyc, cyc / mf))
print("  total %d cycles (first stamp to last); of these %d at the ring's rendezvous (counted vmcnt wait + barrier, %d of them)" % (tot, s[30], 47))
