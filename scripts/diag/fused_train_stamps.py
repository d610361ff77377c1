"""Diagnostics: shader cycles of workgroup 0 / wave 0 per virtual layer of a fused training kernel, and the start / end of EVERY
workgroup on the constant 100 MHz clock (needs a library built with -DV21_T_STAMPS: `make -C 21cmvae_amd/csrc tstamp`,
V21_LIB=.../libv21_tstamp.so; V21_FUSED_TRAIN16=0 / 1 picks csrc/fused_train.h / fused_train16.h).
  python fused_train_stamps.py [rows]"""
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
tr.set_data(0, x, None, w)              # the batch as the trainer's resident training set (what Model.fit steps on)
d_x, _, d_rw, _ = tr.data_dev(0)
tr.enable_stamps(True)
for _ in range(30):
    tr.step_dev(d_x, None, d_rw, rows, rows)
ctx.sync()
s = tr.chain_stamps(32).astype(np.int64)
allw = tr.chain_stamps(2048).astype(np.int64)[64:]
t16 = os.environ.get("V21_FUSED_TRAIN16") == "1"
nwg = -(-rows // (64 if t16 else 128)); nwg = (nwg + 7) // 8 * 8
st, en = allw[0:2 * nwg:2], allw[1:2 * nwg:2]
live = en > 0
if live.any():
    t0 = st[live].min()
    us = lambda ticks: ticks * 0.01      # 100 MHz ticks -> microseconds
    dur = (en - st)[live]
    print("all %d workgroups (wave 0, constant 100 MHz clock): start 0 .. %.1f us after the first; duration min / median / max %.1f / %.1f / %.1f us; last end %.1f us after the first start"
          % (live.sum(), us((st[live] - t0).max()), us(dur.min()), us(np.median(dur)), us(dur.max()), us((en[live] - t0).max())))
    for x in range(8):
        sel = live & (np.arange(nwg) % 8 == x)
        print("   physical workgroups b %% 8 == %d: start %.1f .. %.1f us, duration median %.1f us, last end %.1f us"
              % (x, us((st[sel] - t0).min()), us((st[sel] - t0).max()), us(np.median((en - st)[sel])), us((en[sel] - t0).max())))
print("rows %d: kernel start -> input gathered+flushed %d cycles (100 MHz ticks x clock ratio: s_memtime counts shader cycles)" % (rows, s[1] - s[0]))
tot = s[2 + 2 * L - 1] - s[0]
for v in range(2 * L - 1):
    K, N = vd[v], vd[v + 1]
    t16 = os.environ.get("V21_FUSED_TRAIN16") == "1"      # fused_train16.h: 16 x 16 x 32 MFMAs (16 cycles of the pipe each)
    mf = ((N + 15) // 16) * ((K + 31) // 32) if t16 else ((N + 31) // 32) * ((K + 15) // 16)
    cyc = s[3 + v] - s[2 + v]
    print("  virtual layer %d  %3d -> %3d  %4d MFMAs  %7d cycles  %6.1f cycles per MFMA (%d = pipe busy)" % (v, K, N, mf, cyc, cyc / mf, 16 if t16 else 32))
print("  total %d cycles (first stamp to last); of these %d at the ring's rendezvous (counted vmcnt wait + barrier, %d of them)" % (tot, s[30], 47))
