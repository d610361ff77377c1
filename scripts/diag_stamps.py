"""Diagnostic only: per-ring-block cycle anatomy of the fused forward kernel.
Needs the stamp build:  make -C 21cmvae_amd/csrc stamp ; V21_LIB=21cmvae_amd/libv21_stamp.so python scripts/diag_stamps.py
Its run time is NOT a benchmark (stamps drain LDS reads at every block boundary)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module("21cmvae_amd._native")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
DIMS = [7, 352, 352, 352, 224, 451]
ctx = native.Context(0)
st = native.Stack(ctx, DIMS, [1, 1, 1, 1, 0])
rng = np.random.default_rng(0)
st.set_weights(rng.normal(scale=0.05, size=st.num_params).astype(np.float32))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
x = rng.uniform(-1, 1, size=(B, 7)).astype(np.float32)
d_x, d_y = ctx.malloc(x.nbytes), ctx.malloc(B * 451 * 4)
ctx.h2d(d_x, x)
rows_per_wg = 128  # f32: 4 waves x 1 column tile; f16/bf16 (x2sp): the same, two workgroups per CU
nwg = B // rows_per_wg
dbg = np.zeros((nwg * 4, 512), np.uint64)
d_dbg = ctx.malloc(dbg.nbytes)
for _ in range(3):
    st.forward_dev(d_x, 7, B, d_y, 451, prec, 0)
ctx.sync()
os.environ["V21_FUSED_DBG_PTR"] = str(d_dbg)
ctx.h2d(d_dbg, dbg)
st.forward_dev(d_x, 7, B, d_y, 451, prec, 0)
ctx.sync()
ctx.d2h(dbg, d_dbg)
t = dbg.astype(np.int64)
nb = int(sys.argv[3]) if len(sys.argv) > 3 else (39 if prec == "f32" else 58)  # ring blocks: 24 / 16 fragments each
start, end = t[:, 0], t[:, 127]
print("precision", prec, "blocks", nb, "waves", t.shape[0])
half = t.shape[0] // 2
for nm, sl in (("WGs 0..255 (first on their CU)", slice(0, half)), ("WGs 256..511 (second)", slice(half, None))):
    tt = t[sl]
    print(nm, ": start %d..%d  end %d..%d (cycles rel. to global first start)" % (
        (tt[:, 0] - t[:, 0].min()).min(), (tt[:, 0] - t[:, 0].min()).max(), (tt[:, 127] - t[:, 0].min()).min(), (tt[:, 127] - t[:, 0].min()).max()))
print("wave lifetime cycles: mean %.0f min %d max %d" % ((end - start).mean(), (end - start).min(), (end - start).max()))
print("first-start to last-end over the chip: %d cycles" % (end.max() - start.min()))
wait = np.stack([t[:, 2 * b + 1] - t[:, 2 * b] for b in range(nb)], 1)
comp = np.stack([(t[:, 2 * b + 2] if b + 1 < nb else t[:, 127]) - t[:, 2 * b + 1] for b in range(nb)], 1)
wait, comp = wait[:, :nb], comp[:, :nb]
print("block  wait(mean)  compute(mean)  compute(min)")
for b in range(nb):
    print("%3d %10.0f %12.0f %12d" % (b, wait[:, b].mean(), comp[:, b].mean(), comp[:, b].min()))
print("sum wait %.0f  sum compute %.0f" % (wait.mean(0).sum(), comp.mean(0).sum()))

nt = int(sys.argv[4]) if len(sys.argv) > 4 else 55
ts = t[:, 128:128 + nt]
d = np.diff(np.concatenate([ts, t[:, 127:128]], 1), axis=1)
print("tile  cycles(mean)  cycles(min)")
for g in range(nt):
    print("%3d %10.0f %10d" % (g, d[:, g].mean(), d[:, g].min()))

