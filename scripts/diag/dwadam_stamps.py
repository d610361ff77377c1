"""Phase stamps of eight workgroups of gemm_nt_dwadam_kernel (f32 gradients + Adam; diagnostic build `make fine`,
V21_LIB=21cmvae_amd/libv21_fine.so): kernel start (of that workgroup), operands landed, MFMAs done, partial tiles met,
epilogue issued, stores drained -- relative to the chain kernel's last stamp of the same step."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
NW = 4 if prec == "f32" else 8  # waves per workgroup of the gradient kernel
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
ctx = native.Context(0)
st = native.Stack(ctx, dims, act)
rng = np.random.default_rng(0)
st.set_weights(rng.normal(scale=0.05, size=st.num_params).astype(np.float32))
tr = native.Trainer(st, prec, B)
tr.enable_stamps()
x = rng.normal(size=(B, 451)).astype(np.float32); w = np.full(B, 1 / 451, np.float32)
d_x, d_w = ctx.malloc(x.nbytes), ctx.malloc(w.nbytes)
ctx.h2d(d_x, x); ctx.h2d(d_w, w)
for _ in range(20):
    tr.step_dev(d_x, None, d_w, B, B)
ctx.sync()
s = tr.chain_stamps(2048).astype(np.int64)
chain_end = s[11]
print("chain: start -> last stamp %d cycles" % (s[11] - s[0]))
d = s[1024:1024 + 8 * 8 * NW].reshape(8, 8, NW)
names = ["start", "loads issued", "MFMAs done", "tiles met", "epilogue issued", "stores drained"] if prec == "f32" else ["start", "state requested", "MFMAs done", "tiles met", "Adam done", "end"]
for b in range(8):
    print("workgroup %3d:" % (47 * b), "  ".join("%s %s" % (names[i], [int(v - d[b, 0].min()) for v in d[b, i]]) for i in range(6)))
    if d[b, 6].any():
        print("      (scalars read %s, Adam state requested %s)" % ([int(v - d[b, 0].min()) for v in d[b, 6]], [int(v - d[b, 0].min()) for v in d[b, 7]]))
