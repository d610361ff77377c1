"""Diagnostics (r5): the per-layer cycle budget of one optimizer step of the reference's autoencoder on the
16-bit chain route -- what workgroup 0 of train_chain_kernel spends per phase (s_memtime stamps, v21_trainer_enable_stamps)
beside what the phase NEEDS by the two resources that could bound it: the matrix pipe (MFMAs of the layer's tiles over the
four SIMDs, 32 cycles each) and the CU's load path (the layer's packed weights at the 64 B/clk one CU pulls from L2).
  python scripts/diag/step_budget.py [rows] [f16|bf16]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
native = importlib.import_module("21cmvae_amd._native")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
L = len(act)
ctx = native.Context(0)
st = native.Stack(ctx, dims, act)
rng = np.random.default_rng(0)
st.set_weights(rng.normal(scale=0.05, size=st.num_params).astype(np.float32))
tr = native.Trainer(st, prec, B)
x = rng.normal(size=(B, 451)).astype(np.float32); w = np.full(B, 1 / 451, np.float32)
tr.set_data(0, x, None, w)
d_x, _, d_w, _ = tr.data_dev(0)
for _ in range(200):
    tr.step_dev(d_x, None, d_w, B, B)
ctx.sync()
t0 = time.perf_counter()
for _ in range(200):
    tr.step_dev(d_x, None, d_w, B, B)
ctx.sync()
step_us = (time.perf_counter() - t0) / 200 * 1e6
tr.enable_stamps()
samples = []
for _ in range(25):
    tr.step_dev(d_x, None, d_w, B, B); ctx.sync()
    samples.append(np.diff(tr.chain_stamps(2 + L + 1 + (L - 1)).astype(np.int64)))
d = np.median(np.array(samples), axis=0)
steps16 = lambda n: ((n + 15) // 16 + 3) // 4 * 4            # train_kernels.h: chain_steps (k-steps of 16, padded to chunks of 4)
names = ["gather rows + targets"] + ["forward %d -> %d" % (dims[l], dims[l + 1]) for l in range(L)] + ["loss + dL/dz"] + \
        ["activation gradient %d <- %d" % (dims[l], dims[l + 1]) for l in range(L - 1, 0, -1)]
need = [(0, 0)]
for l in range(L):
    K, N = dims[l], dims[l + 1]
    nt, ks = (N + 31) // 32, steps16(K)
    need.append((nt * ((K + 15) // 16) * 32 / 4, nt * ks * 1024 / 64))
need.append((0, 0))
for l in range(L - 1, 0, -1):
    K, N = dims[l], dims[l + 1]
    kt, ns = (K + 31) // 32, steps16(N)
    need.append((kt * ((N + 15) // 16) * 32 / 4, kt * ns * 1024 / 64))
print("autoencoder %s, %d rows per step, %s: %.1f us per step (wall, 200 steps); chain kernel, workgroup 0, median of 25 stamped steps" % (dims, B, prec, step_us))
print("%-34s %9s %14s %16s" % ("phase", "cycles", "MFMA need/SIMD", "weights @64 B/clk"))
for n, c, (m, wb) in zip(names, d, need):
    print("%-34s %9.0f %14.0f %16.0f" % (n, c, m, wb))
tot_need = sum(max(m, wb) for m, wb in need)
print("%-34s %9.0f %14s %16.0f   <- sum over layers of max(MFMA need, weight-stream need)" % ("chain kernel, total", d.sum(), "", tot_need))
big = [i for i, (m, wb) in enumerate(need) if max(m, wb) > 3000]
small = [i for i, (m, wb) in enumerate(need) if 0 < max(m, wb) <= 3000]
print("three big layers: %.0f cycles measured against %.0f needed; six small layers: %.0f against %.0f; gather %.0f; loss %.0f" % (
    d[big].sum(), sum(max(*need[i]) for i in big), d[small].sum(), sum(max(*need[i]) for i in small), d[0], d[L + 1]))
