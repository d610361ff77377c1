"""Diagnostics (r5): does a sweep gain from running as TWO half-groups on two streams?  A 16-bit group step is two launches
of complementary character -- the chain of every member (latency-bound per workgroup, ~76 MB of HBM-side traffic for 32
members) and every member's gradients + Adam (bound by the optimizer state's bytes: 390 MB at 3.85 TB/s) -- run one after
the other.  Here the members are split into two sweeps on two contexts (two streams of the one GPU), each driven by its own
host thread: one half's chain launch can run while the other half's Adam launch waits for HBM.
  python scripts/diag/sweep_two_streams_probe.py [f16|f32] [members] [epochs]"""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
native = importlib.import_module("21cmvae_amd._native")
synth = importlib.import_module("21cmvae_amd.synth")
pp = importlib.import_module("21cmvae_amd.preprocess")
losses = importlib.import_module("21cmvae_amd.losses")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batch, spe = 256, 24
sig = synth.make_signals(batch * spe, seed=77)
y = pp.preproc(sig, sig)
rw = losses.relative_mse_loss(sig)._v21_row_weight(y).astype(np.float32)
cfgs = bench.sweep_configs(G)


def make(ctx, members):
    trs = []
    for i in members:
        lat, he, hd = cfgs[i]
        dims = [451, he, lat, hd[0], hd[1], 451]
        st = native.Stack(ctx, dims, bench.AE_ACT); st.set_weights(bench.glorot(dims, seed=50 + i))
        tr = native.Trainer(st, prec, batch); tr.set_adam(lr=1e-3)
        trs.append(tr)
    trs[0].set_data(0, y, None, rw)
    return native.Sweep(trs), trs


def run(sweeps, ctxs):
    for sw in sweeps:
        sw.run_epoch(None, batch)
    for c in ctxs:
        c.sync()
    out = [None] * len(sweeps)

    def work(k):
        for _ in range(epochs):
            out[k] = sweeps[k].run_epoch(None, batch)
        ctxs[k].sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(sweeps))]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    return G * epochs * spe / dt, 1e6 * dt / (epochs * spe), out


c0 = native.Context(0)
one, keep1 = make(c0, range(G))
r1 = run([one], [c0])
print("%s %d members, ONE group on one stream : %8.0f model-steps/s  %7.1f us per step of all members" % (prec, G, r1[0], r1[1]))
for parts in (2, 4):
    ctxs = [native.Context(0) for _ in range(parts)]
    sw, keep = [], []
    for k in range(parts):
        s, t = make(ctxs[k], range(k * G // parts, (k + 1) * G // parts))
        sw.append(s); keep.append(t)
    r = run(sw, ctxs)
    flat = [v for part in r[2] for v in part]
    same = max(abs(a - b) / abs(b) for a, b in zip(flat, r1[2][0]))
    print("%s %d members, %d groups on %d streams    : %8.0f model-steps/s  %7.1f us per step of all members  (%.2fx; epoch losses equal to the one-group run to %.1e)"
          % (prec, G, parts, parts, r[0], r[1], r[0] / r1[0], same))
    del sw, keep
