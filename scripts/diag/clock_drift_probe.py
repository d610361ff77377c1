"""Diagnostics (r5): how long the shader clock takes to settle under the headline kernel.  The clock-stamped instantiation
(v21_debug_forward_clocked) launched back to back for `seconds`; every 200th launch's stamps are kept: GHz (all
workgroups of that launch) and the launch's length in microseconds and kilocycles against the time since the first launch.
  python scripts/diag/clock_drift_probe.py [seconds] [f16|bf16]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
native = importlib.import_module("21cmvae_amd._native")
synth = importlib.import_module("21cmvae_amd.synth")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
ctx = native.Context.default()
st = native.Stack(ctx, bench.DIMS, bench.ACT)
st.set_weights(bench.glorot(bench.DIMS, seed=3))
B = bench.BATCH
x = np.random.default_rng(0).uniform(-1, 1, size=(B, 7)).astype(np.float32)
d_x, d_y = ctx.malloc(x.nbytes), ctx.malloc(B * 451 * 4)
ctx.h2d(d_x, x)
nwg = (B + 127) // 128
keep = 400
d_s = ctx.malloc((keep + 1) * nwg * 40)
ctx.memset(d_s, 0, (keep + 1) * nwg * 40)
time.sleep(1.0)   # from an idle chip
t0 = time.perf_counter(); k = 0; i = 0; stamps_t = []
while time.perf_counter() - t0 < seconds and k < keep:
    for j in range(200):
        slot = k if j == 199 else keep
        st.forward_clocked(d_x, 7, B, d_y, 451, d_s + slot * nwg * 40, prec, 0)
    ctx.sync(); stamps_t.append(time.perf_counter() - t0); k += 1
h = np.empty((keep + 1, nwg, 5), np.uint64); ctx.d2h(h, d_s)
print("t_s  GHz(all workgroups)  launch_us  kcycles_per_launch")
for q in range(k):
    s = h[q]
    dc = (s[:, 2] - s[:, 0]).astype(np.float64); dt = (s[:, 3] - s[:, 1]).astype(np.float64)
    ghz = dc.sum() / dt.sum() * 0.1
    span = float(s[:, 3].max() - s[:, 1].min()) * 0.01
    if q < 10 or q % 5 == 0:
        print("%.3f  %.3f  %.2f  %.1f" % (stamps_t[q], ghz, span, span * ghz))

# ---- the last kept launch, workgroup by workgroup: do the two workgroups of a CU finish together?
s = h[k - 1]
dc = (s[:, 2] - s[:, 0]).astype(np.float64) / 1e3
hw = (s[:, 4] >> np.uint64(8)).astype(np.int64)
xcc = (s[:, 4] & np.uint64(0xF)).astype(np.int64)
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)      # cu_id | sh_id | se_id | xcd
print("kcycles per workgroup of one launch: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (dc.min(), np.percentile(dc, 10), np.median(dc), np.percentile(dc, 90), dc.max()))
hist, edges = np.histogram(dc, bins=14)
print("histogram:", " ".join("%.0f-%.0f:%d" % (edges[i], edges[i + 1], hist[i]) for i in range(len(hist))))
ends = (s[:, 2]).astype(np.float64)
pairs = {}
for i in range(len(cu)):
    pairs.setdefault(int(cu[i]), []).append(i)
sizes = sorted(len(v) for v in pairs.values())
print("distinct CUs (by XCD / SE / SH / CU id of wave 0): %d; workgroups per CU: min %d max %d" % (len(pairs), sizes[0], sizes[-1]))
two = [v for v in pairs.values() if len(v) == 2]
if two:
    fast = np.array([min(dc[v[0]], dc[v[1]]) for v in two]); slow = np.array([max(dc[v[0]], dc[v[1]]) for v in two])
    print("CUs with two workgroups: %d; the faster of the pair %.1f kcycles (mean), the slower %.1f; slower / faster %.2f" % (len(two), fast.mean(), slow.mean(), (slow / fast).mean()))
