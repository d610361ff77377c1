"""Probe: MFMA occupancy of the WIDEST HIDDEN LAYER of the headline stack (352 -> 352, BASELINE.json's ">= 50 % MFMA on the
widest hidden layer") without a diagnostic build: a stack that is almost nothing else -- 7 -> 352 x 6 -> 9, five
352 -> 352 layers = 99.5 % of its multiply-adds, 28 B in and 36 B out per row -- through the same fused kernel,
instantiated at run time (csrc/jit.hip), 65,536 rows, device-resident.  Run plainly for the launch time, under
`rocprofv3 --kernel-trace --stats` for the kernel duration and under `--pmc SQ_VALU_MFMA_BUSY_CYCLES ... / GRBM_GUI_ACTIVE`
for the pipe-busy fraction (scripts/pmc_timed.py fused_fwd 200 0 <csv>...)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module("21cmvae_amd._native")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 352
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ctx = native.Context(0)
B = 65536
dims = [7] + [width] * depth + [9]
act = [1] * depth + [0]
rng = np.random.default_rng(3)
st = native.Stack(ctx, dims, act)
st.set_weights((rng.normal(size=st.num_params) * 0.05).astype(np.float32))
st.jit(prec)
x = rng.uniform(-1, 1, size=(B, 7)).astype(np.float32)
d_x, d_y = ctx.malloc(x.nbytes), ctx.malloc(B * 9 * 4)
ctx.h2d(d_x, x)
t_s = time.perf_counter()
while time.perf_counter() - t_s < 0.3:   # (the clock settles to what the chip holds under this load)
    for _ in range(20):
        st.forward_dev(d_x, 7, B, d_y, 9, prec, 0)
    ctx.sync()
K = 200
ctx.clock_probe_start(K * 0.06, 25.0)    # (about as long as the K launches beside it)
t0 = time.perf_counter()
for _ in range(K):
    st.forward_dev(d_x, 7, B, d_y, 9, prec, 0)
ctx.sync()
dt = (time.perf_counter() - t0) / K
ghz = ctx.clock_probe_read()
macs = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
hidden = (depth - 1) * width * width
tf = 2.0 * macs * B / dt / 1e12
peak = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}[prec]
print("stack %s  %s: %.1f us per launch, %.0f TFLOP/s = %.3f of the %.0f TFLOP/s dense peak; %d x %d->%d layers = %.1f %% of the multiply-adds"
      % ("-".join(map(str, dims)), prec, dt * 1e6, tf, tf / peak, peak, depth - 1, width, width, 100.0 * hidden / macs))
print("clock beside the launches: %.2f GHz (%.2f .. %.2f, %d samples) -> %.3f of the peak at that clock"
      % (ghz["ghz_mean"], ghz["ghz_min"], ghz["ghz_max"], ghz["samples"], tf / (peak * ghz["ghz_mean"] / 2.4)))
