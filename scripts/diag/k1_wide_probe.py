"""Diagnostics (r4): the headline kernel with HALF the LDS bytes per MFMA -- one workgroup per CU, one
wave per SIMD with two column tiles (fused_fwd<S1, PrecF16>, instantiated through the run-time route with V21_JIT_WIDE=1) --
against the shipped form (two workgroups per CU, one column tile per wave), with the shader clock sampled beside each."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["V21_JIT_WIDE"] = "1"
native = importlib.import_module("21cmvae_amd._native")
dims, act, B = [7, 352, 352, 352, 224, 451], [1, 1, 1, 1, 0], 65536
ctx = native.Context.default()
rng = np.random.default_rng(3)
flat = []
for k, n in zip(dims[:-1], dims[1:]):
    lim = np.sqrt(6.0 / (k + n)); flat += [rng.uniform(-lim, lim, size=(k, n)).astype(np.float32).ravel(), rng.normal(scale=0.05, size=n).astype(np.float32)]
st = native.Stack(ctx, dims, act); st.set_weights(np.concatenate(flat))
x = rng.uniform(-1, 1, size=(B, 7)).astype(np.float32)
d_x, d_y = ctx.malloc(x.nbytes), ctx.malloc(B * 451 * 4)
ctx.h2d(d_x, x)
for prec in ("f16", "bf16"):
    for name, fl in (("shipped x2sp (2 workgroups per CU, 1 column tile per wave)", 0), ("wide (1 workgroup per CU, 2 column tiles per wave)", native.FWD_FORCE_JIT)):
        t0 = time.time()
        while time.time() - t0 < 0.4:
            for _ in range(100):
                st.forward_dev(d_x, 7, B, d_y, 451, prec, fl)
            ctx.sync()
        a, b = ctx.event(), ctx.event()
        ctx.record(a)
        for _ in range(200):
            st.forward_dev(d_x, 7, B, d_y, 451, prec, fl)
        ctx.record(b); ctx.sync()
        us = ctx.elapsed_ms(a, b) / 200 * 1e3
        for _ in range(200):
            st.forward_dev(d_x, 7, B, d_y, 451, prec, fl)
        ctx.clock_probe_start(0.6 * us * 200 / 1e3, period_us=50)
        ctx.sync()
        ck = ctx.clock_probe_read()
        tf = 860288 * B / (us * 1e-6) / 1e12
        print("%s %-62s %6.2f us  %5.0f TFLOP/s = %.3f of 2.5 PF; clock %.3f GHz (min %.3f max %.3f, %d samples) -> %.3f of the peak at that clock"
              % (prec, name, us, tf, tf / 2500, ck["ghz_mean"], ck["ghz_min"], ck["ghz_max"], ck["samples"], tf / (2500 * ck["ghz_mean"] / 2.4)), flush=True)
