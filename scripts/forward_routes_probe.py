"""Probe: the three forward routes on the headline stack (65,536 rows, f16, device-resident, fused transforms) and the
table-driven route on the sample notebook's custom stack -- run under rocprofv3 for per-kernel durations."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module("21cmvae_amd._native")
synth = importlib.import_module("21cmvae_amd.synth")
pp = importlib.import_module("21cmvae_amd.preprocess")
ctx = native.Context(0)
B = 65536
par_train = synth.make_params(synth.N_TRAIN, seed=1, corners=True)
ps, ss = pp.ParamStats.of(par_train), pp.SignalStats.of(synth.make_signals(4096, seed=301))
params = synth.make_params(B, seed=1000, dtype=np.float32)
d_x, d_y = ctx.malloc(params.nbytes), ctx.malloc(B * 451 * 4)
ctx.h2d(d_x, params)
flags = native.FWD_IN_TRANSFORM | native.FWD_OUT_TRANSFORM
rng = np.random.default_rng(3)
for dims, act, routes in (([7, 352, 352, 352, 224, 451], [1, 1, 1, 1, 0], (("compiled fused kernel", 0), ("fused kernel instantiated at run time (FWD_FORCE_JIT)", native.FWD_FORCE_JIT),
                                                                           ("table-driven one-launch (FWD_FORCE_CHAIN)", native.FWD_FORCE_CHAIN),
                                                                           ("per-layer K-loop (FWD_FORCE_GENERIC)", native.FWD_FORCE_GENERIC))),
                           ([7, 64, 128, 451], [1, 1, 0], (("fused kernel instantiated at run time (default route, r4)", 0), ("table-driven one-launch (FWD_FORCE_CHAIN; default until r3)", native.FWD_FORCE_CHAIN),
                                                           ("per-layer K-loop (FWD_FORCE_GENERIC)", native.FWD_FORCE_GENERIC)))):
    st = native.Stack(ctx, dims, act)
    st.set_weights((rng.normal(size=st.num_params) * 0.05).astype(np.float32))
    st.set_input_transform(ps.log_mask, ps.zero_floor, ps.lo, ps.hi)
    st.set_output_transform(ss.std, ss.mean)
    if not st.has_fused("f16"):
        st.jit("f16")   # (waits for the compilation, or finds the code object build() left in kernel_cache/)
    for name, fl in routes:
        for _ in range(20):
            st.forward_dev(d_x, dims[0], B, d_y, 451, "f16", flags | fl)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            st.forward_dev(d_x, dims[0], B, d_y, 451, "f16", flags | fl)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 50
        print("%s  %-62s %8.1f us per call  %8.1f M signals/s" % ("-".join(map(str, dims)), name, dt * 1e6, B / dt / 1e6))
