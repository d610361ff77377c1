"""Single-call latency of the class surface: DirectEmulator.predict on one parameter vector and on small batches."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
emu = importlib.import_module("21cmvae_amd.emulator")
synth = importlib.import_module("21cmvae_amd.synth")
data = synth.make_dataset(4000, 400, 400)
for prec in ("f32", "f16"):
    em = emu.DirectEmulator(hidden_dims=[352, 352, 352, 224], precision=prec, **data)
    p1 = data["par_test"][0]
    for n in (1, 32, 1000, 65536):
        x = p1 if n == 1 else synth.make_params(n, seed=5)
        for _ in range(3):
            em.predict(x)
        reps = 200 if n <= 1000 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            y = em.predict(x)
        dt = (time.perf_counter() - t0) / reps
        print("%s predict(%6d rows): %9.1f us per call, %.3g signals/s, out %s" % (prec, n, dt * 1e6, n / dt, y.shape))
