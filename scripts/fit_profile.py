import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
synth = importlib.import_module("21cmvae_amd.synth"); emu = importlib.import_module("21cmvae_amd.emulator")
optm = importlib.import_module("21cmvae_amd.optimizers"); native = importlib.import_module("21cmvae_amd._native")
data = synth.make_dataset()
for prec in ("f16", "f32"):
    ae = emu.AutoEncoderEmulator(precision=prec, **data)
    ae.autoencoder.compile(optimizer=optm.Adam(1e-3), loss=emu.relative_mse_loss(ae.signal_train))
    ae.emulator.compile(optimizer=optm.Adam(1e-3), loss=emu.mean_squared_error)
    ae.train(epochs=1, verbose=0)
    tr = ae.autoencoder._trainer
    n = data["par_train"].shape[0]
    perm = np.random.default_rng(0).permutation(n).astype(np.int32)
    ctx = native.Context.default()
    for name, fn, reps in (("run_epoch (96 steps)", lambda: tr.run_epoch(perm, 256), 10), ("evaluate (val, 11 batches)", lambda: tr.evaluate(1, 256), 10)):
        fn(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        ctx.sync(); print(prec, name, "%.3f ms" % ((time.perf_counter() - t0) / reps * 1e3))
    t0 = time.perf_counter(); ae.train(epochs=5, verbose=0); print(prec, "train() per epoch (both models) %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
