"""Probe: where the host time of AutoEncoderEmulator.train() goes (cProfile, f16, reference recipe sizes)."""
import cProfile, importlib, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("21cmvae_amd.synth"); emu = importlib.import_module("21cmvae_amd.emulator")
optm = importlib.import_module("21cmvae_amd.optimizers")
data = synth.make_dataset()
ae = emu.AutoEncoderEmulator(precision="f16", **data)
ae.autoencoder.compile(optimizer=optm.Adam(1e-3), loss=emu.relative_mse_loss(ae.signal_train))
ae.emulator.compile(optimizer=optm.Adam(1e-3), loss=emu.mean_squared_error)
ae.train(epochs=2, verbose=0)
t0 = time.perf_counter(); ae.train(epochs=12, verbose=0); print("12 epochs: %.1f ms per epoch (both models)" % ((time.perf_counter() - t0) / 12 * 1e3))
pr = cProfile.Profile(); pr.enable(); ae.train(epochs=12, verbose=0); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
