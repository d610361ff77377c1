"""Does reduced-precision TRAINING reach the same place?  The reference recipe (direct emulator, batch 256, Adam,
ReduceLROnPlateau) on the synthetic data set, once per precision from the same initial weights and shuffles."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
emu = importlib.import_module("21cmvae_amd.emulator"); synth = importlib.import_module("21cmvae_amd.synth")
optm = importlib.import_module("21cmvae_amd.optimizers"); cbm = importlib.import_module("21cmvae_amd.callbacks")
eng = importlib.import_module("21cmvae_amd.engine")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
data = synth.make_dataset()
for prec in ("f32", "f16", "bf16"):
    eng.set_random_seed(0)
    em = emu.DirectEmulator(hidden_dims=[288, 352, 288, 224], precision=prec, **data)
    em.emulator.compile(optimizer=optm.Adam(0.003), loss=emu.relative_mse_loss(em.signal_train))
    rl = cbm.ReduceLROnPlateau(monitor="val_loss", patience=5, factor=0.7, min_lr=1e-5)
    t0 = time.perf_counter()
    loss, val = em.train(epochs=epochs, callbacks=[rl], verbose=0)
    dt = time.perf_counter() - t0
    em.emulator.precision = "f32"  # judge every model with the exact forward
    err = em.test_error()
    print("%-4s %d epochs in %.1f s: loss %.3e val %.3e  test error mean %.4f %% median %.4f %%  lr end %.2e"
          % (prec, epochs, dt, loss[-1], val[-1], err.mean(), np.median(err), float(em.emulator.optimizer.lr)))
