"""21cmVAE hot path on MI355X: batched predict()/train() of the 21-cm global-signal
emulators behind the reference's own class surface (see emulator.py, preprocess.py).

The package directory name starts with a digit, so import it with
``importlib.import_module("21cmvae_amd")`` or through the drop-in alias package
``VeryAccurateEmulator`` at the repository root.
"""
__version__ = "0.1.0"

from . import preprocess  # noqa: F401  (numpy only; needs no GPU)


def __getattr__(name):  # emulator/engine import the C-ABI binding lazily
    if name in ("emulator", "engine", "callbacks", "optimizers", "losses", "h5lite", "synth", "_native"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
