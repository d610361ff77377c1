"""Drop-in alias: ``import VeryAccurateEmulator as VAE; VAE.emulator.DirectEmulator(...)``
resolves to the MI355X engine (package ``21cmvae_amd``).  Unlike the reference's
``__init__`` nothing is downloaded."""
import importlib as _il
import sys as _sys

_pkg = _il.import_module("21cmvae_amd")
__version__ = "3.1.0+mi355x." + _pkg.__version__
__path__ = list(_pkg.__path__)

preprocess = _il.import_module("21cmvae_amd.preprocess")
_sys.modules[__name__ + ".preprocess"] = preprocess


def __getattr__(name):
    if name in ("emulator", "engine", "callbacks", "optimizers", "losses"):
        mod = _il.import_module("21cmvae_amd." + name)
        _sys.modules[__name__ + "." + name] = mod
        return mod
    raise AttributeError(name)
