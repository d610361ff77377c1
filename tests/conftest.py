import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    """The product package; its directory name starts with a digit, hence importlib."""
    return importlib.import_module("21cmvae_amd" + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def shipped():
    """The reference's shipped AE-path weights (tests/golden/ae_path_weights.npz)."""
    d = np.load(os.path.join(GOLDEN, "ae_path_weights.npz"))

    def stack(stem):
        n = int(d[stem + "/n_layers"])
        return [d["%s/W%d" % (stem, i)] for i in range(n)], [d["%s/b%d" % (stem, i)] for i in range(n)]

    return {"raw": d, "ae_emulator": stack("ae_emulator"), "encoder": stack("encoder"), "decoder": stack("decoder")}


@pytest.fixture(scope="session")
def ctx():
    native = pkg("_native")
    return native.Context.default()
