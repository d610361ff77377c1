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
    # Code objects compiled at run time go to a directory of THIS session (unless the caller names one): a kernel left in
    # ~/.cache/21cmvae_amd by an earlier process is "ready at once" and changes the route of the first calls -- the table test's
    # "no run-time kernel yet" rows then fail for a reason that is not in the tree (r5: four pytest processes on one box).
    # The kernels build() put next to the library (kernel_cache/, read-only) are found as before.
    if not os.environ.get("V21_KERNEL_CACHE"):
        import tempfile
        d = tempfile.mkdtemp(prefix="v21_kernels_")
        os.chmod(d, 0o700)
        os.environ["V21_KERNEL_CACHE"] = d
        config._v21_kernel_dir = d


def pytest_unconfigure(config):
    d = getattr(config, "_v21_kernel_dir", None)
    if d:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
        os.environ.pop("V21_KERNEL_CACHE", None)


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
