"""Kernel variants selected by the host must agree BIT FOR BIT where they promise the same arithmetic.

The switches are read once per process, so every arm runs in a process of its own (at most one child at a time: the GPU
box allows few processes on the card) and prints a digest of the trained parameters.

* the 16-bit chain body is instantiated by feature set (csrc/train_chain.h: FEAT): a stack without a variational layer takes
  the instantiation without that code, ``V21_CHAIN_PLAIN=0`` sends it through the variational one -- same instructions on
  the path taken, same bits;
* the f32 gradient + Adam launch for steps of <= 256 rows stages its operands through LDS rows (csrc/dw_adam32.h),
  ``V21_DW32_LDS=0`` loads them straight into registers (csrc/gemm_nt.h): same split of the batch over the four waves, same
  MFMA order, same epilogue -- same bits, also on a ragged batch whose rows are shorter than a staged row.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, importlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
native = importlib.import_module("21cmvae_amd._native")
from oracle import ref_numpy as ora
prec, n, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dims, act = [451, 352, 9, 32, 352, 451], [1, 0, 1, 1, 0]
Ws, bs = ora.init_mlp(dims, seed=5)
ctx = native.Context.default()
st = native.Stack(ctx, dims, act)
st.set_weights(ora.flatten_params(Ws, bs))
tr = native.Trainer(st, prec, batch)
tr.set_adam(lr=1e-3)
rng = np.random.default_rng(11)
x = rng.normal(size=(n, 451)).astype(np.float32)
w = rng.uniform(0.5, 1.5, size=n).astype(np.float32) / 451
tr.set_data(0, x, None, w)
tr.set_data(1, x[: max(1, n // 2)], None, w[: max(1, n // 2)])
losses = [tr.run_epoch(ora.epoch_permutation(n, 3, ep), batch) for ep in range(2)]
val = tr.evaluate(1, batch)
it, m, v = tr.get_state()
h = hashlib.sha1()
for a in (st.get_weights(), m, v, np.array(losses + [val], np.float64)):
    h.update(np.ascontiguousarray(a).tobytes())
print("DIGEST", h.hexdigest(), it)
"""


def _run(env, *argv):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, *map(str, argv)], env=e, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")]
    assert line, out.stdout[-2000:]
    return line[-1]


@pytest.mark.parametrize("prec,n,batch", [("f16", 300, 128), ("bf16", 97, 97)])
def test_chain_instantiations_agree_bit_for_bit(prec, n, batch):
    assert _run({"V21_CHAIN_PLAIN": "1"}, prec, n, batch) == _run({"V21_CHAIN_PLAIN": "0"}, prec, n, batch)


@pytest.mark.parametrize("n,batch", [(300, 256), (203, 203), (90, 31)])
def test_f32_gradient_launches_agree_bit_for_bit(n, batch):
    assert _run({"V21_DW32_LDS": "1"}, "f32", n, batch) == _run({"V21_DW32_LDS": "0"}, "f32", n, batch)
