"""Which layer's gradient differs between trainers built the same way?  (diagnosis of a bitwise-twin failure)
usage: twin_layer_diff.py STACK PREC ROWS MAX_BATCH [perm]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from conftest import pkg
from helpers import STACKS, stack_data, init_weights

name, prec, rows, mb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
use_perm = len(sys.argv) > 5
native = pkg("_native")
ctx = native.Context(0)
dims, act = STACKS[name]
x, y, w = stack_data(dims, rows, seed=31)
perm = np.random.default_rng(9).permutation(rows).astype(np.int32) if use_perm else None
Ws, bs, flat = init_weights(dims, seed=len(dims) * 7 + dims[1])
offs = np.cumsum([0] + [a.size + b.size for a, b in zip(Ws, bs)])
for flag in ("0", "1", "0"):
    os.environ["V21_DW_XROWS"] = flag
    gs = []
    for k in range(4):
        st = native.Stack(ctx, dims, act); st.set_weights(flat)
        tr = native.Trainer(st, prec, mb); tr.set_adam(lr=1e-3)
        tr.set_data(0, x, y, w)
        l = tr.run_epoch(perm, rows)
        gs.append((l, tr.get_grad(), tr.last_route()[0]))
    for k in range(1, 4):
        d = np.abs(gs[k][1] - gs[0][1])
        per = [float(d[offs[i]:offs[i + 1]].max()) for i in range(len(offs) - 1)]
        where = [int(np.argmax(d[offs[i]:offs[i + 1]])) for i in range(len(offs) - 1)]
        print("xrows", flag, "trainer", k, "vs 0: loss equal", gs[k][0] == gs[0][0], gs[k][2], "max |dg| per layer",
              ["%.2e" % p for p in per], "at", where, "count", [int((d[offs[i]:offs[i + 1]] > 0).sum()) for i in range(len(offs) - 1)])
