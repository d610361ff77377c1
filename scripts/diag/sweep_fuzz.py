"""Diagnostics (r4): random sweeps (v21_sweep_*: up to 16 models of equal depth stepped in lock step with grouped launches)
against the same models trained one by one, and against a twin sweep bit for bit.
  python sweep_fuzz.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
from oracle import ref_numpy as ora
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DRY = os.environ.get("FUZZ_DRY") == "1"
ctx = None if DRY else native.Context.default()
HID = [1, 8, 16, 17, 31, 32, 33, 64, 96, 100, 128, 224, 256, 288, 352, 400, 512]
bad = 0
for c in range(cases):
    count = int(rng.choice([1, 2, 3, 5, 8, 9, 16]))
    L = int(rng.choice([2, 3, 4, 5]))
    din = int(rng.choice([7, 9, 33, 451])); dout = din if rng.random() < 0.6 else int(rng.choice([9, 17, 451]))
    act = [int(rng.integers(0, 2)) for _ in range(L - 1)] + [0]
    prec = ["f16", "f32", "bf16"][int(rng.integers(0, 3))]
    n = int(rng.choice([40, 256, 300, 700, 1500]))
    batch = int(rng.choice([1, 32, 100, 128, 256, 257, 600, 1024]))
    batch = min(batch, n)
    members = [[din] + [int(rng.choice(HID)) for _ in range(L - 1)] + [dout] for _ in range(count)]
    perm = rng.permutation(n).astype(np.int32) if rng.random() < 0.6 else None
    x = rng.uniform(-1, 1, size=(n, din)).astype(np.float32)
    y = None if din == dout and rng.random() < 0.7 else rng.normal(size=(n, dout)).astype(np.float32)
    w = (rng.uniform(0.5, 1.5, size=n) / dout).astype(np.float32)
    tag = "case %3d %-4s members %-2d L %d %4d->%-4d act %-12s n %-5d batch %-5d %s %s hidden %s" % (
        c, prec, count, L, din, dout, act, n, batch, "perm" if perm is not None else "seq ", "y=x" if y is None else "y  ",
        [m[1:-1] for m in members][:4])
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != c:
        continue
    print(tag, "...", flush=True)
    if DRY:
        continue

    def build():
        trs = []
        for k, dims in enumerate(members):
            Ws, bs = ora.init_mlp(dims, seed=100 * c + k)
            st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
            tr = native.Trainer(st, prec, batch); tr.set_adam(lr=1e-3 * (1 + k % 3))
            trs.append(tr)
        return trs
    try:
        solo = build()
        solo_loss = []
        for tr in solo:
            tr.set_data(0, x, y, w)
            solo_loss.append([tr.run_epoch(perm, batch) for _ in range(2)])
        groups = []
        for _ in range(2):
            trs = build(); trs[0].set_data(0, x, y, w)
            sw = native.Sweep(trs)
            losses = [sw.run_epoch(perm, batch) for _ in range(2)]
            groups.append((losses, [t.stack.get_weights() for t in trs]))
    except native.EngineError as e:
        print(tag, "refused:", str(e)[:120], flush=True)
        continue
    ltol, wtol = (2e-5, 5e-6) if prec == "f32" else ((3e-3, 3e-3) if prec == "f16" else (2e-2, 2e-2))
    (gl, gw), (gl2, gw2) = groups
    worst_l = worst_w = 0.0
    for k, tr in enumerate(solo):
        for e in range(2):
            worst_l = max(worst_l, abs(gl[e][k] - solo_loss[e][k] if False else gl[e][k] - solo_loss[k][e]) / abs(solo_loss[k][e]))
        ws = tr.stack.get_weights()
        worst_w = max(worst_w, float(np.abs(gw[k] - ws).max()))
    twin = all(np.array_equal(a, b) for a, b in zip(gw, gw2)) and gl == gl2
    finite = all(np.isfinite(a).all() for a in gw)
    flag = "OK " if worst_l <= ltol and worst_w <= wtol and twin and finite else "BAD"
    bad += flag == "BAD"
    print(tag, flag, "loss rel %.1e (tol %.0e) weights max diff %.1e (tol %.0e) twin identical %s" % (worst_l, ltol, worst_w, wtol, twin), flush=True)
print("cases %d, BAD %d" % (cases, bad))
