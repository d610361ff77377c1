"""Why can a sweep member's f32 weights differ from the same model trained alone by a learning rate's worth while the losses
agree to 1e-6?  (sweep_fuzz seed 204 case 9, r5.)  Reproduces the case, finds the member and element with the largest weight
difference, and prints that element's first-step gradient in both runs beside the layer's largest gradient.
  python sweep_case_adam_sensitivity.py [seed] [case]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep_fuzz as sf
from oracle import ref_numpy as ora
native = importlib.import_module("21cmvae_amd._native")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 204
case = int(sys.argv[2]) if len(sys.argv) > 2 else 9
k = [c for c in sf.gen_cases(case + 1, seed)][case]
print(sf.tag_of(k))
ctx = native.Context.default()
prec, n, batch, act, members = k["prec"], k["n"], k["batch"], k["act"], k["members"]
rng = np.random.default_rng(k["data_seed"])
perm = rng.permutation(n).astype(np.int32) if k["use_perm"] else None
x = rng.uniform(-1, 1, size=(n, k["din"])).astype(np.float32)
y = None if k["y_is_x"] else rng.normal(size=(n, k["dout"])).astype(np.float32)
w = (rng.uniform(0.5, 1.5, size=n) / k["dout"]).astype(np.float32)


def build():
    trs = []
    for j, dims in enumerate(members):
        Ws, bs = ora.init_mlp(dims, seed=100 * k["c"] + j)
        st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
        tr = native.Trainer(st, prec, batch); tr.set_adam(lr=1e-3 * (1 + j % 3))
        trs.append(tr)
    return trs


def run(grouped, steps_epochs):
    trs = build()
    out = []
    if grouped:
        trs[0].set_data(0, x, y, w)
        sw = native.Sweep(trs)
        for _ in range(steps_epochs):
            sw.run_epoch(perm, batch)
    else:
        for tr in trs:
            tr.set_data(0, x, y, w)
            for _ in range(steps_epochs):
                tr.run_epoch(perm, batch)
    return trs


if os.environ.get("SKIP_TRAIN") == "1":   # (diagnosis: the one-step comparison alone, in a fresh process)
    solo = grp = build()
else:
    solo, grp = run(False, 2), run(True, 2)
diffs = [float(np.abs(a.stack.get_weights() - b.stack.get_weights()).max()) for a, b in zip(solo, grp)]
j = int(os.environ["MEMBER"]) if os.environ.get("MEMBER") else int(np.argmax(diffs))
ws, wg = solo[j].stack.get_weights(), grp[j].stack.get_weights()
d = np.abs(ws - wg)
e = int(np.argmax(d))
dims = members[j]
offs = np.cumsum([0] + [a * b + b for a, b in zip(dims[:-1], dims[1:])])
layer = int(np.searchsorted(offs, e, side="right") - 1)
print("member %d dims %s lr %.0e: max |w_alone - w_grouped| = %.3e at arena element %d (layer %d, element %d of its [W; b] block); elements above 5e-6: %d of %d"
      % (j, dims, 1e-3 * (1 + j % 3), d[e], e, layer, e - offs[layer], int((d > 5e-6).sum()), d.size))
# ONE step of the whole first batch, both ways: the gradient itself
b1 = min(batch, n)
p1 = perm[:b1] if perm is not None else None
s1 = build()[j]; s1.set_data(0, x[:b1] if perm is None else x, (None if y is None else (y[:b1] if perm is None else y)), w[:b1] if perm is None else w)
s1.run_epoch(p1, b1); gs = s1.get_grad()
g_all = build(); g_all[0].set_data(0, x[:b1] if perm is None else x, (None if y is None else (y[:b1] if perm is None else y)), w[:b1] if perm is None else w)
sw = native.Sweep(g_all); sw.run_epoch(p1, b1); gg = g_all[j].get_grad()
lo, hi = offs[layer], offs[layer + 1]
print("first step, that element: gradient alone %.6e, grouped %.6e; the layer's largest |gradient| %.3e; the whole arena: max |g_alone - g_grouped| = %.3e (largest |g| %.3e)"
      % (gs[e], gg[e], float(np.abs(gs[lo:hi]).max()), float(np.abs(gs - gg).max()), float(np.abs(gs).max())))
big = np.where(d > 5e-6)[0]
rel = np.abs(gs[big]) / float(np.abs(gs).max())
print("the %d elements whose weights differ by more than 5e-6 after two epochs: their first-step |gradient| / the arena's largest is at most %.2e (median %.2e)"
      % (big.size, float(rel.max()) if big.size else 0.0, float(np.median(rel)) if big.size else 0.0))
print("Adam: w -= lr * m / (sqrt(v) + 1e-7): an element whose gradient is rounding noise of the sums moves by up to lr per step whatever the gradient's size")
# ... and which of the two is right: the float64 oracle's gradient of the same step
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import oracle_step
Ws, bs = ora.init_mlp(dims, seed=100 * k["c"] + j)
xb = x[:b1] if perm is None else x[p1]
tb = xb if y is None else (y[:b1] if perm is None else y[p1])
wb = w[:b1] if perm is None else w[p1]
lo64, go = oracle_step(Ws, bs, act, xb, tb, wb)
for name, g in (("alone", gs), ("grouped", gg)):
    cos = float(g @ go / (np.linalg.norm(g) * np.linalg.norm(go)))
    print("%-8s vs float64 oracle: max |g - g64| = %.3e (largest |g64| %.3e), cosine %.9f, per layer max diff %s"
          % (name, float(np.abs(g - go).max()), float(np.abs(go).max()), cos, ["%.1e" % float(np.abs(g[offs[i]:offs[i + 1]] - go[offs[i]:offs[i + 1]]).max()) for i in range(len(dims) - 1)]))
# is it ONE (row, unit) of the first layer whose ReLU falls on the other side of zero?  Then the two gradients differ in
# one column of [W0; b0] only, by that row's input times its activation gradient
N0 = dims[1]
dl = np.abs(gs[offs[0]:offs[1]] - gg[offs[0]:offs[1]]).reshape(dims[0] + 1, N0)
cols = np.where(dl.max(0) > 1e-8)[0]
print("first layer: the two gradients differ (> 1e-8) in column(s) %s of [W0; b0] only (%d elements)" % (cols.tolist(), int((dl > 1e-8).sum())))
if cols.size:
    W0 = Ws[0].astype(np.float64); b0 = bs[0].astype(np.float64)
    z = xb.astype(np.float64) @ W0[:, cols] + b0[cols]
    r = np.unravel_index(np.argmin(np.abs(z)), z.shape)
    z32 = (xb @ Ws[0][:, cols] + bs[0][cols])
    print("float64 pre-activation closest to zero in those columns: row %d unit %d: %.3e (numpy float32 product: %.3e) -- the ReLU's kink: d relu / dz is 0 on one side, 1 on the other"
          % (r[0], cols[r[1]], z[r], z32[r]))
