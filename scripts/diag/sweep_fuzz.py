"""Diagnostics (r4; r5: importable -- tests/test_fuzz_gpu.py runs a seeded slice under `pytest -m gpu`): random sweeps
(v21_sweep_*: models of equal depth stepped in lock step with grouped launches) against the same models trained one by one,
and against a twin sweep bit for bit.
  python sweep_fuzz.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HID = [1, 8, 16, 17, 31, 32, 33, 64, 96, 100, 128, 224, 256, 288, 352, 400, 512]
COUNTS = [1, 2, 3, 5, 8, 9, 16, 24, 40, 64]


def gen_cases(cases, seed, max_count=64):
    rng = np.random.default_rng(seed)
    for c in range(cases):
        count = int(rng.choice([k for k in COUNTS if k <= max_count]))
        L = int(rng.choice([2, 3, 4, 5]))
        din = int(rng.choice([7, 9, 33, 451])); dout = din if rng.random() < 0.6 else int(rng.choice([9, 17, 451]))
        act = [int(rng.integers(0, 2)) for _ in range(L - 1)] + [0]
        prec = ["f16", "f32", "bf16"][int(rng.integers(0, 3))]
        n = int(rng.choice([40, 256, 300, 700, 1500]))
        batch = min(int(rng.choice([1, 32, 100, 128, 256, 257, 600, 1024])), n)
        if count > 16 and n // batch > 20:      # (keep a many-member case within seconds)
            batch = max(batch, n // 10)
        members = [[din] + [int(rng.choice(HID)) for _ in range(L - 1)] + [dout] for _ in range(count)]
        yield dict(c=c, count=count, L=L, din=din, dout=dout, act=act, prec=prec, n=n, batch=batch, members=members,
                   use_perm=bool(rng.random() < 0.6), y_is_x=bool(din == dout and rng.random() < 0.7), data_seed=int(rng.integers(0, 1 << 30)))


def tag_of(k):
    return "case %3d %-4s members %-2d L %d %4d->%-4d act %-12s n %-5d batch %-5d %s %s hidden %s" % (
        k["c"], k["prec"], k["count"], k["L"], k["din"], k["dout"], k["act"], k["n"], k["batch"], "perm" if k["use_perm"] else "seq ",
        "y=x" if k["y_is_x"] else "y  ", [m[1:-1] for m in k["members"]][:4])


def run_case(ctx, k):
    """-> ("OK" | "BAD" | "refused", message)"""
    native = importlib.import_module("21cmvae_amd._native")
    from oracle import ref_numpy as ora
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
        return "refused", str(e)[:120]
    ltol, wtol = (2e-5, 5e-6) if prec == "f32" else ((3e-3, 3e-3) if prec == "f16" else (2e-2, 2e-2))
    (gl, gw), (gl2, gw2) = groups
    worst_l = worst_w = 0.0
    for j, tr in enumerate(solo):
        for e in range(2):
            worst_l = max(worst_l, abs(gl[e][j] - solo_loss[j][e]) / abs(solo_loss[j][e]))
        worst_w = max(worst_w, float(np.abs(gw[j] - tr.stack.get_weights()).max()))
    twin = all(np.array_equal(a, b) for a, b in zip(gw, gw2)) and gl == gl2
    finite = all(np.isfinite(a).all() for a in gw)
    ok = worst_l <= ltol and worst_w <= wtol and twin and finite
    msg = "loss rel %.1e (tol %.0e) weights max diff %.1e (tol %.0e) twin identical %s" % (worst_l, ltol, worst_w, wtol, twin)
    if not ok and worst_l <= ltol and twin and finite and worst_w > wtol:
        # weights apart, losses together: a ReLU on the other side of zero in one of the two runs?  (Two f32 kernels that sum
        # in different orders may disagree on the sign of a pre-activation of size 1e-7; the unit's whole column of [W; b]
        # then gets another gradient for that row, and Adam turns ANY gradient difference into a step of up to lr.)  Not
        # assumed: the first diverging step is looked up and the pre-activation shown, or the case stays BAD.
        j = int(np.argmax([float(np.abs(gw[i] - solo[i].stack.get_weights()).max()) for i in range(len(solo))]))
        why = explain_by_relu_kink(ctx, k, j, build, x, y, w, perm)
        if why:
            return "OK", msg + " | member %d EXPLAINED: %s" % (j, why)
        msg += " | member %d: no pre-activation at zero found in the first diverging step" % j
    return ("OK" if ok else "BAD"), msg


def explain_by_relu_kink(ctx, k, j, build, x, y, w, perm):
    """Steps member j alone and inside the group, one optimizer step at a time, up to the first step whose GRADIENTS differ by
    more than summation noise; there, in float64 and with the weights both runs still share, looks for the pre-activation at
    zero that explains it: the top-most layer whose gradient blocks differ must be a ReLU layer, differ in few columns only,
    and one of those units must have |z| < 1e-5 of the layer's largest for a row of the step.  -> text or None."""
    native = importlib.import_module("21cmvae_amd._native")
    n, batch, act, dims = k["n"], k["batch"], k["act"], k["members"][j]
    offs = np.cumsum([0] + [a * b + b for a, b in zip(dims[:-1], dims[1:])])
    alone = build()[j]
    grp = build()
    sw = None
    order = perm if perm is not None else np.arange(n, dtype=np.int32)
    step = 0
    for epoch in range(2):
        for first in range(0, n, batch):
            rows = order[first:first + batch].astype(np.int32)
            w_before = alone.stack.get_weights().astype(np.float64)
            # (one optimizer step = an epoch over a training set that is this step's rows)
            xs, ys, ws_ = x[rows], (None if y is None else y[rows]), w[rows]
            alone.set_data(0, xs, ys, ws_); grp[0].set_data(0, xs, ys, ws_)
            if sw is None:
                sw = native.Sweep(grp)
            alone.run_epoch(None, len(rows)); sw.run_epoch(None, len(rows))
            ga, gg = alone.get_grad(), grp[j].get_grad()
            scale = float(np.abs(ga).max())
            if float(np.abs(ga - gg).max()) <= 1e-6 * scale:
                step += 1
                continue
            # the top-most layer whose blocks differ
            top = max(l for l in range(len(dims) - 1) if float(np.abs(ga[offs[l]:offs[l + 1]] - gg[offs[l]:offs[l + 1]]).max()) > 1e-6 * scale)
            if not act[top]:
                return None
            K, N = dims[top], dims[top + 1]
            d = np.abs(ga[offs[top]:offs[top + 1]] - gg[offs[top]:offs[top + 1]]).reshape(K + 1, N)
            cols = np.where(d.max(0) > 1e-6 * scale)[0]
            if cols.size > 4:
                return None
            h = x[rows].astype(np.float64)
            for l in range(top + 1):
                Wl = w_before[offs[l]:offs[l] + dims[l] * dims[l + 1]].reshape(dims[l], dims[l + 1])
                bl = w_before[offs[l] + dims[l] * dims[l + 1]:offs[l + 1]]
                z = h @ Wl + bl
                h = np.maximum(z, 0) if act[l] else z
            zc = np.abs(z[:, cols])
            r = np.unravel_index(np.argmin(zc), zc.shape)
            if zc[r] > 1e-5 * float(np.abs(z).max()):
                return None
            return ("optimizer step %d: the gradients differ in column(s) %s of layer %d's [W; b] only; float64 pre-activation of unit %d, row %d of the step: %.2e (largest of the layer %.2e) -- a ReLU at its kink"
                    % (step, cols.tolist(), top, int(cols[r[1]]), int(r[0]), float(z[r[0], cols[r[1]]]), float(np.abs(z).max())))
    return None


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    DRY = os.environ.get("FUZZ_DRY") == "1"
    ctx = None if DRY else importlib.import_module("21cmvae_amd._native").Context.default()
    bad = 0
    for k in gen_cases(cases, seed):
        if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != k["c"]:
            continue
        print(tag_of(k), "...", flush=True)
        if DRY:
            continue
        status, msg = run_case(ctx, k)
        bad += status == "BAD"
        print(tag_of(k), status, msg, flush=True)
    print("cases %d, BAD %d" % (cases, bad))
