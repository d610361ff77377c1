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
    return ("OK" if ok else "BAD"), "loss rel %.1e (tol %.0e) weights max diff %.1e (tol %.0e) twin identical %s" % (worst_l, ltol, worst_w, wtol, twin)


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
