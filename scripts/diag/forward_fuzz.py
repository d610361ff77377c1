"""Diagnostics (r4; r5: importable -- tests/test_fuzz_gpu.py runs a seeded slice under `pytest -m gpu`): random forward calls
-- stack, precision, rows, route, transforms on / off, float32 / float64 input, host arrays (pinned result pool, sliced
copies) or device buffers with row strides -- against the float64 oracle, and the same call twice (identical bits).  A
route that cannot take the stack must refuse, never answer wrongly.
  python forward_fuzz.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
FAMILIES = [([7, 352, 352, 352, 224, 451], [1, 1, 1, 1, 0]), ([7, 288, 352, 288, 224, 451], [1, 1, 1, 1, 0]),
            ([7, 352, 352, 352, 224, 9, 32, 352, 451], [1, 1, 1, 1, 0, 1, 1, 0]), ([9, 32, 352, 451], [1, 1, 0]),
            ([7, 64, 128, 451], [1, 1, 0]), ([451, 352, 9], [1, 0]), ([7, 288, 352, 288, 224, 9], [1, 1, 1, 1, 0])]
WIDTHS = [1, 7, 9, 16, 17, 31, 32, 33, 64, 65, 100, 128, 224, 288, 352, 400, 451, 512, 600]
ROWS = [1, 2, 31, 32, 33, 255, 256, 257, 1000, 4095, 4096, 4097, 8193, 20000, 65536, 65553, 70001]
ROUTES = [("default", 0), ("generic", 4), ("table", 16), ("jit", 32), ("no_small", 8)]   # include/v21.h: V21_FWD_*


def gen_cases(cases, seed, families_only=False):
    """families_only: stacks whose run-time kernels build() prebuilt (a pytest slice must not wait for the compiler)."""
    rng = np.random.default_rng(seed)
    for c in range(cases):
        fam = rng.random() < 0.6 or families_only
        if fam:
            dims, act = FAMILIES[int(rng.integers(0, len(FAMILIES)))]
        else:
            L = int(rng.integers(1, 6))
            dims = [int(rng.choice(WIDTHS)) for _ in range(L + 1)]
            act = [int(rng.integers(0, 2)) for _ in range(L - 1)] + [int(rng.random() < 0.15)]
        prec = ["f16", "f32", "bf16"][int(rng.integers(0, 3))]
        n = int(rng.choice(ROWS))
        if max(dims) > 512 and n > 20000:
            n = 4097
        rname, rflag = ROUTES[int(rng.integers(0, len(ROUTES)))]
        yield dict(c=c, dims=list(dims), act=list(act), prec=prec, n=n, rname=rname, rflag=rflag,
                   t_in=bool(dims[0] == 7 and rng.random() < 0.6), t_out=bool(rng.random() < 0.5), f64_in=bool(rng.random() < 0.3),
                   dev=bool(rng.random() < 0.4), pad_x=int(rng.choice([0, 0, 1, 9])), pad_y=int(rng.choice([0, 0, 3, 61])),
                   std=float(rng.uniform(0.5, 40.0)), data_seed=int(rng.integers(0, 1 << 30)))


def tag_of(k):
    ldx, ldy = k["dims"][0] + k["pad_x"], k["dims"][-1] + k["pad_y"]
    return "case %3d %-40s act %-20s %-4s n %-6d %-8s in_t %d out_t %d %s %s" % (
        k["c"], k["dims"], k["act"], k["prec"], k["n"], k["rname"], k["t_in"], k["t_out"], "f64" if k["f64_in"] else "f32",
        ("dev ldx %d ldy %d" % (ldx, ldy)) if k["dev"] else "host")


_PS = {}


def _param_stats():
    if not _PS:
        synth = importlib.import_module("21cmvae_amd.synth")
        pp = importlib.import_module("21cmvae_amd.preprocess")
        _PS["train"] = synth.make_params(2000, seed=1, corners=True)
        _PS["ps"] = pp.ParamStats.of(_PS["train"])
    return _PS["train"], _PS["ps"]


def run_case(ctx, k):
    """-> ("OK" | "BAD" | "refused", message)"""
    native = importlib.import_module("21cmvae_amd._native")
    synth = importlib.import_module("21cmvae_amd.synth")
    pp = importlib.import_module("21cmvae_amd.preprocess")
    from oracle import ref_numpy as ora
    par_train, ps = _param_stats()
    dims, act, prec, n, t_in, t_out, f64_in, dev = k["dims"], k["act"], k["prec"], k["n"], k["t_in"], k["t_out"], k["f64_in"], k["dev"]
    rng = np.random.default_rng(k["data_seed"])
    Ws, bs = ora.init_mlp(dims, seed=1000 + k["c"])
    bs = [rng.normal(scale=0.1, size=b.shape).astype(np.float32) for b in bs]
    if t_in:
        x = synth.make_params(n, seed=k["c"], dtype=np.float64 if f64_in else np.float32)
    else:
        x = rng.uniform(-1, 1, size=(n, dims[0])).astype(np.float64 if f64_in else np.float32)
    std = k["std"]; mean = rng.normal(scale=10.0, size=dims[-1]).astype(np.float32)
    ldx, ldy = dims[0] + k["pad_x"], dims[-1] + k["pad_y"]

    def oracle(xin):  # float64 (the transforms as the reference applies them: preprocess.py par_transform / unpreproc)
        h = pp.par_transform(xin, par_train).astype(np.float64) if t_in else xin.astype(np.float64)
        for W_, b_, a_ in zip(Ws, bs, act):
            h = h @ W_.astype(np.float64) + b_.astype(np.float64)
            h = np.maximum(h, 0) if a_ else h
        return h * std + mean.astype(np.float64) if t_out else h
    ref = oracle(x)
    flags = k["rflag"] | (native.FWD_IN_TRANSFORM if t_in else 0) | (native.FWD_OUT_TRANSFORM if t_out else 0)
    bad_pad = False
    try:
        st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
        if t_in:
            st.set_input_transform(ps.log_mask, ps.zero_floor, ps.lo, ps.hi)
        if t_out:
            st.set_output_transform(std, mean)
        if k["rname"] == "jit":
            st.jit(prec)
        outs = []
        for rep in range(2):
            if dev:
                xs = np.zeros((n, ldx), np.float32); xs[:, :dims[0]] = x   # (device rows are float32)
                d_x, d_y = ctx.malloc(xs.nbytes), ctx.malloc(n * ldy * 4)
                ctx.h2d(d_x, xs); ctx.memset(d_y, 0xFF, n * ldy * 4)
                try:
                    st.forward_dev(d_x, ldx, n, d_y, ldy, prec, flags)
                    ys = np.empty((n, ldy), np.float32); ctx.d2h(ys, d_y)
                finally:
                    ctx.free(d_x); ctx.free(d_y)
                pad = ys[:, dims[-1]:].view(np.uint32)
                bad_pad = bad_pad or bool(pad.size and not (pad == 0xFFFFFFFF).all())
                outs.append(ys[:, :dims[-1]].copy())
            else:
                outs.append(np.array(st.forward(x, prec, flags)))
    except native.EngineError as e:
        return "refused", str(e)[:110]
    y = outs[0]
    if dev and f64_in and t_in:   # the device rows were rounded to float32 before the transform: compare against that
        ref = oracle(x.astype(np.float32))
    scale = max(1.0, np.abs(ref).max())
    tol = {"f32": 3e-5, "f16": 4e-3, "bf16": 4e-2}[prec] * scale
    err = np.abs(y - ref).max() if np.isfinite(y).all() else np.inf
    same = np.array_equal(outs[0], outs[1])
    ok = err <= tol and same and not bad_pad
    extra = " WROTE PAST THE ROW'S %d OUTPUTS (stride %d)" % (dims[-1], ldy) if bad_pad else ""
    if not ok and np.isfinite(err):
        rows = np.flatnonzero((np.abs(y - ref) > tol).any(1)); cols = np.flatnonzero((np.abs(y - ref) > tol).any(0))
        extra += " rows %s.. (%d) cols %s.. (%d)" % (rows[:6], len(rows), cols[:6], len(cols))
    return ("OK" if ok else "BAD"), "err %.3g (tol %.3g) repeat identical %s route %s%s" % (err, tol, same, st.last_route()[0], extra)


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    DRY = os.environ.get("FUZZ_DRY") == "1"
    ctx = None if DRY else importlib.import_module("21cmvae_amd._native").Context.default()
    bad = refused = 0
    for k in gen_cases(cases, seed):
        if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != k["c"]:
            continue
        print(tag_of(k), "...", flush=True)
        if DRY:
            continue
        status, msg = run_case(ctx, k)
        bad += status == "BAD"; refused += status == "refused"
        print(tag_of(k), status, msg, flush=True)
    print("cases %d: refused %d, BAD %d" % (cases, refused, bad))
