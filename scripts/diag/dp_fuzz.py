"""Diagnostics (r4; r5: importable -- tests/test_fuzz_gpu.py runs a seeded slice under `pytest -m gpu`): the data-parallel path with 2-4 ranks on ONE GPU (gloo process group + the library's host-staged
collective transport: everything but the RCCL calls is the production path) on random shapes: stacks, precision, rows,
batch sizes that do NOT divide by the world size (ranks with fewer rows than others, ranks with NO rows in a partial last
batch), all-reduce or sharded Adam.  Invariants: every rank ends with bit-identical weights, losses and Adam moments,
and they agree with ONE process fed the same seed to the precision's tolerance.   python dp_fuzz.py [cases] [seed]"""
import importlib, os, socket, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def fit(cfg, world=1, rank=0, port=0):
    eng = importlib.import_module("21cmvae_amd.engine")
    native = importlib.import_module("21cmvae_amd._native")
    losses = importlib.import_module("21cmvae_amd.losses")
    optm = importlib.import_module("21cmvae_amd.optimizers")
    rng = np.random.default_rng(cfg["seed"])
    x = rng.uniform(-1, 1, size=(cfg["n"], cfg["dims"][0])).astype(np.float32)
    y = x if cfg["ae"] else rng.normal(size=(cfg["n"], cfg["dims"][-1])).astype(np.float32)
    xv = x[: cfg["nv"]]; yv = y[: cfg["nv"]]
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        importlib.import_module("21cmvae_amd.parallel").init_engine_comm(native.Context.default(), backend="host", sharded=cfg["sharded"])
    eng.set_random_seed(1000 + rank)          # different on every rank: rank 0's weights and shuffles must win
    acts = ["relu" if a else None for a in cfg["act"]]
    m = eng.Sequential([eng.Input((cfg["dims"][0],))] + [eng.Dense(u, a) for u, a in zip(cfg["dims"][1:], acts)])
    m.precision = cfg["prec"]
    m.compile(optimizer=optm.Adam(2e-3), loss=losses.mean_squared_error)
    h = m.fit(x, y, batch_size=cfg["batch"], epochs=2, validation_data=(xv, yv), verbose=0)
    out = (np.concatenate([a.ravel() for a in m.get_weights()]), h.history["loss"], h.history["val_loss"], m._trainer.get_state())
    if world > 1:
        import torch.distributed as dist
        native.Context.default().comm_destroy()
        dist.destroy_process_group()
    return out


def worker(rank, world, port, cfg, q):
    try:
        q.put((rank, fit(cfg, world, rank, port)))
    except Exception as e:
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))


HID = [8, 17, 32, 48, 100, 224, 352]


def gen_cases(cases, seed):
    rng = np.random.default_rng(seed)
    for c in range(cases):
        world = int(rng.choice([2, 3, 4]))
        L = int(rng.integers(1, 5))
        din = int(rng.choice([7, 33, 451])); ae = bool(rng.random() < 0.5)
        dims = [din] + [int(rng.choice(HID)) for _ in range(L - 1)] + [din if ae else int(rng.choice([9, 33, 451]))]
        act = [int(rng.integers(0, 2)) for _ in range(L - 1)] + [0]
        cfg = dict(dims=dims, act=act, ae=ae, prec=["f32", "f16", "bf16"][int(rng.integers(0, 3))], n=int(rng.choice([10, 61, 300, 1001])),
                   batch=int(rng.choice([1, 2, 5, 32, 100, 257])), nv=int(rng.choice([1, 7, 40])), sharded=bool(rng.random() < 0.5), seed=int(rng.integers(0, 1 << 30)))
        cfg["batch"] = min(cfg["batch"], cfg["n"]); cfg["nv"] = min(cfg["nv"], cfg["n"])
        if cfg["n"] // cfg["batch"] > 200:     # (keep a case within seconds)
            cfg["batch"] = max(cfg["batch"], cfg["n"] // 100)
        cfg["c"] = c; cfg["world"] = world
        yield cfg


def tag_of(cfg):
    return "case %3d world %d %-5s %-9s %-30s act %-12s n %-4d batch %-3d val %-2d" % (
        cfg["c"], cfg["world"], cfg["prec"], "sharded" if cfg["sharded"] else "allreduce", cfg["dims"], cfg["act"], cfg["n"], cfg["batch"], cfg["nv"])


def run_case(cfg):
    """-> ("OK" | "BAD", message).  Starts cfg["world"] processes (spawn) on the one GPU, then repeats the fit in THIS process."""
    # (the standard library's multiprocessing, not torch's: THIS process must not import torch after it has used libv21.so --
    #  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so, and a torch imported second brings a
    #  second HIP runtime into the process: `double free or corruption` at exit, r5.  The ranks import torch first.)
    import multiprocessing as mp
    world = cfg["world"]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=worker, args=(r, world, port, cfg, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    except Exception as e:
        for p in procs:
            p.kill()
        return "BAD", "a rank did not answer (%s)" % type(e).__name__
    for p in procs:
        p.join(timeout=60)
    why = [out[-600:] for _, out in res if isinstance(out, str)]
    if not why and any(p.exitcode != 0 for p in procs):
        why.append("exit codes %s" % [p.exitcode for p in procs])
    if not why:
        w0, l0, v0, s0 = res[0][1]
        for r, (w, l, v, s) in res[1:]:
            if not (np.array_equal(w, w0) and l == l0 and v == v0 and s[0] == s0[0] and np.array_equal(s[1], s0[1]) and np.array_equal(s[2], s0[2])):
                why.append("rank %d differs from rank 0 (weights max diff %.2e)" % (r, float(np.abs(w - w0).max())))
        ws, ls, vs, ss = fit(dict(cfg), 1, 0, 0)
        tol = {"f32": 5e-5, "f16": 5e-3, "bf16": 4e-2}[cfg["prec"]]
        rel = max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(l0 + v0, ls + vs))
        if not rel <= tol:
            why.append("losses differ from one process by %.2e (tol %.0e)" % (rel, tol))
        if s0[0] != ss[0]:
            why.append("optimizer steps %d against %d" % (s0[0], ss[0]))
        # weights: f32 element by element; 16-bit operands: by the distance travelled (hundreds of Adam steps on batches
        # of a few rows amplify another summation order -- every rank still ends bit-identical to the others)
        eng = importlib.import_module("21cmvae_amd.engine")
        eng.set_random_seed(1000)
        m0 = eng.Sequential([eng.Input((cfg["dims"][0],))] + [eng.Dense(u, "relu" if a else None) for u, a in zip(cfg["dims"][1:], cfg["act"])])
        wi = np.concatenate([a.ravel() for a in m0.get_weights()])
        if cfg["prec"] == "f32":
            wtol = 2e-5 * max(1e-3, float(np.abs(ws).max()))
            if not np.isfinite(w0).all() or float(np.abs(w0 - ws).max()) > wtol:
                why.append("weights differ from one process by %.2e (tol %.1e)" % (float(np.abs(w0 - ws).max()), wtol))
        else:
            relw = float(np.linalg.norm(w0 - ws) / max(1e-30, np.linalg.norm(ws - wi)))
            if not np.isfinite(w0).all() or relw > {"f16": 0.1, "bf16": 0.3}[cfg["prec"]]:
                why.append("weights: |dp - single| / |single - initial| = %.3f" % relw)
    return ("OK" if not why else "BAD"), " | ".join(why)


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for cfg in gen_cases(cases, seed):
        print(tag_of(cfg), "...", flush=True)
        status, msg = run_case(cfg)
        bad += status == "BAD"
        print(tag_of(cfg), status, msg, flush=True)
    print("cases %d, BAD %d" % (cases, bad))
