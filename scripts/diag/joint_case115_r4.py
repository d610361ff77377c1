"""Diagnostics (r5): r4's joint_fuzz.py kept VERBATIM in its case generator (one random stream for shapes and data) so that
case 115 of `joint_fuzz.py 200 105` -- f32, 1,500 single-row steps per epoch, emulator loss 5.4e-5 off the separate trainer
 -- can be drawn again; run as `FUZZ_ONLY=115 python joint_case115_r4.py 200 105`: it then prints the
relative difference epoch by epoch over FOUR epochs (1,500 / 3,000 / 4,500 / 6,000 optimizer steps).
(r4) random joint steps (v21_joint_*: autoencoder + latent emulator stepping on the same rows, BASELINE
configs[2]) checked through two invariants that need no second implementation:
  (1) both models training: the autoencoder's half is BIT-IDENTICAL to the same autoencoder trained alone;
  (2) autoencoder frozen (lr 0): the emulator's epochs equal those of a separate trainer fed the float64 oracle's
      latents of the (unchanged) encoder, to the precision's tolerance;
and a twin joint object bit for bit.   python joint_fuzz.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
native = importlib.import_module("21cmvae_amd._native")
from oracle import ref_numpy as ora
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DRY = os.environ.get("FUZZ_DRY") == "1"
ctx = None if DRY else native.Context.default()
HID = [8, 16, 17, 32, 33, 64, 96, 128, 224, 288, 352, 400, 512]
bad = 0
for c in range(cases):
    D = int(rng.choice([33, 100, 451]))
    lat = int(rng.choice([1, 4, 9, 12, 16, 32]))
    enc = [int(rng.choice(HID)) for _ in range(int(rng.integers(0, 3)))]
    dec = [int(rng.choice(HID)) for _ in range(int(rng.integers(0, 3)))]
    ae_dims = [D] + enc + [lat] + dec + [D]
    ae_act = [1] * len(enc) + [0] + [1] * len(dec) + [0]
    gl = len(enc)                                              # the encoder's linear output layer
    em_dims = [7] + [int(rng.choice(HID)) for _ in range(int(rng.integers(1, 5)))] + [lat]
    em_act = [1] * (len(em_dims) - 2) + [0]
    prec = ["f16", "f32", "bf16"][int(rng.integers(0, 3))]
    n = int(rng.choice([40, 256, 300, 700, 1500]))
    batch = min(n, int(rng.choice([1, 32, 100, 128, 256, 257, 600, 1024, 2048])))
    perm = rng.permutation(n).astype(np.int32) if rng.random() < 0.6 else None
    tag = "case %3d %-4s ae %-34s em %-28s n %-5d batch %-5d %s" % (c, prec, ae_dims, em_dims, n, batch, "perm" if perm is not None else "seq ")
    x = rng.uniform(-1, 1, size=(n, D)).astype(np.float32)
    par = rng.uniform(-1, 1, size=(n, 7)).astype(np.float32)
    wa = (rng.uniform(0.5, 1.5, size=n) / D).astype(np.float32)
    Wa, ba = ora.init_mlp(ae_dims, seed=300 + c)
    We, be = ora.init_mlp(em_dims, seed=600 + c)
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != c:
        continue
    print(tag, "...", flush=True)
    if DRY:
        continue
    h = x.astype(np.float64)
    for W_, b_, a_ in list(zip(Wa, ba, ae_act))[:gl + 1]:
        h = h @ W_.astype(np.float64) + b_.astype(np.float64)
        h = np.maximum(h, 0) if a_ else h
    z = h
    wz = ora.mse_row_weight(z.astype(np.float32)).astype(np.float32)

    def trainer(dims, act, Ws, bs, lr):
        st = native.Stack(ctx, dims, act); st.set_weights(ora.flatten_params(Ws, bs))
        tr = native.Trainer(st, prec, batch); tr.set_adam(lr=lr)
        return st, tr
    try:
        # (2) frozen encoder against a separate trainer on the oracle's latents
        sta, tra = trainer(ae_dims, ae_act, Wa, ba, 0.0)
        ste, tre = trainer(em_dims, em_act, We, be, 1e-3)
        st2, tr2 = trainer(em_dims, em_act, We, be, 1e-3)
        tra.set_data(0, x, None, wa)
        tre.set_data(0, par, np.zeros((n, lat), np.float32), wz)
        tr2.set_data(0, par, z.astype(np.float32), wz)
        joint = native.Joint(tra, tre, latent_layer=gl)
        NE = 4 if os.environ.get("FUZZ_ONLY") else 2
        lj = [joint.run_epoch(perm, batch)[1] for _ in range(NE)]
        l2 = [tr2.run_epoch(perm, batch) for _ in range(NE)]
        if os.environ.get("FUZZ_ONLY"):
            print("epoch losses, joint    :", lj)
            print("epoch losses, separate :", l2)
            print("relative difference by epoch (%d steps each):" % (-(-n // batch)), [abs(a - b) / abs(b) for a, b in zip(lj, l2)])
            print("emulator weights after %d steps: max |joint - separate| = %.3e of max |w| %.3e" % (
                NE * -(-n // batch), float(np.abs(ste.get_weights() - st2.get_weights()).max()), float(np.abs(st2.get_weights()).max())))
        frozen_ok = np.array_equal(sta.get_weights(), ora.flatten_params(Wa, ba))
        # (1) both training against the autoencoder alone; twin joint
        res = []
        for _ in range(2):
            a_st, a_tr = trainer(ae_dims, ae_act, Wa, ba, 1e-3)
            e_st, e_tr = trainer(em_dims, em_act, We, be, 1e-3)
            a_tr.set_data(0, x, None, wa); e_tr.set_data(0, par, np.zeros((n, lat), np.float32), wz)
            jj = native.Joint(a_tr, e_tr, latent_layer=gl)
            losses = [jj.run_epoch(perm, batch) for _ in range(2)]
            res.append((losses, a_st.get_weights(), e_st.get_weights()))
        s_st, s_tr = trainer(ae_dims, ae_act, Wa, ba, 1e-3)
        s_tr.set_data(0, x, None, wa)
        ls = [s_tr.run_epoch(perm, batch) for _ in range(2)]
    except native.EngineError as e:
        print(tag, "refused:", str(e)[:120], flush=True)
        continue
    tol = {"f32": 1e-4, "f16": 5e-3, "bf16": 4e-2}[prec]   # (f32: thousands of single-row steps on fp32 latents against the float64 ones reach 5e-5)
    rel2 = max(abs(a - b) / abs(b) for a, b in zip(lj, l2))
    (lo1, wa1, we1), (lo2, wa2, we2) = res
    # (f32: the joint launch may group the rows of the batch loss differently from the single model's -- the same sums per
    #  weight, the reported loss to ~1e-7)
    alone = np.array_equal(wa1, s_st.get_weights()) and max(abs(a[0] - b) / abs(b) for a, b in zip(lo1, ls)) <= (1e-6 if prec == "f32" else 0.0)
    alone_diff = (max(abs(a[0] - b) / abs(b) for a, b in zip(lo1, ls)), float(np.abs(wa1 - s_st.get_weights()).max()))
    twin = lo1 == lo2 and np.array_equal(wa1, wa2) and np.array_equal(we1, we2)
    finite = np.isfinite(we1).all() and np.isfinite(wa1).all()
    flag = "OK " if rel2 <= tol and frozen_ok and alone and twin and finite else "BAD"
    bad += flag == "BAD"
    print(tag, flag, "frozen: emulator loss rel %.1e (tol %.0e), encoder untouched %s | autoencoder == alone %s | twin identical %s"
          % (rel2, tol, frozen_ok, alone if alone else "False (loss rel %.1e, weights max diff %.1e)" % alone_diff, twin), flush=True)
print("cases %d, BAD %d" % (cases, bad))
