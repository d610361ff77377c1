"""The library's data-parallel path with MORE THAN ONE RANK: two processes, each with its own context on the one
GPU of the test box, a gloo process group, and the library's host-staged collective transport
(v21_comm_init_host).  Everything except the RCCL calls themselves is the production path: per-rank sharding of
every global batch in v21_trainer_run_epoch, the 1/B_global scaling, the loss numerator riding in the gradient
arena, all-reduce + replicated Adam or reduce-scatter + sharded Adam + all-gather, and engine.Model.fit's
broadcast of weights / optimizer state / shuffle.  No multi-rank RCCL run has happened on hardware (the driver's
8-GPU node is the first)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

DIMS, ACT = [451, 48, 9, 24, 451], ["relu", None, "relu", None]
# the reference's autoencoder at full width: the stack csrc/fused_train.h is compiled for (archs.h: T1)
DIMS_T1, ACT_T1 = [451, 352, 9, 32, 352, 451], ["relu", None, "relu", "relu", None]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data():
    synth = pkg("synth")
    sig = synth.make_signals(300, seed=4)
    sv = synth.make_signals(60, seed=5)
    pp = pkg("preprocess")
    return pp.preproc(sig, sig).astype(np.float32), pp.preproc(sv, sig).astype(np.float32), sig


def _fit(prec, seed, world=1, rank=0, sharded=False, port=0, fused=False, buckets=1, batch=128, rows=300):
    import importlib
    if fused:  # every step through a fused training kernel: the 16-rows-per-wave one (what a trainer of this size takes), or "32": the 128-row one
        os.environ["V21_FUSED_TRAIN_ROWS"] = "1"
        os.environ["V21_FUSED_TRAIN16"] = "0" if fused == "32" else "1"
    else:
        os.environ.pop("V21_FUSED_TRAIN_ROWS", None)
        os.environ.pop("V21_FUSED_TRAIN16", None)
    dims, acts = (DIMS_T1, ACT_T1) if fused else (DIMS, ACT)
    sys.path.insert(0, ROOT)
    eng = importlib.import_module("21cmvae_amd.engine")
    native = importlib.import_module("21cmvae_amd._native")
    losses = importlib.import_module("21cmvae_amd.losses")
    optm = importlib.import_module("21cmvae_amd.optimizers")
    y, yv, sig = _data()
    if rows != 300:   # (the bucket tests: steps of > 1,023 rows per rank take the split-K weight-gradient kernel and its slab sums)
        sig = pkg("synth").make_signals(rows, seed=4)
        y = pkg("preprocess").preproc(sig, sig).astype(np.float32)
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        importlib.import_module("21cmvae_amd.parallel").init_engine_comm(native.Context.default(), backend="host", sharded=sharded, buckets=buckets)
    eng.set_random_seed(seed)  # different on every rank: rank 0's weights and shuffles must win
    m = eng.Sequential([eng.Input((451,))] + [eng.Dense(u, a) for u, a in zip(dims[1:], acts)])
    m.precision = prec
    m.compile(optimizer=optm.Adam(2e-3), loss=losses.relative_mse_loss(sig))
    h = m.fit(y, y, batch_size=batch, epochs=3, validation_data=(yv, yv), verbose=0)   # (default: 128 + 128 + 44 rows per epoch)
    out = (np.concatenate([a.ravel() for a in m.get_weights()]), h.history["loss"], h.history["val_loss"],
           m._trainer.get_state())
    if world > 1:
        import torch.distributed as dist
        native.Context.default().comm_destroy()
        dist.destroy_process_group()
    os.environ.pop("V21_FUSED_TRAIN_ROWS", None)
    os.environ.pop("V21_FUSED_TRAIN16", None)
    return out


def _worker(rank, world, port, prec, sharded, q, fused=False, buckets=1, batch=128, rows=300):
    try:
        q.put((rank, _fit(prec, seed=100 + rank, world=world, rank=rank, sharded=sharded, port=port, fused=fused, buckets=buckets, batch=batch, rows=rows)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))


@pytest.mark.parametrize("prec,sharded,fused", [("f32", False, False), ("f32", True, False), ("f16", True, False), ("f16", False, True), ("f16", True, "32")],
                         ids=["f32-allreduce", "f32-sharded", "f16-sharded", "f16-allreduce-fused_train16", "f16-sharded-fused_train"])
def test_two_ranks_on_one_gpu_equal_one_process(prec, sharded, fused):
    """(the last case: every rank's share of every batch goes through the fused training kernel, csrc/fused_train.h, then
    the split-K weight gradients, the exchange and Adam -- the route of large data-parallel steps)"""
    world, port = 2, _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, prec, sharded, q, fused)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, out in res:
        assert not isinstance(out, str), out
    (w0, l0, v0, s0), (w1, l1, v1, s1) = res[0][1], res[1][1]
    np.testing.assert_array_equal(w0, w1)                      # the replicas end identical ...
    assert l0 == l1 and v0 == v1                               # ... and saw the same losses (callbacks decide alike)
    assert s0[0] == s1[0] == 9
    np.testing.assert_array_equal(s0[1], s1[1])                # gathered Adam moments
    ws, ls, vs, ss = _fit(prec, seed=100, fused=fused)         # one process, rank 0's seed
    tol = 2e-5 if prec == "f32" else 2e-3                      # the split batch sums in another order (and in f16 operands)
    assert max(abs(a - b) / b for a, b in zip(l0, ls)) < tol, (l0, ls)
    assert max(abs(a - b) / b for a, b in zip(v0, vs)) < tol
    d1, ds = w0 - _init_weights(fused), ws - _init_weights(fused)
    cos = float(d1 @ ds / (np.linalg.norm(d1) * np.linalg.norm(ds)))
    assert cos > (0.99999 if prec == "f32" else 0.999), cos
    # (f16: the two halves of a batch meet other waves and other 16-bit roundings than the whole batch in one process;
    #  with the 16-rows-per-wave kernel 5 of 333,420 first moments came to 2.5e-3 of the largest)
    np.testing.assert_allclose(s0[1], ss[1], rtol=0, atol=(1e-5 if prec == "f32" else 4e-3) * np.abs(ss[1]).max())


def _run_world(world, prec, fused, buckets, batch, rows):
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, prec, False, q, fused, buckets, batch, rows)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, out in res:
        assert not isinstance(out, str), out
    return [out for _, out in res]


@pytest.mark.parametrize("case", ["chain_small_steps", "chain_split_k_slabs", "fused_train16"])
def test_two_gradient_buckets_equal_one_message_bit_for_bit(case):
    """r5: v21_comm_set_buckets(ctx, 2) forms the weight gradients in two launches, the output-side layers
    first, and all-reduces their half of the arena (with the loss slot) while the input-side half is still being formed.
    The same workgroups do the same sums; with two ranks a + b = b + a: weights, losses and Adam moments after three epochs
    are BIT-IDENTICAL to the one-message form, on every rank -- small steps (gradients straight into the arena), steps whose
    shares of 1,152 rows take the split-K kernel + the per-bucket slab sums, and shares through the fused training kernel."""
    prec = "f16"
    fused, batch, rows = {"chain_small_steps": (False, 128, 300), "chain_split_k_slabs": (False, 2304, 2304 + 700),
                          "fused_train16": (True, 128, 300)}[case]
    one = _run_world(2, prec, fused, 1, batch, rows)
    two = _run_world(2, prec, fused, 2, batch, rows)
    for (w1, l1, v1, s1), (w2, l2, v2, s2) in zip(one, two):
        np.testing.assert_array_equal(w1, w2)
        assert l1 == l2 and v1 == v2 and s1[0] == s2[0]
        np.testing.assert_array_equal(s1[1], s2[1]); np.testing.assert_array_equal(s1[2], s2[2])
    np.testing.assert_array_equal(two[0][0], two[1][0])        # and the two ranks agree with each other


def test_null_transport_takes_the_multi_rank_step_structure_on_one_gpu(ctx):
    """r5: v21_comm_init_null -- `nranks` ranks, nothing exchanged: this rank's share of the global batch through the N > 1
    route (chain -> split-K weight gradients -> [no exchange] -> Adam), for bench.py's dp_compute_only leg.  The gradient is
    this share's contribution to the global-batch mean: the single-rank gradient of the same rows times rows / global rows."""
    native = pkg("_native")
    from oracle import ref_numpy as ora
    dims, act = DIMS_T1, [1, 0, 1, 1, 0]
    y, _, sig = _data()
    w = ora.relative_mse_row_weight(y, sig).astype(np.float32)
    Ws, bs = ora.init_mlp(dims, seed=3)
    flat = ora.flatten_params(Ws, bs)
    c2 = native.Context(0)
    c2.comm_init_null(4, 0)
    assert c2.comm_info() == (4, 0, "null")
    res = {}
    for name, c, batch in (("null", c2, 300), ("single", ctx, 75)):
        st = native.Stack(c, dims, act); st.set_weights(flat)
        tr = native.Trainer(st, "f16", 300); tr.set_adam(lr=0.0)
        tr.set_data(0, y[:300] if name == "null" else y[:75], None, w[:300] if name == "null" else w[:75])
        tr.run_epoch(None, batch)          # null: rank 0 of 4 takes rows [0, 75) of the 300-row global batch
        g, route = tr.get_grad(), tr.last_route()[0]
        res[name] = (g, route, tr.phase_profile(lambda: tr.run_epoch(None, batch), steps=8))   # (lr = 0: the weights stay put)
    c2.comm_destroy()
    assert res["null"][1] == ("chain16", "dw16_splitk") and res["single"][1] == ("chain16", "dw16_adam")
    g0, g1 = res["null"][0], res["single"][0] * (75.0 / 300.0)
    assert float(g0 @ g1 / (np.linalg.norm(g0) * np.linalg.norm(g1))) > 0.99999 and abs(np.linalg.norm(g0) / np.linalg.norm(g1) - 1) < 1e-3
    ph, ph1 = res["null"][2], res["single"][2]
    assert ph["stamped_step_us"] > 0 and ph["weight_gradients_us"] > 1.0 and ph["forward_and_activation_gradients_us"] > 1.0
    assert abs(ph["exchange_exposed_us"]) < 3.0                # nothing is exchanged
    # one launch for gradients + Adam: reported under the Adam phase (the cut points coincide: differences of two stamped runs)
    assert abs(ph1["weight_gradients_us"]) < 3.0 and ph1["adam_and_repack_us"] > 1.0


def _fit_joint(seed, world=1, rank=0, port=0, prec="f16"):
    """AutoEncoderEmulator.train(joint=True) (BASELINE configs[2]) on one rank or as one of `world` ranks."""
    import importlib
    sys.path.insert(0, ROOT)
    eng = importlib.import_module("21cmvae_amd.engine")
    native = importlib.import_module("21cmvae_amd._native")
    emu = importlib.import_module("21cmvae_amd.emulator")
    optm = importlib.import_module("21cmvae_amd.optimizers")
    synth = importlib.import_module("21cmvae_amd.synth")
    data = synth.make_dataset(300, 60, 20, seed=4)
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        importlib.import_module("21cmvae_amd.parallel").init_engine_comm(native.Context.default(), backend="host")
    eng.set_random_seed(seed)  # different on every rank: rank 0's weights and shuffles must win
    ae = emu.AutoEncoderEmulator(precision=prec, enc_hidden_dims=[48], dec_hidden_dims=[24, 48], em_hidden_dims=[40, 40], **data)
    ae.autoencoder.compile(optimizer=optm.Adam(2e-3), loss=emu.relative_mse_loss(ae.signal_train))
    ae.emulator.compile(optimizer=optm.Adam(2e-3), loss=emu.mean_squared_error)
    out = ae.train(epochs=3, verbose=0, joint=True, batch_size=128)   # 128 + 128 + 44 rows per epoch
    res = (np.concatenate([a.ravel() for a in ae.autoencoder.get_weights()]),
           np.concatenate([a.ravel() for a in ae.emulator.get_weights()]), [list(h) for h in out])
    if world > 1:
        import torch.distributed as dist
        native.Context.default().comm_destroy()
        dist.destroy_process_group()
    return res


def _worker_joint(rank, world, port, q, prec="f16"):
    try:
        q.put((rank, _fit_joint(seed=200 + rank, world=world, rank=rank, port=port, prec=prec)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))


@pytest.mark.parametrize("prec", ["f16", "f32"])
def test_joint_step_two_ranks_equal_one_process(prec):
    """v21_joint_run_epoch with a communicator: every rank carries its share of every batch through both models, the
    two gradient arenas are summed over the ranks (one all-reduce each), Adam is replicated.  Two ranks on the one GPU
    of the test box against one process with rank 0's seed.  f16, and f32 (the joint chain launch of
    train_chain32s.h, then each model's sliced gradient launches, the exchange and Adam)."""
    world, port = 2, _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker_joint, args=(r, world, port, q, prec)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, out in res:
        assert not isinstance(out, str), out
    (wa0, we0, h0), (wa1, we1, h1) = res[0][1], res[1][1]
    np.testing.assert_array_equal(wa0, wa1)
    np.testing.assert_array_equal(we0, we1)
    assert h0 == h1
    was, wes, hs = _fit_joint(seed=200, prec=prec)
    for a, b in zip(h0, hs):
        assert len(a) == len(b) == 3
        assert max(abs(x - y) / y for x, y in zip(a, b)) < (5e-3 if prec == "f16" else 1e-4), (a, b)


def _init_weights(fused=False):
    eng = pkg("engine")
    eng.set_random_seed(100)
    dims, acts = (DIMS_T1, ACT_T1) if fused else (DIMS, ACT)
    m = eng.Sequential([eng.Input((451,))] + [eng.Dense(u, a) for u, a in zip(dims[1:], acts)])
    return np.concatenate([a.ravel() for a in m.get_weights()])


def test_collectives_single_rank_identity(ctx):
    """One rank: every collective of the C ABI is the identity, sharded mode included (what a 1-GPU run of the
    data-parallel bench leg exercises)."""
    buf = np.arange(40, dtype=np.float32)
    d = ctx.malloc(buf.nbytes)
    ctx.h2d(d, buf)
    ctx.allreduce(d, 40); ctx.reduce_scatter(d, 40); ctx.allgather(d, 40)
    out = np.empty_like(buf)
    ctx.d2h(out, d)
    np.testing.assert_array_equal(out, buf)
    ctx.free(d)


def test_predict_on_several_devices_equals_one(ctx):
    """predict(devices=[...]): rows cut into blocks, one replica per list entry, evaluated concurrently, put back
    in order -- bit-identical to the single-stack result (two replicas on the one GPU of the test box)."""
    synth, emu = pkg("synth"), pkg("emulator")
    data = synth.make_dataset(600, 80, 50)
    em = emu.DirectEmulator(hidden_dims=[64, 96], **data)
    par = synth.make_params(1001, seed=9)
    one = em.predict(par)
    two = em.predict(par, devices=[0, 0])
    np.testing.assert_array_equal(one, two)
    three = em.predict(par[:7], devices=[0, 0, 0])
    np.testing.assert_array_equal(one[:7], three)
    em.emulator.set_weights([w * 1.5 for w in em.emulator.get_weights()])   # replicas must follow weight changes
    np.testing.assert_array_equal(em.predict(par), em.predict(par, devices=[0, 0]))
    assert not np.array_equal(one, em.predict(par, devices=[0, 0]))
