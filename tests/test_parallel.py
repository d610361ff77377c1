"""CPU, world_size 2 over gloo: the data-parallel path is equivalent to one process
training on the whole global batch (sharding rule, 1/B_global scaling, flat all-reduce,
identical Adam on every rank, all-reduced loss numerator, unique-id broadcast)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


def test_shard_bounds_partition():
    par = pkg("parallel")
    for world in (1, 2, 3, 8):
        for rows in (1, 5, 255, 256, 4097):
            segs = [par.shard_bounds(1000, rows, r, world) for r in range(world)]
            assert segs[0][0] == 1000 and segs[-1][1] == 1000 + rows
            for a, b in zip(segs, segs[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in segs]
            assert max(sizes) - min(sizes) <= 1
    plan = par.epoch_plan(100, 32, 1, 2)
    assert [p[2] for p in plan] == [32, 32, 32, 4] and plan[-1][:2] == (98, 100)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch.distributed as dist
    from oracle import ref_numpy as ora
    par = importlib.import_module("21cmvae_amd.parallel")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = bytes(range(128)) if rank == 0 else b""
        got = par.broadcast_bytes(uid, 128)
        dims = [7, 16, 12, 9]
        Ws, bs = ora.init_mlp(dims, seed=2, dtype=np.float64)
        rng = np.random.default_rng(5)
        n, batch = 75, 32
        x = rng.normal(size=(n, 7)); y = rng.normal(size=(n, 9)); w = rng.uniform(0.5, 1.5, size=n)
        st = ora.AdamState(ora.flatten_params(Ws, bs).size, dtype=np.float64, lr=1e-2)
        perm = ora.epoch_permutation(n, 3, 0)
        epoch_num = 0.0
        for lo, hi, rows in par.epoch_plan(n, batch, rank, world):
            idx = perm[lo:hi]
            if len(idx):
                acts = ora.mlp_forward(Ws, bs, x[idx], keep=True)
                # local rows, but the loss gradient carries 1/B_GLOBAL
                _, g = ora.batch_loss_and_grad(acts[-1], y[idx], w[idx], denom=rows)
                dWs, dbs, _ = ora.mlp_backward(Ws, acts, g)
                gflat = ora.flatten_params(dWs, dbs)
                num = float(np.sum(ora.per_sample_loss(acts[-1], y[idx], w[idx])))
            else:
                gflat, num = np.zeros(st.m.size), 0.0
            red = par.allreduce_flat(np.concatenate([gflat, [num]]))  # grads + loss numerator, one message
            flat = ora.adam_step(ora.flatten_params(Ws, bs), red[:-1].astype(np.float64), st)
            Ws, bs = ora.unflatten_params(flat, dims)
            epoch_num += float(red[-1])
        q.put((rank, got, ora.flatten_params(Ws, bs), epoch_num / n))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    from oracle import ref_numpy as ora
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full global batches
    dims = [7, 16, 12, 9]
    Ws, bs = ora.init_mlp(dims, seed=2, dtype=np.float64)
    rng = np.random.default_rng(5)
    n, batch = 75, 32
    x = rng.normal(size=(n, 7)); y = rng.normal(size=(n, 9)); w = rng.uniform(0.5, 1.5, size=n)
    st = ora.AdamState(ora.flatten_params(Ws, bs).size, dtype=np.float64, lr=1e-2)
    Ws, bs, hist = ora.fit(Ws, bs, st, x, y, w, epochs=1, batch=batch, seed=3, dtype=np.float64)
    ref = ora.flatten_params(Ws, bs)
    assert res[0][1] == res[1][1] == bytes(range(128))          # unique-id broadcast
    np.testing.assert_array_equal(res[0][2], res[1][2])          # replicas stay identical
    np.testing.assert_allclose(res[0][2], ref, rtol=1e-5, atol=1e-7)  # gloo sums in f32
    assert abs(res[0][3] - hist["loss"][0]) / hist["loss"][0] < 1e-5


def test_every_row_is_used_exactly_once_per_epoch():
    """The C++ epoch driver's sharding rule (parallel.epoch_plan / shard_bounds) over a shared permutation: the
    union of all ranks' index sets is the whole permutation, without overlap, for ragged sizes."""
    par = pkg("parallel")
    for n, batch, world in ((75, 32, 2), (1000, 256, 3), (24562, 4096 * 8, 8), (7, 32, 4)):
        perm = np.random.default_rng(n).permutation(n)
        seen = np.concatenate([perm[lo:hi] for r in range(world) for lo, hi, _ in par.epoch_plan(n, batch, r, world)])
        assert seen.size == n and np.array_equal(np.sort(seen), np.arange(n))


def _sharded_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch
    import torch.distributed as dist
    from oracle import ref_numpy as ora
    par = importlib.import_module("21cmvae_amd.parallel")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dims = [7, 16, 12, 9]
        # ranks start DIFFERENT on purpose; rank 0's copy must win (what engine.Model.fit does before epoch 0)
        Ws, bs = ora.init_mlp(dims, seed=2 + rank, dtype=np.float64)
        flat = par.broadcast_array(ora.flatten_params(Ws, bs))
        Ws, bs = ora.unflatten_params(flat, dims)
        rng = np.random.default_rng(5)
        n, batch = 75, 32
        x = rng.normal(size=(n, 7)); y = rng.normal(size=(n, 9)); w = rng.uniform(0.5, 1.5, size=n)
        P = flat.size
        S = (P + 1 + world - 1) // world          # elements per rank, the loss numerator rides in slot P
        st = ora.AdamState(P, dtype=np.float64, lr=1e-2)
        own = np.random.default_rng(100 + rank).permutation(n).astype(np.int32) if rank else ora.epoch_permutation(n, 3, 0)
        perm = par.broadcast_array(own)  # rank 0's shuffle, everywhere
        epoch_num = 0.0
        for lo, hi, rows in par.epoch_plan(n, batch, rank, world):
            idx = perm[lo:hi]
            buf = np.zeros(S * world)
            if len(idx):
                acts = ora.mlp_forward(Ws, bs, x[idx], keep=True)
                _, g = ora.batch_loss_and_grad(acts[-1], y[idx], w[idx], denom=rows)
                dWs, dbs, _ = ora.mlp_backward(Ws, acts, g)
                buf[:P] = ora.flatten_params(dWs, dbs)
                buf[P] = float(np.sum(ora.per_sample_loss(acts[-1], y[idx], w[idx])))
            # reduce-scatter: this rank ends up with the sums of ITS slice only
            t = torch.from_numpy(buf.copy()); dist.all_reduce(t)
            mine = t.numpy()[rank * S:(rank + 1) * S]
            wflat = ora.flatten_params(Ws, bs)
            lo_, hi_ = min(P, rank * S), min(P, rank * S + S)
            # Adam on the slice: a state object that only ever sees its own elements
            st.t += 1
            alpha = ora.adam_alpha(st.lr, st.beta1, st.beta2, st.t, np.float64)
            gsl = mine[:hi_ - lo_]
            st.m[lo_:hi_] += (gsl - st.m[lo_:hi_]) * (1 - st.beta1)
            st.v[lo_:hi_] += (gsl * gsl - st.v[lo_:hi_]) * (1 - st.beta2)
            new = np.zeros(S * world)
            new[lo_:hi_] = wflat[lo_:hi_] - (st.m[lo_:hi_] * alpha) / (np.sqrt(st.v[lo_:hi_]) + st.eps)
            if P // S == rank:
                new[P] = mine[P - rank * S]      # the loss numerator travels with the weights
            parts = [torch.empty(S, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(new[rank * S:(rank + 1) * S].copy()))
            full = torch.cat(parts).numpy()
            Ws, bs = ora.unflatten_params(full[:P], dims)
            epoch_num += float(full[P])
        q.put((rank, ora.flatten_params(Ws, bs), epoch_num / n))
    finally:
        dist.destroy_process_group()


def test_sharded_adam_equals_one_process():
    """reduce-scatter -> Adam on each rank's slice -> all-gather (api_trainer.hip: reduce_and_update, sharded form),
    restated with the oracle over gloo: the gathered weights and the epoch loss are the single-process result."""
    from oracle import ref_numpy as ora
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dims = [7, 16, 12, 9]
    Ws, bs = ora.init_mlp(dims, seed=2, dtype=np.float64)
    rng = np.random.default_rng(5)
    n, batch = 75, 32
    x = rng.normal(size=(n, 7)); y = rng.normal(size=(n, 9)); w = rng.uniform(0.5, 1.5, size=n)
    st = ora.AdamState(ora.flatten_params(Ws, bs).size, dtype=np.float64, lr=1e-2)
    Ws, bs, hist = ora.fit(Ws, bs, st, x, y, w, epochs=1, batch=batch, seed=3, dtype=np.float64)
    ref = ora.flatten_params(Ws, bs)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-9, atol=1e-12)
    assert abs(res[0][2] - hist["loss"][0]) / hist["loss"][0] < 1e-9
