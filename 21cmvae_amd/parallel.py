"""Data-parallel plumbing: one process per GPU, launched by ``torch.distributed.run``.

The gradient all-reduce itself runs inside libv21.so on RCCL (``v21_comm_*``); this
module only (a) bootstraps that communicator by broadcasting the RCCL unique id over
``torch.distributed`` (backend ``nccl`` on GPUs, ``gloo`` in the CPU tests), (b) states the
batch-sharding rule the C++ epoch driver uses, so it can be tested without a GPU, and
(c) offers the same flat-buffer all-reduce over ``torch.distributed`` for hosts that
drive the steps from Python.
"""
import numpy as np


def shard_bounds(first, batch_rows, rank, world):
    """Rows [lo, hi) of the global batch [first, first + batch_rows) that `rank` trains on:
    contiguous, as even as possible, every row exactly once (v21_trainer_run_epoch)."""
    lo = first + batch_rows * rank // world
    hi = first + batch_rows * (rank + 1) // world
    return lo, hi


def epoch_plan(n, batch, rank, world):
    """[(lo, hi, global_rows)] for every step of an epoch, partial last batch kept."""
    out = []
    for first in range(0, n, batch):
        rows = min(batch, n - first)
        lo, hi = shard_bounds(first, rows, rank, world)
        out.append((lo, hi, rows))
    return out


def _dist():
    import torch.distributed as dist
    return dist


def _pg_device(device=None):
    """Where a tensor handed to the process group must live: the host under gloo; under nccl (= RCCL) the GPU with
    index `device` -- torch's current device is per THREAD, a worker thread starts on GPU 0 whatever the rank's GPU is,
    so callers that know their context pass its device index."""
    import torch
    if _dist().get_backend() != "nccl":
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device() if device is None else int(device))


def broadcast_bytes(payload, nbytes, src=0, device=None):
    """Rank `src` passes `payload` (bytes); every rank gets the same bytes back."""
    import torch
    dist = _dist()
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=_pg_device(device))
    if dist.get_rank() == src:
        buf.copy_(torch.tensor(list(payload), dtype=torch.uint8))
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().tolist())


def init_engine_comm(ctx, backend="auto", sharded=False, buckets=1):
    """Attach a data-parallel communicator to `ctx` for the current torch.distributed process group.

    backend "rccl": the in-library RCCL communicator (unique id broadcast over the process group) -- what a
    one-process-per-GPU run uses.  "host": the library's host-staged transport driven by the process group's own
    collectives (gloo in the CPU/1-GPU tests; under an nccl group the buffer is staged through a device tensor).  "auto": RCCL when the process group is nccl, else host.
    sharded: reduce-scatter -> Adam on this rank's slice -> all-gather instead of one all-reduce.
    buckets: 2 = the all-reduce form's exchange in two messages, the output-side layers' half overlapping the second
    weight-gradient launch (include/v21.h: v21_comm_set_buckets; r5)."""
    from . import _native
    dist = _dist()
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return
    if backend == "auto":
        backend = "rccl" if dist.get_backend() == "nccl" else "host"
    if backend == "rccl":
        uid = ctx.comm_unique_id() if rank == 0 else b""
        uid = broadcast_bytes(uid, _native.COMM_ID_BYTES, device=ctx.device)
        ctx.comm_init(world, rank, uid)
    else:
        import torch
        on_gpu = dist.get_backend() == "nccl"  # the group's tensors live on the device: stage the host buffer through it
        dev = _pg_device(ctx.device)

        def _t(buf):
            t = torch.from_numpy(buf)
            return t.to(dev) if on_gpu else t

        def allreduce(buf, n):
            t = _t(buf)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if on_gpu:
                torch.from_numpy(buf).copy_(t)

        reduce_scatter = allreduce  # gloo has no reduce_scatter: sum everything, the slices hold what they must

        def allgather(buf, n_per):
            t = _t(buf)
            parts = [torch.empty(n_per, dtype=torch.float32, device=t.device) for _ in range(world)]
            dist.all_gather(parts, t[rank * n_per:(rank + 1) * n_per].clone())
            out = torch.from_numpy(buf)
            for r, part in enumerate(parts):
                out[r * n_per:(r + 1) * n_per] = part.cpu() if on_gpu else part
        ctx.comm_init_host(world, rank, allreduce, reduce_scatter, allgather)
    ctx.comm_set_sharded(sharded)
    ctx.comm_set_buckets(buckets)


def broadcast_array(arr, src=0, device=None):
    """Every rank returns rank `src`'s copy of the numpy array (same shape and dtype everywhere)."""
    import torch
    dist = _dist()
    a = np.ascontiguousarray(arr)
    t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(_pg_device(device))
    dist.broadcast(t, src=src)
    return t.cpu().numpy().view(a.dtype).reshape(a.shape)


def in_group():
    """True inside an initialised torch.distributed process group of more than one rank."""
    try:
        dist = _dist()
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:
        return False


def allreduce_flat(arr, device=None):
    """Sum a flat float32 numpy buffer over the process group (CPU tests / Python-driven
    loops); returns a new array."""
    import torch
    dist = _dist()
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32).copy()).to(_pg_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
