"""Data-parallel helpers: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Semantics = the reference's in-graph towers (cifar10/gan_resnet.py:183-192, 529-552, 697, 786): the global
batch is split contiguously, every tower computes its loss on its own shard (own batch-norm statistics),
the cost is the MEAN of the tower costs -- so the gradient is the mean of the per-rank gradients.  One
all-reduce(sum) per optimiser step on the flat fp32 gradient slab; the 1/world factor is applied inside
the Adam kernel (``grad_scale``)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_rows(a, rank, world):
    """rank r owns rows [r*B/N, (r+1)*B/N) of every fed array (tf.split(axis=0), gan_resnet.py:529-538)."""
    a = np.asarray(a)
    n = a.shape[0]
    if n % world:
        raise ValueError("batch %d is not divisible by %d ranks" % (n, world))
    k = n // world
    return a[rank * k:(rank + 1) * k]


def allreduce_sum_(flat, stream=None):
    """In-place sum over ranks of a flat gradient slab (no-op for a single rank)."""
    if not dist.is_available() or not dist.is_initialized():
        return flat
    if stream is not None:
        with torch.cuda.stream(stream):
            dist.all_reduce(flat)
    else:
        dist.all_reduce(flat)
    return flat


def init_comm(ctx, world, rank, stub=False):
    """Create the context's in-ABI communicator (rcgan_comm_init: RCCL over xGMI, one rank per process / GPU).  The 128-byte RCCL
    unique id of rank 0 reaches the other ranks through torch.distributed when a process group exists, else through a TCP store at
    MASTER_ADDR:MASTER_PORT+1 (the launcher's environment).  stub: the single-process test double (rcgan_comm_init_stub)."""
    import ctypes as C
    import os
    if stub:
        ctx.check(ctx.lib.rcgan_comm_init_stub(ctx.h, int(world)))
        return
    buf = C.create_string_buffer(128)
    if rank == 0:
        rc = ctx.lib.rcgan_comm_unique_id(buf)
        if rc != 0:
            from ._lib import RcganError
            raise RcganError(rc, "rcgan_comm_unique_id failed (is librccl.so loadable?)")
    if world == 1:
        ident = bytes(buf.raw)
    elif dist.is_available() and dist.is_initialized():
        objs = [bytes(buf.raw) if rank == 0 else None]
        dist.broadcast_object_list(objs, src=0, device=ctx.device if dist.get_backend() == "nccl" else None)
        ident = objs[0]
    else:
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29533")) + 1
        store = dist.TCPStore(addr, port, world, is_master=(rank == 0))
        if rank == 0:
            store.set("rcgan_comm_id", bytes(buf.raw))
        ident = store.get("rcgan_comm_id")
    ctx.check(ctx.lib.rcgan_comm_init(ctx.h, C.c_char_p(bytes(ident)), int(world), int(rank)))


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def mean_over_ranks(value, device=None):
    """Mean of a host scalar over the ranks (the reference averages tower costs, gan_resnet.py:697)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else None)
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()
