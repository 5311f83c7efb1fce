"""Batch sharding across the GPUs of one node (one process per GPU).

Instances are independent, so the solve needs no collective: rank r owns the contiguous slice
[r*B/W, (r+1)*B/W) of a global batch.  The only optional exchange is an
all-gather of the optimal first controls u0 (nu doubles per instance) when one consumer wants them
on every rank; it goes through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  The reference has no counterpart: it solves one instance in one process.
"""
import numpy as np


def shard_bounds(total, world, rank):
    """Contiguous, balanced slice [lo, hi) of `total` instances for `rank` of `world`."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(array, world, rank):
    lo, hi = shard_bounds(array.shape[0], world, rank)
    return array[lo:hi]


def gather_first_controls(u0_local, total, group=None):
    """All-gather per-rank u0 blocks [B_r, nu] (torch tensors, CPU for gloo / device for nccl) into
    the global [total, nu] tensor on every rank.  Shards may be ragged by one instance."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    nu = u0_local.shape[1]
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    maxb = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxb, nu), dtype=u0_local.dtype, device=u0_local.device)
    pad[: u0_local.shape[0]] = u0_local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([out[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def device_tensor(ptr, shape, device_index=0):
    """Zero-copy torch view of a solver device buffer (usvmpc_get_device_ptr) of float64 `shape`."""
    from . import _capi
    if _capi.loaded_before_torch:
        raise RuntimeError("import torch before creating the first solver: torch and libusvmpc.so must share one "
                           "HIP runtime for zero-copy views / RCCL on solver buffers")
    import torch

    class _Wrap:
        pass

    w = _Wrap()
    w.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": "<f8",
                                  "data": (int(ptr), False), "version": 2, "strides": None}
    return torch.as_tensor(w, device=torch.device("cuda", device_index))


def first_controls_view(solver, device_index=0):
    """[B, nu] strided view of u[:, 0, :] living in the solver's device memory."""
    t = device_tensor(solver.device_ptr("u"), (solver.B, solver.N, solver.nu), device_index)
    return t[:, 0, :]


def split_workload(wl, world, rank):
    """Slice every per-instance array of a scenario workload for this rank."""
    out = {}
    for k, v in wl.items():
        out[k] = shard(v, world, rank) if isinstance(v, np.ndarray) else v
    return out
