"""Batch sharding across the GPUs of one node (one process per GPU).

Instances are independent, so the solve needs no collective: rank r owns the contiguous slice
[r*B/W, (r+1)*B/W) of a global batch.  The only optional exchange is an
all-gather of results when one consumer wants them on every rank - the first controls u0, the next
state x1 or the whole optimal trajectories; it goes through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  The reference has no counterpart: it solves one instance in one process.
"""
import numpy as np


def shard_bounds(total, world, rank):
    """Contiguous, balanced slice [lo, hi) of `total` instances for `rank` of `world`: instance b -> rank floor(b * world / total)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    # instance b lives on rank floor(b * world / total), i.e. rank r owns ceil(r T / W) <= b < ceil((r + 1) T / W)
    # (equal slices when world divides total, ragged by one instance otherwise)
    total, world = int(total), int(world)
    lo = -((-rank * total) // world)
    return lo, -((-(rank + 1) * total) // world)


def shard(array, world, rank):
    lo, hi = shard_bounds(array.shape[0], world, rank)
    return array[lo:hi]


def gather_rows(local, total, group=None):
    """All-gather per-rank blocks [B_r, ...] (torch tensors, CPU for gloo / device for nccl = RCCL) into the global
    [total, ...] tensor on every rank, in rank order.  Shards may be ragged by one instance (shard_bounds)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    maxb = max(hi - lo for lo, hi in sizes)
    local = local.contiguous()
    if all(hi - lo == maxb for lo, hi in sizes):
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)   # one collective, no padding copies
        return out
    pad = torch.zeros((maxb,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([out[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def gather_first_controls(u0_local, total, group=None):
    """All-gather of the optimal first controls: per-rank [B_r, nu] -> [total, nu] on every rank."""
    return gather_rows(u0_local, total, group=group)


def gather_results(solver, what, total, device_index=0, group=None):
    """The optional exchange of SURVEY.md 8(e) on the solver's own device buffers (zero-copy views, so with the
    nccl backend RCCL reads them in place): what = "u0" -> [total, nu]; "x1" -> [total, nx] (the state the closed
    loop continues from); "trajectory" -> [total, (N+1)*nx + N*nu], every instance's x_0..x_N followed by u_0..u_{N-1}."""
    import torch
    B, N, nx, nu = solver.B, solver.N, solver.nx, solver.nu
    x = device_tensor(solver.device_ptr("x"), (B, N + 1, nx), device_index)
    u = device_tensor(solver.device_ptr("u"), (B, N, nu), device_index)
    if what == "u0":
        return gather_rows(u[:, 0, :], total, group=group)
    if what == "x1":
        return gather_rows(x[:, 1, :], total, group=group)
    if what == "trajectory":
        return gather_rows(torch.cat([x.reshape(B, -1), u.reshape(B, -1)], dim=1), total, group=group)
    raise ValueError(what)


def device_tensor(ptr, shape, device_index=0):
    """Zero-copy torch view of a solver device buffer (usvmpc_get_device_ptr) of float64 `shape`."""
    from . import _capi
    if _capi.loaded_before_torch:
        raise RuntimeError("import torch before creating the first solver (or set USVMPC_PRELOAD_TORCH=1): torch and libusvmpc.so must "
                           "share one HIP runtime for zero-copy views / RCCL on solver buffers")
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
