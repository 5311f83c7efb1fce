"""N > 1 path on CPU: two processes over gloo shard a global batch contiguously (no data-path
collective) and all-gather the first controls u0, as bench.py --gpus N / RCCL would on a node."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpc_collisionavoidance_amd import scenario, sharding


def test_shard_bounds_cover_and_balance():
    for total in (1, 5, 8, 13, 65536, 262144):
        for world in (1, 2, 3, 8):
            sl = [sharding.shard_bounds(total, world, r) for r in range(world)]
            assert sl[0][0] == 0 and sl[-1][1] == total
            assert all(sl[r][1] == sl[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in sl]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(8, 2, 2)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = scenario.make_batch("usv_model_pf_ca", 6, 3, total, seed=1234)   # same global batch on both ranks
    mine = sharding.split_workload(wl, world, rank)
    lo, hi = sharding.shard_bounds(total, world, rank)
    assert mine["x0"].shape[0] == hi - lo and np.array_equal(mine["p"], wl["p"][lo:hi])
    # stand-in for the per-rank solve: a deterministic function of the instance data
    u0 = torch.from_numpy(np.stack([mine["x0"][:, 3] * 2.0, mine["x0"][:, 12] - 1.0], axis=1))
    full = sharding.gather_first_controls(u0, total)
    expect = np.stack([wl["x0"][:, 3] * 2.0, wl["x0"][:, 12] - 1.0], axis=1)
    ok = full.shape == (total, 2) and np.array_equal(full.numpy(), expect)
    # timing reduction used by bench.py: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and t.item() == float(world)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7])
def test_two_rank_sharding_and_gather(total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}
