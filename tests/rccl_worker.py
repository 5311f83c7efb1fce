"""Child process of tests/test_gpu_rccl.py: a world_size-1 RCCL ("nccl") process group bound to cuda:0, a real BatchOcpSolver,
and sharding.gather_results on zero-copy views of the solver's own device buffers.  Deliberately imports the solver package
BEFORE torch, with USVMPC_PRELOAD_TORCH=1: the library then brings torch in first itself (the opt-in of _capi.load; without it the
order "package, then torch" is refused by sharding.device_tensor with an explanation - tests/test_host_api.py)."""
import os
import sys

os.environ["USVMPC_PRELOAD_TORCH"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, sharding, usv_models  # noqa: E402


def main():
    name, N, K, B = "usv_model_pf_ca", 20, 6, 203
    wl = scenario.make_bench_batch(name, N, K, B, seed=77)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    s = BatchOcpSolver(ocp, B)            # loads libusvmpc.so (and, through _capi.load with USVMPC_PRELOAD_TORCH=1, torch before it)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    scenario.load_into(s, wl)
    for _ in range(2):
        s.solve()
    x, u = s.get_all("x"), s.get_all("u")
    full = {w: sharding.gather_results(s, w, B, device_index=0) for w in ("u0", "x1", "trajectory")}
    torch.cuda.synchronize()
    assert full["u0"].is_cuda and tuple(full["u0"].shape) == (B, s.nu)
    assert np.array_equal(full["u0"].cpu().numpy(), u[:, 0])
    assert np.array_equal(full["x1"].cpu().numpy(), x[:, 1])
    assert np.array_equal(full["trajectory"].cpu().numpy(), np.concatenate([x.reshape(B, -1), u.reshape(B, -1)], axis=1))
    # the collectives bench.py's timing uses
    t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert t.item() == 1.25
    dist.destroy_process_group()
    s.close()
    print("RCCL_WORLD1_OK")


if __name__ == "__main__":
    main()
