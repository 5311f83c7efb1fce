"""The RCCL path on the one GPU a test box has (`pytest -m gpu`): a world_size-1 "nccl" process group with the device bound,
sharding.gather_results on the solver's own device buffers (zero-copy), and bench.py --gpus 1 under torch.distributed.run so
that its distributed branch (process group, barrier, max-reduction, all-gather) executes.  No scaling curve is measured here:
that needs a multi-GPU node."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_world1_nccl_gather_on_solver_device_buffers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_distributed_branch_under_the_launcher():
    env = _env()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "2048", "--cpu-sample", "0", "--bind-numa", "on"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["config"]["ranks_seen"] == 1
    g = d["allgather"]
    assert g is not None and g["u0"]["shape"] == [2048, 2] and g["x1"]["shape"] == [2048, 14]
    assert g["trajectory"]["shape"] == [2048, 41 * 14 + 40 * 2]
    # the multi-GPU line's placement record: the rank pinned itself to its GPU's NUMA node (or an even share of the CPUs), per-rank step times
    c = d["config"]
    assert ("NUMA node" in c["host_binding_rank0"] or "even share" in c["host_binding_rank0"]) and len(c["per_rank_ms_per_step"]) == 1
    assert c["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
