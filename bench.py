#!/usr/bin/env python3
"""Headline benchmark: batched SQP-RTI solves/s (BASELINE.json metric) on N MI355X of one node.

A "step" is one pass of the hot path over one batch: one SQP-RTI iteration (linearise + QP + full
step) of every instance, followed by the closed-loop hand-over x0 <- x_1 + N(0, sigma) that the
reference's callers perform between ticks (scripts/usv_guidance_ca1/main.py:169-175), both on the
device.  Workload at N=1: BASELINE.json configs[2] - batch 65536, usv_model_pf_ca (3-DOF model,
path-following LS cost, circular obstacles), horizon N=40, 10 obstacles, FP64.  With --gpus N every
rank runs the same batch size on its own GPU (weak scaling, no data-path collective: instances are
independent); the only collective is the timing barrier / max.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (usv_qp_rti), timed with HIP
events on the stream the kernels run on; `cpu_baseline` times the CPU oracle (a port, not the
reference) on a bounded sample of the same workload with one thread.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak


def algorithmic_bytes(nx, nu, N, K):
    """inputs + warm-start iterate in + iterate out, FP64, static obstacle set:
    8*[nx + ny + ny_e + np + nh + 2*((N+1)*nx + N*nu)] + 4 (status)."""
    ny, ny_e, npar, nh = nx + nu, nx, 2 * K, K
    return 8 * (nx + ny + ny_e + npar + nh + 2 * ((N + 1) * nx + N * nu)) + 4


def usable_cores():
    """CPUs this process may actually use: affinity mask and the cgroup CPU quota (the GPU boxes show 256 logical
    CPUs but run under a 16-CPU quota; more threads than that only get throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="usv_model_pf_ca")
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--obstacles", type=int, default=10)
    ap.add_argument("--sigma", type=float, default=0.0, help="std of the Gaussian disturbance added at the hand-over (reference loop: 0)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="instances timed on the CPU oracle (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models

    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    name, N, B = args.model, args.horizon, args.batch
    K = 0 if name == "usv_model" else args.obstacles
    dt = scenario.DT[name]
    ocp = usv_models.make_ocp(name, N * dt, N, None if name == "usv_model" else K)
    wl = scenario.make_batch(name, N, K, B, seed=1234 + rank)
    solver = BatchOcpSolver(ocp, B, device=local_rank)
    scenario.load_into(solver, wl)
    if K > 0:
        # the obstacle set of this workload is the same on every stage (as the reference's callers set it:
        # usv_pf_ca/main.py sets one pobs on all stages); the solver then keeps it in registers
        assert float(np.ptp(wl["p"], axis=1).max()) == 0.0 and float(np.ptp(wl["lh"], axis=1).max()) == 0.0
        solver.set_option("static_obstacles", 1)
    nx, nu = solver.nx, solver.nu

    def barrier():
        solver.sync()                      # hipStreamSynchronize on the stream the kernels run on
        if dist is not None:
            torch.cuda.synchronize()       # and the whole device, before and after the rendezvous
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warmup (un-timed); the first warmup step doubles as the parity spot check
    parity = None
    cpu_baseline = None
    first_x = first_u = None
    for w in range(args.warmup):
        solver.solve_async()
        if w == 0 and rank == 0 and args.gpus == 1:
            solver.sync()
            first_x, first_u = solver.get_all("x"), solver.get_all("u")
            first_qs = solver.get_int("qp_status")
        solver.advance(args.sigma, seed=1000 + w)
    barrier()

    # ---- timed region: exactly K steps
    t0 = time.perf_counter()
    for k in range(args.steps):
        solver.solve_async()
        solver.advance(args.sigma, seed=2000 + k)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    nk = min(args.steps, 64)
    lin_ms, qp_ms = solver.kernel_ms(nk)
    st = solver.get_int("status")
    qi = solver.get_int("qp_iter")
    qs = solver.get_int("qp_status")

    total_solves = world * B * args.steps
    value = total_solves / elapsed
    balg = algorithmic_bytes(nx, nu, N, K)
    qp_avg_s = float(qp_ms.mean()) * 1e-3
    achieved = balg * B / qp_avg_s / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            for e in json.load(open(pmc)):
                if e.get("model") == name and e.get("N") == N and e.get("K") == K and e.get("batch") == B:
                    traffic = e.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # the algorithmic FP64 flop count of one solve (dense count, no sparsity credit), with the
    # measured mean IPM iteration count
    nz_, nh_ = nx + nu, K
    c_f = {"usv_model": 60, "usv_model_guidance_ca1": 60, "usv_model_pf_ca": 150}.get(name, 100)
    n_ipm = float(qi.mean()) + 1.0  # factorisations = iterations + the final residual pass
    falg = N * (4 * (2 * nx * nx * nz_ + c_f) + 8 * nx * (nz_ + 1)) + \
        n_ipm * N * (nx * nx * (nz_ + 1) + nx * nz_ * (nz_ + 1) + nz_ ** 3 / 3.0 + nh_ * nz_ * (nz_ + 1) + 8 * nz_ * nz_)
    fp64_tflops = falg * B / qp_avg_s / 1e12

    # ---- CPU baseline + parity spot check (rank 0, single GPU runs only)
    if rank == 0 and args.gpus == 1 and args.cpu_sample != 0:
        from oracle import binding as ob
        per_solve_ms = {"usv_model": 0.12, "usv_model_guidance_ca1": 1.2, "usv_model_pf_ca": 3.5}[name] * N / 40.0
        cores = usable_cores()
        S1 = int(max(32, min(B, 4000.0 / per_solve_ms)))               # ~4 s on one core
        S = args.cpu_sample if args.cpu_sample > 0 else int(min(B, max(64, 12000.0 / per_solve_ms * cores)))  # ~12 s on all
        S = min(S, B)
        spec = ob.spec(_ID[name], N, N * dt, K)
        native = ob.native_lib() is not None   # -O3 -march=native build of the same sources, for timing only
        args1 = (wl["x0"][:S1], wl["yref"][:S1], wl["yref_e"][:S1], wl["p"][:S1], wl["lh"][:S1])
        # parity spot check: the checker build (bit-stable flags), sequential
        xo, uo = wl["x_init"][:S1].copy(), wl["u_init"][:S1].copy()
        sto, ito = ob.rti_batch(spec, xo, uo, *args1)
        # single core, then all usable cores, timed on the native build
        x1, u1 = wl["x_init"][:S1].copy(), wl["u_init"][:S1].copy()
        c0 = time.perf_counter()
        ob.rti_batch(spec, x1, u1, *args1, native=True)
        c1sec = time.perf_counter() - c0
        xa, ua = wl["x_init"][:S].copy(), wl["u_init"][:S].copy()
        c0 = time.perf_counter()
        ob.rti_batch(spec, xa, ua, wl["x0"][:S], wl["yref"][:S], wl["yref_e"][:S], wl["p"][:S], wl["lh"][:S],
                     threads=cores, native=True)
        csec = time.perf_counter() - c0
        cpu_baseline = {"value": S / csec, "unit": "solves/s", "cores": cores, "kind": "port",
                        "single_core_value": S1 / c1sec, "single_instance_latency_ms": c1sec / S1 * 1e3,
                        "sample": "first %d instances of the same batch, 1 RTI iteration from the same initial guess, "
                                  "oracle/usv_oracle.c built %s, one instance per OpenMP thread on %d threads (%.1f s); "
                                  "single-core figure from the first %d instances (%.1f s)"
                                  % (S, "-O3 -march=native on this host" if native else "with the checker's flags (-O2)",
                                     cores, csec, S1, c1sec)}
        S = S1  # the parity spot check below covers the checker-build sample
        if first_x is not None:
            ok = (sto == 0) & (ito < spec.opts.qp_iter_max) & (first_qs[:S] == 0)
            ex = float(np.abs(first_x[:S][ok] - xo[ok]).max() / max(1.0, np.abs(xo).max()))
            eu = float(np.abs(first_u[:S][ok] - uo[ok]).max() / max(1.0, np.abs(uo).max()))
            parity = {"instances": int(ok.sum()), "of": int(S), "max_rel_err_x": ex, "max_rel_err_u": eu,
                      "vs": "CPU oracle (port; parity vs acados itself is unpinned)"}

    if rank == 0:
        out = {
            "metric": "batched SQP-RTI solves/sec (USV, N=%d horizon, %d obstacles)" % (N, K),
            "value": value,
            "unit": "solves/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[2]: batch=%d per GPU, %s, N=%d, %d static obstacles, dt=%g s, GN SQP-RTI, "
                            "closed loop x0<-x1+N(0,%g), seed 1234+rank" % (B, name, N, K, dt, args.sigma),
                "ocp": name, "instances_per_gpu": B, "instances_total": world * B, "horizon": N, "obstacles": K,
                "sharding": "batch-sharded x%d, no collective" % world,
            },
            "roofline": {
                "bound": "hbm", "kernel": "usv_qp_rti",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_GBs": (traffic / qp_avg_s / 1e9) if traffic else None,
                "traffic_frac": (traffic / qp_avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "fp64_alg_tflops": fp64_tflops, "fp64_frac": fp64_tflops / FP64_PEAK_TFLOPS,
                "algorithmic_bytes_per_solve": balg, "algorithmic_flops_per_solve": falg,
                "kernel_ms": {"usv_linearize": float(lin_ms.mean()), "usv_qp_rti": float(qp_ms.mean())},
                "note": "`achieved` is SURVEY 8(d)'s algorithmic bytes / kernel time (structurally tiny for this path); "
                        "`traffic` is the measured HBM streaming of the per-stage planes per launch (rocprofv3 PMC, "
                        "profiles/pmc_traffic.json) and traffic_GBs / traffic_frac its rate; fp64_* from the "
                        "algorithmic flop count.  See DESIGN.md section 4.",
            },
            "cpu_baseline": cpu_baseline,
            "workload_stats": {
                "status_nonzero_frac": float((st != 0).mean()),
                "qp_not_converged_frac": float((qs != 0).mean()),
                "converged_solves_per_s": value * float(1.0 - (qs != 0).mean()),
                "qp_iter_mean": float(qi.mean()), "qp_iter_p50": float(np.percentile(qi, 50)),
                "qp_iter_p99": float(np.percentile(qi, 99)), "qp_iter_max": int(qi.max()),
            },
            "parity": parity,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


_ID = {"usv_model": 0, "usv_model_guidance_ca1": 1, "usv_model_pf_ca": 2}

if __name__ == "__main__":
    main()
