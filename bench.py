#!/usr/bin/env python3
"""Headline benchmark: batched SQP-RTI solves/s (BASELINE.json metric) on N MI355X of one node.

A "step" is one pass of the hot path over one batch: one SQP-RTI iteration (linearise + QP + full
step) of every instance, followed by the closed-loop hand-over x0 <- x_1 + N(0, sigma) that the
reference's callers perform between ticks (scripts/usv_guidance_ca1/main.py:169-175), both on the
device.  Workload at N=1: BASELINE.json configs[2] as SURVEY.md 8(d) spells it out - batch 65536,
usv_model_pf_ca (3-DOF model, path-following LS cost, hard circular obstacles), horizon N=40 over
Tf = 2 s, 10 static obstacles, FP64, seed 1234, 3 un-timed warm-up iterations, then the timed closed loop with
sigma = 1e-3 (mpc_collisionavoidance_amd/scenario.py states the generator and its one clip).  `--workload r01`
replays the round-1 workload (dt = 0.01 s, obstacles beside the roll-out, no disturbance) for kernel-to-kernel
comparisons across rounds.

With --gpus N every rank runs the same batch size on its own GPU (weak scaling, no data-path collective:
instances are independent); the only collective inside the timed region is the closing barrier.  Started
without a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the environment) it re-executes itself under
torch.distributed.run with N ranks; it refuses to run if the node has fewer than N GPUs or if the launcher's
WORLD_SIZE disagrees with --gpus.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (usv_qp_rti), timed with HIP
events on the stream the kernels run on; `cpu_baseline` times the CPU oracle (a port, not the
reference) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak
_ID = {"usv_model": 0, "usv_model_guidance_ca1": 1, "usv_model_pf_ca": 2}


def algorithmic_bytes(nx, nu, N, K, moving=False):
    """inputs + warm-start iterate in + iterate out, FP64:
    8*[nx + ny + ny_e + P + nh + 2*((N+1)*nx + N*nu)] + 4 (status), P = np (static set) or (N+1)*np (moving)."""
    ny, ny_e, npar, nh = nx + nu, nx, 2 * K, K
    P = (N + 1) * npar if moving else npar
    return 8 * (nx + ny + ny_e + P + nh + 2 * ((N + 1) * nx + N * nu)) + 4


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


def bind_to_gpu_numa(local_rank, world, mode="auto"):
    """Pin this rank's host threads to the CPUs of its GPU's NUMA node (each rank's host thread feeds a persistent-kernel stream and polls
    events: eight of them on one socket's cores - or migrating - cost launch latency).  The node comes from the GPU's PCI device in sysfs
    (/sys/bus/pci/devices/<bdf>/numa_node, the figure `rocm-smi --showtoponuma` prints); without it (no NUMA information, one node) the
    usable CPUs are split evenly among the ranks.  mode: "auto" (only under a multi-rank launch), "on", "off".  Returns what was done."""
    if mode == "off" or (mode == "auto" and world <= 1) or not hasattr(os, "sched_setaffinity"):
        return "not bound"
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        return "not bound (no affinity interface)"
    node, cpus = None, None
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node >= 0:
            cpus = []
            for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
                a, _, b = part.partition("-")
                cpus += list(range(int(a), int(b or a) + 1))
            cpus = [c for c in cpus if c in allowed]
    except Exception:
        node, cpus = None, None
    if not cpus:   # no NUMA information: an even share of the usable CPUs
        n = max(1, len(allowed) // max(1, world))
        cpus = allowed[local_rank * n:(local_rank + 1) * n] or allowed
        how = "even share of the %d usable CPUs (no NUMA node for the GPU in sysfs)" % len(allowed)
    else:
        # the ranks of one node share its CPUs evenly too
        how = "NUMA node %d of GPU %d" % (node, local_rank)
    try:
        os.sched_setaffinity(0, cpus)
    except Exception as ex:
        return "not bound (%r)" % (ex,)
    return "%s: %d CPUs (%d..%d)" % (how, len(cpus), cpus[0], cpus[-1])


def baseline_config(name, B, world, N, K, moving):
    """Which BASELINE.json config this run is (the line must name itself, whatever the flags)."""
    total = B * world
    if name == "usv_model" and N == 20 and K == 0 and total == 1:
        return "BASELINE.json configs[0]"
    if N == 20 and K == 3 and total == 1024 and world == 1:
        return "BASELINE.json configs[1]"
    if N == 40 and K == 10 and B == 65536 and world == 1:
        return "BASELINE.json configs[2]"
    if N == 40 and K == 10 and total == 262144 and world == 8:
        return "BASELINE.json configs[3]"
    if N == 80 and K == 20 and moving and total == 65536:
        return "BASELINE.json configs[4]"
    if N == 40 and K == 10 and B == 65536:
        return "BASELINE.json configs[2] per GPU x%d GPUs (weak scaling)" % world
    return "custom (not a BASELINE.json config)"


def make_workload(name, N, K, batch, global_batch, workload, moving, rank, world):
    """The synthetic inputs of one rank: (workload dict, instances on this rank, dt, RK4 steps per interval, default sigma, noise mask).
    global_batch = 0: weak scaling, every rank generates its own `batch` instances (seed 1234 + rank).
    global_batch = G: SURVEY.md 8(d) configs[3] / [4] - ONE batch of G instances from seed 1234, instance b on rank floor(b * world / G)
    (the contiguous slices of sharding.shard_bounds); every rank generates the whole batch and keeps its slice."""
    from mpc_collisionavoidance_amd import scenario, sharding
    gen_B, gen_seed = (global_batch, 1234) if global_batch else (batch, 1234 + rank)
    if workload in ("survey", "survey-verbatim"):
        verbatim = workload == "survey-verbatim"
        dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
        wl = scenario.make_bench_batch(name, N, K, gen_B, seed=gen_seed, moving=moving, verbatim=verbatim)
        sigma, mask = 1e-3, (scenario.ALL_STATES_MASK if verbatim else scenario.NOISE_MASK[name])
    else:
        dt, steps = scenario.DT[name], 1
        wl = scenario.make_batch(name, N, K, gen_B, seed=gen_seed, moving=moving)
        sigma, mask = 0.0, (1 << 14) - 1
    if global_batch:
        lo, hi = sharding.shard_bounds(global_batch, world, rank)
        wl = sharding.split_workload(wl, world, rank)
        batch = hi - lo
    return wl, batch, dt, steps, sigma, mask


def survey_verbatim_leg(args, name, N, K, B, device, parity_sample):
    """A second, shorter timed region of the same run on SURVEY.md 8(d)'s generator TO THE LETTER (scenario.make_batch "survey_verbatim": no
    obstacle clip, acados' own initial guess x_k = x0, disturbance on every state; `--workload survey-verbatim` is the full-length line): the
    figure the survey's own wording gives, beside the headline's stated departures.  Warm-up of 3 ticks with the parity leg on the first
    `parity_sample` instances (0: none), then min(steps, 10) timed ticks.  Returns the `survey_verbatim` object of the line."""
    from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234, verbatim=True)
    ocp = usv_models.make_ocp(name, N * dt, N, None if name == "usv_model" else K)
    ocp.solver_options.sim_method_num_steps = steps
    ocp.solver_options.hpipm_mode = args.hpipm_mode
    s = BatchOcpSolver(ocp, B, device=device)
    scenario.load_into(s, wl)
    s.set_option("disturbance_mask", scenario.ALL_STATES_MASK)
    for kv in args.option:
        s.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    if K > 0:
        s.set_option("static_obstacles", 1)
    par = None
    S1 = int(parity_sample)
    if S1 > 0:
        from oracle import binding as ob
        from tests import parity_rule, util
        spec = ob.spec(_ID[name], N, N * dt, K, sim_steps=steps, hpipm_mode=args.hpipm_mode)
        data = tuple(wl[k][:S1] for k in ("yref", "yref_e", "p", "lh"))
        x0o = wl["x0"][:S1].copy()
        errs, n_same, n_above, n_bad, n_ok = [], 0, 0, 0, 0
    for w in range(3):
        if S1 > 0:
            s.sync()
            xin, uin = s.get_all("x")[:S1].copy(), s.get_all("u")[:S1].copy()
        s.solve_async()
        if S1 > 0:
            s.sync()
            xo, uo = xin.copy(), uin.copy()
            sto, ito = ob.rti_batch(spec, xo, uo, x0o, *data, threads=usable_cores())
            stg, qsg = s.get_int("status")[:S1], s.get_int("qp_status")[:S1]
            n_same += int((stg == sto).sum())
            ok = (sto == 0) & (ito < spec.opts.qp_iter_max) & (qsg == 0)
            n_ok += int(ok.sum())
            if ok.any():
                e = np.maximum(util.rel_err_per_instance(s.get_all("x")[:S1][ok], xo[ok]), util.rel_err_per_instance(s.get_all("u")[:S1][ok], uo[ok]))
                errs.append(e)
                # (the rule of tests/parity_rule.py on the sample: its indices are the first S1 of the batch)
                r = parity_rule.check(ob, spec, s, ok, e, xin, uin, x0o, data, soft=(name == "usv_model_guidance_ca1" and K > 0), max_frac=1.0)
                n_above += r["above"]
                n_bad += len(r["violations"])
        s.advance(1e-3, seed=3000 + w)
        if S1 > 0:
            s.sync()
            x0o = s.get("x0", 0)[:S1].copy()
    if S1 > 0 and errs:
        ee = np.concatenate(errs)
        par = {"ticks": 3, "instances": S1, "compared": int(ee.size), "status_agreement_frac": n_same / float(3 * S1),
               "converged_on_both_sides_frac": n_ok / float(3 * S1),
               "rel_err_per_instance": {"p50": float(np.percentile(ee, 50)), "p99": float(np.percentile(ee, 99)), "max": float(ee.max())},
               "count_above_1e-5": n_above, "rule_violations": n_bad}
    s.sync()
    nst = max(1, min(args.steps, 10))
    unconv0 = s.unconverged_total()
    t0 = time.perf_counter()
    for k in range(nst):
        s.solve_async()
        s.advance(1e-3, seed=4000 + k)
    s.sync()
    el = time.perf_counter() - t0
    unconv = s.unconverged_total() - unconv0
    qs, st, qi = s.get_int("qp_status"), s.get_int("status"), s.get_int("qp_iter")
    tmin = s.get("obs_tmin", 0) if K > 0 else np.full(B, 1e300)
    out = {"value": (B * nst - unconv) / el, "unit": "converged solves/s", "ms_per_step": el / nst * 1e3, "steps": nst, "warmup": 3,
           "solves_per_s_counting_unconverged_ones": B * nst / el,
           "qp_not_converged_frac": float((qs != 0).mean()), "status_nonzero_frac": float((st != 0).mean()),
           "qp_iter_mean": float(qi.mean()), "active_row_frac": float((tmin < 1e-3).mean()) if K > 0 else 0.0,
           "generator": "SURVEY.md 8(d) to the letter (scenario 'survey_verbatim': no obstacle clip, initial guess x_k = x0, disturbance on every state; "
                        "%d RK4 step(s) per interval); same batch size, seed and profile as the headline; the full-length line: --workload survey-verbatim" % steps,
           "parity": par}
    s.close()
    return out


def configs4_condensed_leg(args, device):
    """A third short region of the default line: BASELINE configs[4] ("+ partial condensing") at its per-GPU share - usv_model_pf_ca, N = 80, 20 MOVING
    obstacles, 8192 instances - solved with qp_solver_cond_N = 10 (usv_qp_cond: condense on the device, IPM on the ten dense stages, expand) and, on the
    same inputs, uncondensed (usv_qp_rti, the default formulation).  Tick 0 (both cold starts coincide) compares the two solutions; then 2 warm-up +
    4 timed closed-loop ticks each.  Returns the `configs4_condensed` object of the line."""
    from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
    from tests import util
    name, N, K, B, N2 = "usv_model_pf_ca", 80, 20, 8192, 10
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234, moving=True)
    out = {"workload": "BASELINE.json configs[4] at its per-GPU share: batch=%d, %s, N=%d, %d moving obstacles; qp_solver_cond_N=%d against the uncondensed default" % (B, name, N, K, N2)}
    sol = {}
    for key, cn in (("condensed", N2), ("uncondensed", 0)):
        ocp = usv_models.make_ocp(name, N * dt, N, K)
        ocp.solver_options.sim_method_num_steps = steps
        ocp.solver_options.hpipm_mode = args.hpipm_mode
        if cn:
            ocp.solver_options.qp_solver_cond_N = cn
        s = BatchOcpSolver(ocp, B, device=device)
        scenario.load_into(s, wl)
        s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
        st = s.solve()
        sol[key] = (st.copy(), s.get_int("qp_status").copy(), s.get_int("qp_iter").copy(), s.get_all("x").copy(), s.get_all("u").copy())
        for w in range(2):
            s.advance(1e-3, seed=5000 + w)
            s.solve_async()
        s.advance(1e-3, seed=5002)
        s.sync()
        nst = 4
        unconv0 = s.unconverged_total()
        t0 = time.perf_counter()
        for k in range(nst):
            s.solve_async()
            s.advance(1e-3, seed=6000 + k)
        s.sync()
        el = time.perf_counter() - t0
        unconv = s.unconverged_total() - unconv0
        out[key] = {"value": (B * nst - unconv) / el, "unit": "converged solves/s", "ms_per_step": el / nst * 1e3, "steps": nst,
                    "kernel": "usv_qp_cond" if cn else "usv_qp_rti", "qp_not_converged_frac": float((s.get_int("qp_status") != 0).mean()),
                    "qp_iter_mean": float(s.get_int("qp_iter").mean())}
        s.close()
    (sc, qc, ic, xc, uc), (su, qu, iu, xu, uu) = sol["condensed"], sol["uncondensed"]
    both = (qc == 0) & (qu == 0)
    e = np.maximum(util.rel_err_per_instance(xc[both], xu[both]), util.rel_err_per_instance(uc[both], uu[both])) if both.any() else np.zeros(1)
    out["tick0_condensed_vs_uncondensed"] = {"status_agreement_frac": float((sc == su).mean()), "converged_on_both_sides_frac": float(both.mean()),
                                             "same_iteration_count_frac": float((ic == iu)[both].mean()) if both.any() else 0.0,
                                             "rel_err_per_instance": {"p50": float(np.percentile(e, 50)), "p99": float(np.percentile(e, 99)), "max": float(e.max())}}
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become N ranks under torch.distributed.run."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but this node has %d visible GPU(s); refusing to run a "
                         "mislabelled benchmark\n" % (args.gpus, have))
        sys.exit(3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="usv_model_pf_ca")
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="SURVEY.md 8(d) configs[3] / [4]: ONE batch of this many instances (seed 1234) generated once and sharded over the "
                         "ranks, shard b -> GPU floor(b * gpus / batch) (contiguous slices, sharding.shard_bounds); overrides --batch")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--obstacles", type=int, default=10)
    ap.add_argument("--moving", action="store_true", help="obstacles move: per-stage p (BASELINE configs[4])")
    ap.add_argument("--workload", default="survey", choices=["survey", "survey-verbatim", "r01"],
                    help="survey: SURVEY.md 8(d) (dt 0.05 s) with the stated departures (config.workload); survey-verbatim: the generator as "
                         "SURVEY words it - no obstacle clip, initial guess x_k = x0, disturbance on every state (the RK4 step count stays); "
                         "r01: the round-1 workload (reference dt, no disturbance)")
    ap.add_argument("--sigma", type=float, default=None,
                    help="std of the Gaussian disturbance added at the hand-over (default: 1e-3 for survey, 0 for r01)")
    ap.add_argument("--cond-N", type=int, default=0, help="qp_solver_cond_N (0: acados default = N, no condensing)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="instances timed on the CPU oracle (0 = skip)")
    ap.add_argument("--oracle-opt", action="append", default=[], metavar="NAME=VALUE",
                    help="option of the CPU oracle in the un-timed parity leg (oracle/usv_oracle.h usv_opts), e.g. itref_corr_max=2 or "
                         "cond_pred_corr=1: HPIPM options the restatement leaves off by default; may be repeated")
    ap.add_argument("--hpipm-mode", default="BALANCE", choices=["BALANCE", "SPEED", "ROBUST", "R04"],
                    help="QP solver profile of the device AND of the oracle of the parity leg (include/usvmpc.h USVMPC_HPIPM_*; --oracle-opt overrides the oracle's)")
    ap.add_argument("--spread-mode", default="R04", choices=["BALANCE", "SPEED", "ROBUST", "R04", "none"],
                    help="second profile for parity.profile_spread: the device under --hpipm-mode against the device under this profile on the parity sample")
    ap.add_argument("--no-survey-verbatim", action="store_true",
                    help="skip the extra timed regions of the default line: SURVEY 8(d)'s generator to the letter (`survey_verbatim`) and BASELINE configs[4] with "
                         "qp_solver_cond_N = 10 against its uncondensed default (`configs4_condensed`)")
    ap.add_argument("--bind-numa", default="auto", choices=["auto", "on", "off"],
                    help="pin each rank's host threads to its GPU's NUMA node (auto: under a multi-rank launch)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="solver run-time option (usvmpc_set_option), e.g. dynamic_rows=0; may be repeated")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: the launcher started %d rank(s) but --gpus is %d; refusing to print a line whose "
                         "n_gpus is not the number of GPUs that worked" % (world, args.gpus))

    import torch
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (visible: %d)" % (local_rank, torch.cuda.device_count()))
    from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, sharding, usv_models
    binding = bind_to_gpu_numa(local_rank, world, args.bind_numa)

    dist = None
    ranks_seen = 1
    # under a launcher (WORLD_SIZE set) the ranks form an RCCL process group - also a single rank, so that the N > 1 code path
    # (barrier, max-reduction of the time, all-gather on the solver's device buffers) can be exercised on one GPU
    if world > 1 or "WORLD_SIZE" in os.environ:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        ones = torch.ones(1, dtype=torch.int32, device="cuda")
        dist.all_reduce(ones)                       # every rank that takes part adds one: RCCL really spans them
        ranks_seen = int(ones.item())
        if ranks_seen != args.gpus:
            raise SystemExit("bench.py: all-reduce saw %d ranks, --gpus is %d" % (ranks_seen, args.gpus))

    name, N, B = args.model, args.horizon, args.batch
    K = 0 if name == "usv_model" else args.obstacles
    G = args.global_batch
    wl, B, dt, steps, sigma0, mask = make_workload(name, N, K, B, G, args.workload, args.moving, rank, world)
    sigma = sigma0 if args.sigma is None else args.sigma
    ocp = usv_models.make_ocp(name, N * dt, N, None if name == "usv_model" else K)
    ocp.solver_options.sim_method_num_steps = steps
    if args.cond_N:
        ocp.solver_options.qp_solver_cond_N = args.cond_N
    ocp.solver_options.hpipm_mode = args.hpipm_mode
    solver = BatchOcpSolver(ocp, B, device=local_rank)
    cond_applied = bool(args.cond_N and args.cond_N != N)   # the QP is condensed to cond_N dense stages on the device (csrc/cond_ipm.hpp)
    scenario.load_into(solver, wl)
    solver.set_option("disturbance_mask", mask)
    for kv in args.option:
        solver.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    static = K > 0 and float(np.ptp(wl["p"], axis=1).max()) == 0.0 and float(np.ptp(wl["lh"], axis=1).max()) == 0.0
    if static:
        # the obstacle set of this workload is the same on every stage (as the reference's callers set it:
        # usv_pf_ca/main.py puts one pobs on all stages); the solver then keeps it in registers
        solver.set_option("static_obstacles", 1)
    nx, nu = solver.nx, solver.nu

    def barrier():
        solver.sync()                      # hipStreamSynchronize on the stream the kernels run on
        if dist is not None:
            torch.cuda.synchronize()       # and the whole device, before and after the rendezvous
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warmup (un-timed).  On a single GPU it doubles as a closed-loop parity check: the first S1 instances are
    # replayed tick by tick on the CPU oracle, fed the x0 the device's hand-over produced (disturbance included)
    check = rank == 0 and args.gpus == 1 and args.cpu_sample != 0
    parity = cpu_baseline = None
    if check:
        from oracle import binding as ob
        per_solve_ms = {"usv_model": 0.12, "usv_model_guidance_ca1": 1.2, "usv_model_pf_ca": 3.5}[name] * N / 40.0 * (1 + 0.15 * (steps - 1))
        S1 = min(B, int(max(32, min(B, 4000.0 / per_solve_ms))))       # ~4 s per tick on one core (batches below 32: all of them)
        def _oval(k, v):   # hpipm_mode=R04, cpc_factor=2.5, itref_corr_max=0 ...
            if k == "hpipm_mode":
                return v
            return float(v) if k in ("cpc_factor", "mu0", "alpha_min", "thr0") else int(float(v))
        oopts = dict({"hpipm_mode": args.hpipm_mode}, **{kv.split("=")[0]: _oval(*kv.split("=")[:2]) for kv in args.oracle_opt})
        spec = ob.spec(_ID[name], N, N * dt, K, sim_steps=steps, **oopts)
        x0o = wl["x0"][:S1].copy()
        errs, errs_x, errs_u = [], [], []
        same_status = n_ok = n_conv_dev = n_cert = n_above = n_above_uncert = 0
        worst_uncert = 0.0
        from tests import kkt as kkt_check   # independent acceptance: KKT conditions of every device solution (tests/kkt.py)
        soft_rows = name == "usv_model_guidance_ca1" and K > 0
        if cond_applied:
            solver.set_option("keep_multipliers", 1)   # (the condensed solve writes "lam" / "t" itself, into buffers that must exist)
        # parity.profile_spread: what the profile choice - the part of "parity unpinned" that is a choice - is worth on THIS sample: the same
        # instances, from the same inputs every tick, on the device under a second profile
        spread_solver, spread = None, []
        if args.spread_mode not in ("none", args.hpipm_mode) and not cond_applied:
            ocp2 = usv_models.make_ocp(name, N * dt, N, None if name == "usv_model" else K)
            ocp2.solver_options.sim_method_num_steps = steps
            ocp2.solver_options.hpipm_mode = args.spread_mode
            spread_solver = BatchOcpSolver(ocp2, S1, device=local_rank)
            scenario.load_into(spread_solver, {k: (v[:S1] if isinstance(v, np.ndarray) else v) for k, v in wl.items()})
            if static:
                spread_solver.set_option("static_obstacles", 1)
    for w in range(args.warmup):
        if check:   # the oracle starts every tick from the iterate and x0 the device starts it from ("same inputs")
            solver.sync()
            xo, uo = solver.get_all("x")[:S1].copy(), solver.get_all("u")[:S1].copy()
        solver.solve_async()
        if check:
            solver.sync()
            xo_in, uo_in = xo, uo
            xo, uo = xo.copy(), uo.copy()
            sto, ito = ob.rti_batch(spec, xo, uo, x0o, wl["yref"][:S1], wl["yref_e"][:S1], wl["p"][:S1], wl["lh"][:S1],
                                    threads=usable_cores())
            xg, ug = solver.get_all("x")[:S1], solver.get_all("u")[:S1]
            stg, qsg = solver.get_int("status")[:S1], solver.get_int("qp_status")[:S1]
            if spread_solver is not None:
                spread_solver.set_all("x", xo_in); spread_solver.set_all("u", uo_in); spread_solver.set("x0", 0, x0o)
                spread_solver.solve()
                both = (qsg == 0) & (spread_solver.get_int("qp_status") == 0)
                if both.any():
                    xs2, us2 = spread_solver.get_all("x"), spread_solver.get_all("u")
                    scx = np.maximum(1e-2, np.abs(xg[both]).max(axis=(0, 1))); scu = np.maximum(1e-2, np.abs(ug[both]).max(axis=(0, 1)))
                    spread.append(np.maximum((np.abs(xs2[both] - xg[both]) / scx).reshape(int(both.sum()), -1).max(axis=1),
                                             (np.abs(us2[both] - ug[both]) / scu).reshape(int(both.sum()), -1).max(axis=1)))
            # every solve the DEVICE calls converged must satisfy the KKT conditions of its QP (stat <= 1e-6, eq / ineq / comp
            # <= 1e-8), evaluated in numpy on the oracle's linearisation - independent of anybody's iteration path
            qpd = kkt_check.linearize_batch(ob, spec, xo_in, uo_in, x0o, wl["yref"][:S1], wl["yref_e"][:S1], wl["p"][:S1], wl["lh"][:S1])
            dzd = np.zeros((S1, N + 1, nx + nu))
            dzd[:, :N, :nu], dzd[:, :, nu:] = ug - uo_in, xg - xo_in
            pid = np.concatenate([np.zeros((S1, 1, nx)), solver.get_all("pi")[:S1]], axis=1)
            pad = lambda a: np.concatenate([a, np.zeros_like(a[:, :1])], axis=1)   # noqa: E731
            kr = kkt_check.kkt_batch(qpd, dzd, pid, solver.get_all("lam")[:S1], solver.get_all("t")[:S1],
                                     pad(solver.get_all("sl")[:S1]) if soft_rows else None,
                                     pad(solver.get_all("su")[:S1]) if soft_rows else None)
            conv_dev = (qsg == 0) & (stg == 0)
            n_conv_dev += int(conv_dev.sum())
            cert = kkt_check.certified(kr, 1.02e-6, 1.02e-8, 1.02e-8, 1.02e-8)
            n_cert += int((conv_dev & cert).sum())
            same_status += int((stg == sto).sum())
            ok = (sto == 0) & (ito < spec.opts.qp_iter_max) & (qsg == 0)
            n_ok += int(ok.sum())
            if ok.any():
                sc = np.maximum(1e-2, np.abs(xo[ok]).max(axis=(0, 1)))   # per-component scale
                su_ = np.maximum(1e-2, np.abs(uo[ok]).max(axis=(0, 1)))
                ex_ = (np.abs(xg[ok] - xo[ok]) / sc).reshape(int(ok.sum()), -1).max(axis=1)
                eu_ = (np.abs(ug[ok] - uo[ok]) / su_).reshape(int(ok.sum()), -1).max(axis=1)
                e_ = np.maximum(ex_, eu_)
                errs.append(e_); errs_x.append(ex_); errs_u.append(eu_)
                # the documented rule (tests/parity_rule.py): <= 1e-5 (north_star), or KKT-certified and then <= 5e-3
                above = e_ > 1e-5
                n_above += int(above.sum())
                unc = above & (~cert[ok] | (e_ > 5e-3))
                n_above_uncert += int(unc.sum())
                if unc.any():
                    worst_uncert = max(worst_uncert, float(e_[unc].max()))
        solver.advance(sigma, seed=1000 + w)
        if check:
            solver.sync()
            x0o = solver.get("x0", 0)[:S1].copy()
    if check and args.warmup > 0:
        parity = {"ticks": args.warmup, "instances": int(S1), "converged_on_both_sides_frac": n_ok / float(args.warmup * S1),
                  "status_agreement_frac": same_status / float(args.warmup * S1),
                  "rel_err_per_instance": {"p50": float(np.percentile(np.concatenate(errs), 50)),
                                           "p99": float(np.percentile(np.concatenate(errs), 99)),
                                           "max": float(np.concatenate(errs).max())} if errs else None,
                  "rel_err_x": {"p50": float(np.percentile(np.concatenate(errs_x), 50)), "p99": float(np.percentile(np.concatenate(errs_x), 99)),
                                "max": float(np.concatenate(errs_x).max())} if errs else None,
                  "rel_err_u": {"p50": float(np.percentile(np.concatenate(errs_u), 50)), "p99": float(np.percentile(np.concatenate(errs_u), 99)),
                                "max": float(np.concatenate(errs_u).max())} if errs else None,
                  "frac_above_1e-5": float((np.concatenate(errs) > 1e-5).mean()) if errs else None,
                  "count_above_1e-5": n_above, "compared": int(sum(len(e) for e in errs)),
                  "above_1e-5_without_kkt_certificate_or_beyond_5e-3": n_above_uncert,
                  "rule": "every instance <= 1e-5 (north_star) unless KKT-certified, then <= 5e-3 (tests/parity_rule.py); a violation makes this "
                          "run exit with status 4 after printing the line",
                  "kkt_certified_frac": n_cert / float(max(1, n_conv_dev)),
                  "kkt": "every solve the device reports converged, checked against the KKT conditions of its QP (stat <= 1e-6, "
                         "eq / ineq / comp <= 1e-8, lam, t >= 0) by tests/kkt.py on the oracle's linearisation: %d of %d" % (n_cert, n_conv_dev),
                  "profile_spread": ({"device_profile": args.hpipm_mode, "against_device_profile": args.spread_mode,
                                      "compared": int(sum(len(e) for e in spread)),
                                      "p50": float(np.percentile(np.concatenate(spread), 50)), "p99": float(np.percentile(np.concatenate(spread), 99)),
                                      "max": float(np.concatenate(spread).max()), "count_above_1e-5": int((np.concatenate(spread) > 1e-5).sum()),
                                      "note": "the SAME kernels under two QP solver profiles (include/usvmpc.h USVMPC_HPIPM_*), same instances, same inputs "
                                              "every tick: what the recalled-not-read part of acados' defaults (mu0, alpha_min, cond_pred_corr) moves the "
                                              "solution inside the IPM's exit tolerance ball - the measured size of 'parity unpinned' beyond rounding"}
                                     if spread else None),
                  "oracle_options": oopts,
                  "oracle_profile": "oracle/usv_oracle.c usv_opts_profile: HPIPM mode + acados' overwrites as recalled (BALANCE: mu0 1, alpha_min 1e-8, "
                                    "cond_pred_corr 1, itref_corr_max 2); the device runs its descriptor's profile (config.hpipm_mode)",
                  "vs": "CPU oracle (port; parity vs acados itself is unpinned); closed loop, every tick from the iterate "
                        "and x0 the device starts it from; error of an instance = max over (x, u) components of |dev - oracle| / "
                        "(that component's max |oracle| over the sample)"}
    if check and spread_solver is not None:
        spread_solver.close()
    barrier()
    unconv_before = solver.unconverged_total()

    # ---- timed region: exactly K steps
    t0 = time.perf_counter()
    for k in range(args.steps):
        solver.solve_async()
        solver.advance(sigma, seed=2000 + k)
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        mine = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        every = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]   # (each rank's own clock around the same K steps)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    nk = min(args.steps, 64)
    lin_ms, qp_ms = solver.kernel_ms(nk)
    fu_ms = solver.followup_ms(nk)   # (the follow-up launch of the hand-over, usv_qp_resume: part of qp_ms)
    # per-step times on the device clock ("report median"): start of tick i to start of tick i + 1 on the solver's stream;
    # the last step closes with the wall-clock remainder of the timed region
    # (device event deltas only - one fewer than steps: the last step has no following tick to close it, and the wall-clock remainder
    # would mix the host's launch offset and the final synchronisation into one sample)
    tick_ms = [float(v) for v in solver.tick_ms(nk)]
    mapping = solver.last_mapping()
    pipelined = B >= 16384 and not any(kv.split("=")[0] == "pipeline_linearize" and float(kv.split("=")[1]) == 0.0 for kv in args.option)
    fails = solver.fail_counts(nk)
    st = solver.get_int("status")
    qi = solver.get_int("qp_iter")
    qs = solver.get_int("qp_status")
    tmin = solver.get("obs_tmin", 0) if K > 0 else np.full(B, 1e300)

    # ---- optional exchange (north_star): all-gather of the optimal trajectories over RCCL, un-timed, on the solver's
    # own device buffers (zero-copy views)
    gather = None
    if dist is not None:
        g = {}
        for what in ("u0", "x1", "trajectory"):
            torch.cuda.synchronize()
            dist.barrier()
            c0 = time.perf_counter()
            full = sharding.gather_results(solver, what, B * world, device_index=local_rank)
            torch.cuda.synchronize()
            g[what] = {"ms": (time.perf_counter() - c0) * 1e3, "shape": list(full.shape),
                       "GB": full.numel() * 8 / 1e9}
            del full
        gather = g

    # a solve = one RTI iteration of one instance whose IPM converged to the stated tolerance - counted on the device by
    # every launch (usvmpc_unconverged_counts: qp_status != 0)
    unconv = float(solver.unconverged_total() - unconv_before)   # (a device-side running sum: exact for any number of steps)
    solves_here = float(B * args.steps)
    b_min = b_max = B
    if dist is not None:
        t = torch.tensor([unconv, solves_here], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        unconv, total_solves = float(t[0].item()), float(t[1].item())
        # (--global-batch not divisible by the ranks: the shards are ragged - the line reports the sum and the extremes, not rank 0's size)
        tb = torch.tensor([float(B), -float(B)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        b_max, b_min = int(tb[0].item()), int(-tb[1].item())
    else:
        total_solves = solves_here
    instances_total = int(round(total_solves / args.steps))
    value_all = total_solves / elapsed
    value = (total_solves - unconv) / elapsed
    balg = algorithmic_bytes(nx, nu, N, K, moving=args.moving)
    qp_avg_s = float(qp_ms.mean()) * 1e-3
    achieved = balg * B / qp_avg_s / 1e9
    # `traffic` comes from a rocprofv3 PMC profile (profiles/pmc_traffic.json, tools/profile_round.sh + save_profiles.py) and is
    # only printed when that profile was taken with THIS binary: the entry carries the library's sha256
    from mpc_collisionavoidance_amd import _capi
    lib_hash = _capi.lib_sha256()
    traffic, traffic_note = None, "no PMC profile of this workload in profiles/pmc_traffic.json"
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            for e in json.load(open(pmc)):
                if (e.get("model") == name and e.get("N") == N and e.get("K") == K and e.get("batch") == B
                        and e.get("workload", "r01") == args.workload and bool(e.get("moving", False)) == bool(args.moving)
                        and int(e.get("cond_N", 0) or 0) == (args.cond_N if cond_applied else 0)):
                    if e.get("lib_sha256") == lib_hash:
                        traffic, traffic_note = e.get("hbm_bytes_per_launch"), "profile %s (git %s), same library sha256" % (e.get("round"), e.get("git_head"))
                    elif traffic is None:
                        traffic_note = ("newest PMC profile of this workload (%s) was taken with another build of libusvmpc.so "
                                        "(sha256 %s, loaded %s): not quoted" % (e.get("round"), e.get("lib_sha256"), lib_hash))
        except Exception as ex:
            traffic, traffic_note = None, "profiles/pmc_traffic.json unreadable: %r" % (ex,)

    # the algorithmic FP64 flop count of one solve (dense count, no sparsity credit), with the
    # measured mean IPM iteration count
    nz_, nh_ = nx + nu, K
    c_f = {"usv_model": 60, "usv_model_guidance_ca1": 60, "usv_model_pf_ca": 150}.get(name, 100)
    n_ipm = float(qi.mean()) + 1.0  # factorisations = iterations + the final residual pass
    falg = steps * N * (4 * (2 * nx * nx * nz_ + c_f) + 8 * nx * (nz_ + 1)) + \
        n_ipm * N * (nx * nx * (nz_ + 1) + nx * nz_ * (nz_ + 1) + nz_ ** 3 / 3.0 + nh_ * nz_ * (nz_ + 1) + 8 * nz_ * nz_)
    fp64_tflops = falg * B / qp_avg_s / 1e12

    # ---- CPU baseline (rank 0, single GPU runs only): the oracle sources built -O3 -march=native on this host
    if check:
        cores = usable_cores()
        S = args.cpu_sample if args.cpu_sample > 0 else int(min(B, max(64, 12000.0 / per_solve_ms * cores)))  # ~12 s on all
        S = min(S, B)
        native = ob.native_lib() is not None   # -O3 -march=native build of the same sources, for timing only
        S1t = min(S1, 1024)
        reps1 = max(1, 32 // S1t)   # (a batch of a few instances: the same solves repeated, the mean is quoted)
        c0 = time.perf_counter()
        for _ in range(reps1):
            x1, u1 = wl["x_init"][:S1t].copy(), wl["u_init"][:S1t].copy()
            ob.rti_batch(spec, x1, u1, wl["x0"][:S1t], wl["yref"][:S1t], wl["yref_e"][:S1t], wl["p"][:S1t], wl["lh"][:S1t], native=True)
        c1sec = (time.perf_counter() - c0) / reps1
        xa, ua = wl["x_init"][:S].copy(), wl["u_init"][:S].copy()
        c0 = time.perf_counter()
        ob.rti_batch(spec, xa, ua, wl["x0"][:S], wl["yref"][:S], wl["yref_e"][:S], wl["p"][:S], wl["lh"][:S],
                     threads=cores, native=True)
        csec = time.perf_counter() - c0
        cpu_baseline = {"value": S / csec, "unit": "solves/s", "cores": cores, "kind": "port",
                        "single_core_value": S1t / c1sec, "single_instance_latency_ms": c1sec / S1t * 1e3,
                        "sample": "first %d instances of the same batch, 1 RTI iteration from the same initial guess, "
                                  "oracle/usv_oracle.c built %s, one instance per OpenMP thread on %d threads (%.1f s); "
                                  "single-core figure from the first %d instances (%.1f s)"
                                  % (S, "-O3 -march=native on this host" if native else "with the checker's flags (-O2)",
                                     cores, csec, S1t, c1sec)}

    # ---- the survey's generator to the letter, as a second short timed region of the same run (single-GPU default-workload lines)
    survey_verbatim = None
    if rank == 0 and args.gpus == 1 and args.workload == "survey" and not args.no_survey_verbatim and not cond_applied and not args.moving and K > 0:
        solver.sync()
        survey_verbatim = survey_verbatim_leg(args, name, N, K, B, local_rank, min(S1, 128) if check else 0)

    # ---- BASELINE configs[4] with its partial condensing applied, beside the uncondensed default at the same shape (same conditions as the region above)
    configs4_condensed = None
    if survey_verbatim is not None:
        configs4_condensed = configs4_condensed_leg(args, local_rank)

    departures = "none"
    if args.workload == "survey" and name == "usv_model_pf_ca":
        departures = ("(1) %d RK4 steps per interval (the model cannot take one 0.05 s step); (2) obstacle clip: an obstacle whose keep-out circle "
                      "the course ray would enter closer than 0.4 m + %.2g s * u is moved outwards along its bearing (11 %% of the obstacles at "
                      "N=40), field scaled with the horizon (range up to %.3g m); (3) initial guess = zero-input roll-out with the solver's "
                      "integrator instead of x_k = x0; (4) disturbance on (u, r) only" % (steps, 1.1 * N * dt if N * dt > 2.0 + 1e-9 else 0.6 * N * dt, 3.0 * N * dt))
    elif args.workload == "survey-verbatim":
        departures = ("%d RK4 step(s) per interval%s; otherwise the generator as SURVEY 8(d) words it: no obstacle clip, initial guess x_k = x0 "
                      "(acados' own), disturbance on every state" % (steps, " (one RK4 step of 0.05 s is outside usv_model_pf_ca's stability region: "
                                                                            "scenario.py DT)" if steps > 1 else ""))
    elif args.workload == "survey":
        departures = "obstacle field scaled with the horizon (range up to %.3g m); initial guess = straight-line roll-out" % (3.0 * N * dt)
    if rank == 0:
        ff = fails / float(B)
        at = lambda i: float(ff[i - 1]) if 1 <= i <= len(ff) else None   # noqa: E731
        out = {
            "metric": "batched SQP-RTI solves/sec (USV, N=%d horizon, %d obstacles)" % (N, K),
            "value": value,   # converged solves only; workload_stats has the rate counting every solve
            "unit": "solves/s",
            "n_gpus": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median": float(np.median(tick_ms)) if tick_ms else None,
            "ms_per_step_min_max": [float(min(tick_ms)), float(max(tick_ms))] if tick_ms else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: batch=%d per GPU, %s, N=%d, Tf=%g s (dt=%g s, %d RK4 step(s) per interval), %d %s obstacles, "
                            "GN SQP-RTI, generator '%s', closed loop x0<-x1+N(0,%g) on states mask 0x%x, %s; departures from SURVEY 8(d): %s"
                            % (baseline_config(name, B if not G else G // world, world, N, K, args.moving), b_max, name, N, N * dt, dt, steps, K,
                               "moving" if args.moving else "static", wl["generator"], sigma, mask,
                               ("ONE seed-1234 batch of %d, shard b -> GPU floor(b*%d/%d)" % (G, world, G)) if G else "seed 1234+rank",
                               departures),
                "ocp": name, "instances_per_gpu": B if b_min == b_max else None, "instances_per_gpu_min_max": [b_min, b_max],
                "instances_total": instances_total, "horizon": N, "obstacles": K,
                "qp_solver_cond_N": args.cond_N if cond_applied else N,
                "qp_formulation": ("partially condensed on the device: %d stages -> %d dense stages of %d, IPM + Riccati on those, expansion "
                                   "(csrc/cond_ipm.hpp)" % (N, args.cond_N, N // args.cond_N)) if cond_applied
                else "uncondensed: Riccati over the %d stages (blocks of one stage, the reference's own setting)" % N,
                "mapping": ("one OCP instance per wavefront (option 'wide': the rows of the wave share out the stage-local row work)" if mapping == 1
                            else "one OCP instance per workgroup of four wavefronts (options 'wide' / 'wide_waves')" if mapping == 4
                            else "four OCP instances per wavefront (one per 16-lane row)"),
                "hpipm_mode": "%s (QP solver profile of the device, include/usvmpc.h: %s)"
                              % (args.hpipm_mode, "HPIPM's values without acados' overwrites, no cond_pred_corr - the library's behaviour up to round 5"
                                 if args.hpipm_mode == "R04" else "acados' overwrites mu0 1 / alpha_min 1e-8 / tolerances 1e-6, 1e-8 / iter_max 50, cond_pred_corr 1"),
                "lib_sha256": lib_hash,
                "sharding": "batch-sharded x%d, no data-path collective" % world, "ranks_seen": ranks_seen,
                "per_rank_ms_per_step": per_rank_ms, "per_rank_ms_per_step_min_max": [min(per_rank_ms), max(per_rank_ms)],
                "host_binding_rank0": binding,
                "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_SOCKET_IFNAME", "HIP_VISIBLE_DEVICES",
                                                       "ROCR_VISIBLE_DEVICES", "OMP_NUM_THREADS") if os.environ.get(k) is not None},
            },
            "roofline": {
                "bound": "hbm", "kernel": "usv_qp_cond" if cond_applied else ("usv_qp_rti + usv_qp_resume" if float(fu_ms.mean()) > 0.0 else "usv_qp_rti"),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_note,
                "traffic_GBs": (traffic / qp_avg_s / 1e9) if traffic else None,
                "traffic_frac": (traffic / qp_avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "traffic_over_algorithmic": (traffic / (balg * B)) if traffic else None,
                "fp64_alg_tflops": fp64_tflops, "fp64_frac": fp64_tflops / FP64_PEAK_TFLOPS,
                "algorithmic_bytes_per_solve": balg, "algorithmic_flops_per_solve": falg,
                "kernel_ms": dict({"usv_linearize": float(lin_ms.mean()), ("usv_qp_cond" if cond_applied else "usv_qp_rti"): float((qp_ms - fu_ms).mean())},
                                  **({"usv_qp_resume": float(fu_ms.mean())} if float(fu_ms.mean()) > 0.0 else {})),
                "kernel_ms_followup_note": ("the QP of a tick is two launches: usv_qp_rti, whose rows hand instances past 20 IPM iterations over once the launch's "
                                            "queue is empty, and usv_qp_resume, which finishes them on the latency mapping (option handover_iter); "
                                            "`achieved`, `traffic_GBs` and fp64_* are taken over the sum of the two, `traffic` is the sum of their bytes") if float(fu_ms.mean()) > 0.0 else None,
                "kernel_ms_note": ("with pipeline_linearize (default for >= 16384 instances) the lineariser of tick t + 1 runs on a second stream "
                                   "in the tail of tick t's QP launch; kernel_ms.usv_linearize is then only the fix-up pass for the instances "
                                   "it had to skip (--option pipeline_linearize=0 shows the full lineariser)") if pipelined else None,
                "note": "`achieved` is SURVEY 8(d)'s algorithmic bytes / kernel time (structurally tiny for this path); "
                        "`traffic` is the measured HBM streaming of the per-stage planes per launch (rocprofv3 PMC, "
                        "profiles/pmc_traffic.json) and traffic_GBs / traffic_frac its rate; fp64_* from the "
                        "algorithmic flop count.  See DESIGN.md section 4.",
            },
            "cpu_baseline": cpu_baseline,
            "workload_stats": {
                "status_nonzero_frac": float((st != 0).mean()),
                "status_nonzero_frac_at_step": {"5": at(5), "10": at(10), "20": at(20), "last": float(ff[-1])},
                "status_nonzero_frac_per_step": [float(v) for v in ff],
                "qp_not_converged_frac": float((qs != 0).mean()),
                "solves_per_s_counting_unconverged_ones": value_all,
                "unconverged_solves_in_timed_region": unconv,
                "unconverged_counted_over_steps": args.steps,   # (device-side running sum over the whole timed region)
                "active_row_frac": float((tmin < 1e-3).mean()) if K > 0 else 0.0,
                "qp_iter_mean": float(qi.mean()), "qp_iter_p50": float(np.percentile(qi, 50)),
                "qp_iter_p99": float(np.percentile(qi, 99)), "qp_iter_max": int(qi.max()),
                "qp_iter_histogram": np.bincount(np.clip(qi, 0, None)).tolist(),
            },
            "parity": parity,
            "survey_verbatim": survey_verbatim,
            "configs4_condensed": configs4_condensed,
            "allgather": gather,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and parity.get("above_1e-5_without_kkt_certificate_or_beyond_5e-3"):
        sys.stderr.write("bench.py: parity rule violated (%d instance(s), worst %.3g): see the line's `parity`\n"
                         % (parity["above_1e-5_without_kkt_certificate_or_beyond_5e-3"], worst_uncert))
        sys.exit(4)


if __name__ == "__main__":
    main()
