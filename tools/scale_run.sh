#!/bin/bash
# The multi-GPU evidence in one command, for the day an 8-GPU MI355X node is at hand (no scaling curve has been measured yet: DESIGN.md section 5).
#   tools/scale_run.sh [out_dir] [max_gpus]
# Per N in {1, 2, 4, 8} (up to max_gpus / what the node has):
#   weak     bench.py --gpus N                                        65 536 instances per GPU (the driver's SCALE run)
#   cfg3     bench.py --gpus N --global-batch 262144                  BASELINE configs[3]: ONE seed-1234 batch, shard b -> GPU floor(b N / 262144)
#   cfg4     bench.py --gpus N --global-batch 65536 --horizon 80 --obstacles 20 --moving     BASELINE configs[4]'s OCP and batch
# One JSON line per run (bench.py's contract) in <out_dir>/<kind>_n<N>.json, each with config.ranks_seen, config.per_rank_ms_per_step
# (min / max over the ranks), config.host_binding_rank0 and `allgather` (u0 / x1 / trajectory over RCCL, un-timed); summary.json has value
# and efficiency per N.  Launch: bench.py re-executes itself under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1).
# Host placement: every rank pins itself to its GPU's NUMA node (bench.py --bind-numa, sysfs numa_node of the GPU's PCI device: what
# `rocm-smi --showtoponuma` prints); recorded in the line.  Environment: HSA_ENABLE_IPC_MODE_LEGACY=0 (the driver only supports dmabuf IPC;
# without it RCCL fails with hipIpcGetMemHandle: invalid argument), NCCL_DEBUG=WARN; both are written into the line (config.env).
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/scale}; mkdir -p "$out"
have=$(python -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)
max=${2:-$have}; [ "$max" -gt "$have" ] && max=$have
export HSA_ENABLE_IPC_MODE_LEGACY=0
export NCCL_DEBUG=${NCCL_DEBUG:-WARN}
{ rocm-smi --showtoponuma 2>/dev/null || true; } > "$out/topo_numa.txt"
echo "GPUs visible: $have, running up to $max" | tee "$out/scale_run.log"
for n in 1 2 4 8; do
  [ "$n" -gt "$max" ] && break
  extra=""; [ "$n" -gt 1 ] && extra="--cpu-sample 0"
  python bench.py --gpus $n $extra > "$out/weak_n$n.json" 2> "$out/weak_n$n.err"; echo "weak n=$n rc $?" | tee -a "$out/scale_run.log"
  python bench.py --gpus $n --global-batch 262144 --cpu-sample 0 > "$out/cfg3_n$n.json" 2> "$out/cfg3_n$n.err"; echo "cfg3 n=$n rc $?" | tee -a "$out/scale_run.log"
  python bench.py --gpus $n --global-batch 65536 --horizon 80 --obstacles 20 --moving --cpu-sample 0 > "$out/cfg4_n$n.json" 2> "$out/cfg4_n$n.err"; echo "cfg4 n=$n rc $?" | tee -a "$out/scale_run.log"
done
python - "$out" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
summ = {}
for kind in ("weak", "cfg3", "cfg4"):
    rows = []
    for f in sorted(glob.glob(os.path.join(out, kind + "_n*.json")), key=lambda f: int(f.rsplit("_n", 1)[1][:-5])):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        c = d["config"]
        rows.append({"n_gpus": d["n_gpus"], "ranks_seen": c.get("ranks_seen"), "value": d["value"], "ms_per_step": d["ms_per_step"],
                     "per_rank_ms_per_step_min_max": c.get("per_rank_ms_per_step_min_max"), "instances_total": c.get("instances_total"),
                     "allgather_ms": {k: v["ms"] for k, v in (d.get("allgather") or {}).items()}})
    if rows:
        base = rows[0]["value"] / rows[0]["n_gpus"]
        for r in rows:   # weak: value / (N x the 1-GPU value); strong (cfg3 / cfg4: fixed total): the same ratio is the speed-up per GPU
            r["efficiency_vs_n1"] = r["value"] / (r["n_gpus"] * base)
        summ[kind] = rows
json.dump(summ, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summ, indent=1))
PY
