# A/B of solver library builds on the bench workloads (GPU box): tools/ab.sh <out-tag> <lib1> <lib2> ...   ("stock" = in-tree lib)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
tag=$1; shift
mkdir -p gpurun_out/$tag
for lib in "$@"; do
  for wl in survey r01 ${EXTRA:-}; do
    if [ "$lib" = stock ]; then unset USVMPC_LIB; else export USVMPC_LIB=$PWD/build_ab/libusvmpc_$lib.so; fi
    args="--workload $wl"; [ "$wl" = m1 ] && args="--model usv_model_guidance_ca1"
    python bench.py $args --steps 10 --cpu-sample ${CPUS:-0} > gpurun_out/$tag/${lib}_$wl.json 2> gpurun_out/$tag/${lib}_$wl.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/${lib}_$wl.json"))
    w=d["workload_stats"]; print("%-8s %-6s %9.0f solves/s  lin %.2f ms  qp %.2f ms  iters %.2f  fail %.4f active %.3f parity %s" % ("$lib","$wl",d["value"],d["roofline"]["kernel_ms"]["usv_linearize"],d["roofline"]["kernel_ms"]["usv_qp_rti"],w["qp_iter_mean"],w["status_nonzero_frac"],w["active_row_frac"], (d["parity"] or {}).get("rel_err_per_instance")))
except Exception as e:
    print("$lib $wl FAILED", e); print(open("gpurun_out/$tag/${lib}_$wl.err").read()[-800:])
PY
  done
done
