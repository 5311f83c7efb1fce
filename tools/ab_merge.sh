# A/B of the one-row-pass option on the GPU box: tools/ab_merge.sh <lib-tag|stock>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
[ "$1" != stock ] && export USVMPC_LIB=$PWD/build_ab/libusvmpc_$1.so
mkdir -p gpurun_out/abm
for rep in 1 2; do
for cfg in "m1k10:--model usv_model_guidance_ca1" "m2k9:--obstacles 9" "m2k4n20:--obstacles 4 --horizon 20" "cfg1:--batch 1024 --horizon 20 --obstacles 3"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  for m in 0 1; do
    python bench.py $args --steps 10 --cpu-sample 0 --option merge_box_rows=$m > gpurun_out/abm/${tag}_$m.json 2> gpurun_out/abm/${tag}_$m.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/abm/${tag}_$m.json")); w=d["workload_stats"]
    print("%-8s merge=$m %9.0f solves/s  lin %.2f ms  qp %.2f ms  iters %.2f  fail %.4f  parity %s" % ("$tag", d["value"], d["roofline"]["kernel_ms"]["usv_linearize"], d["roofline"]["kernel_ms"]["usv_qp_rti"], w["qp_iter_mean"], w["status_nonzero_frac"], (d["parity"] or {}).get("rel_err_per_instance")))
except Exception as e:
    print("$tag $m FAILED", e); print(open("gpurun_out/abm/${tag}_$m.err").read()[-600:])
PY
  done
done
done
