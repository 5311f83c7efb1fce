#!/bin/bash
# Does the QP kernel get cheaper per stage visit when the planes of all instances in flight fit the 256 MB Infinity Cache?
# The number of instances in flight is fixed by the launch (2048 waves x 4 rows); their resident set shrinks with the horizon.
# usage (GPU box): tools/mall_probe.sh <out-tag> [lib-tag]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
tag=$1; [ -n "$2" ] && export USVMPC_LIB=$PWD/build_ab/libusvmpc_$2.so
mkdir -p gpurun_out/$tag
for N in 40 20 10 6 4 3; do
  for W in 0 1024; do
    python bench.py --horizon $N --steps 6 --cpu-sample 0 --option max_waves=$W > gpurun_out/$tag/n${N}_w$W.json 2> gpurun_out/$tag/n${N}_w$W.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/n${N}_w$W.json")); w=d["workload_stats"]; r=d["roofline"]
    B=65536; N=$N; it=w["qp_iter_mean"]; qp=r["kernel_ms"]["usv_qp_rti"]
    waves = $W if $W else 2048
    visits=B*(N+1)*it
    print("N=%2d waves %4d  qp %7.2f ms  iters %5.2f  fail %.4f  ->  %.3f ns per instance-stage-iteration;  resident set of the rows in flight: %5.0f MB" % (N, waves, qp, it, w["status_nonzero_frac"], qp*1e6/visits, waves*4*(N+1)*21*128/1e6))
except Exception as e:
    print("N=$N W=$W FAILED", e); print(open("gpurun_out/$tag/n${N}_w$W.err").read()[-600:])
PY
  done
done
