#!/bin/bash
# Development aid: A/B of the two QP formulations at BASELINE configs[4]'s shape (uncondensed Riccati vs partial condensing to 10 stages).
# usage: tools/cond_ab.sh <tag> <batch> [steps]
tag=$1; B=${2:-8192}; steps=${3:-4}
for c in 0 10; do
  timeout 1200 python bench.py --horizon 80 --obstacles 20 --moving --batch $B --steps $steps --warmup 2 --cpu-sample 0 --cond-N $c > gpurun_out/${tag}_cfg4_b${B}_cond$c.json 2> gpurun_out/${tag}_cfg4_b${B}_cond$c.err
done
python - <<PY
import json
for c in (0, 10):
    try:
        j = json.loads(open("gpurun_out/${tag}_cfg4_b${B}_cond%d.json" % c).read().strip().splitlines()[-1])
        w = j["workload_stats"]
        print("cond_N", c, "solves/s %.0f" % j["value"], "ms/step %.1f" % j["ms_per_step"], j["roofline"]["kernel_ms"], "iter mean", w["qp_iter_mean"], "fail", w["status_nonzero_frac"])
    except Exception as ex:
        print("cond_N", c, "failed:", ex, open("gpurun_out/${tag}_cfg4_b${B}_cond%d.err" % c).read()[-400:])
PY
