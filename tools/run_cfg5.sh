cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/cfg5
timeout 1500 python -m pytest tests/test_condensing.py -m gpu -x -q > gpurun_out/cfg5/pytest.log 2>&1; tail -15 gpurun_out/cfg5/pytest.log
timeout 900 python bench.py --horizon 80 --obstacles 20 --moving --batch 65536 --cond-N 10 --steps 10 > gpurun_out/cfg5/bench_b65536.json 2> gpurun_out/cfg5/bench_b65536.err
timeout 900 python bench.py --horizon 80 --obstacles 20 --moving --batch 8192 --cond-N 10 --steps 10 --cpu-sample 0 > gpurun_out/cfg5/bench_b8192.json 2> gpurun_out/cfg5/bench_b8192.err
for f in b65536 b8192; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/cfg5/bench_$f.json")); w=d["workload_stats"]
    print("$f", d["value"], d["roofline"]["kernel_ms"], "iters", w["qp_iter_mean"], "fail", w["status_nonzero_frac_at_step"], "active", w["active_row_frac"], d["parity"], d["config"]["workload"][:60])
except Exception as e:
    print("$f FAILED", e); print(open("gpurun_out/cfg5/bench_$f.err").read()[-1500:])
PY
done
