# round 3, first GPU call: the whole -m gpu suite, the bound experiment, the parity tail, the default bench line
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03a; mkdir -p $O
timeout 600 python tools/bound_experiment.py > $O/bound_experiment.txt 2>&1; echo "rc $?" >> $O/bound_experiment.txt
timeout 900 python tools/parity_tail.py 2048 10 $O/parity_tail.txt > $O/parity_tail.log 2>&1; echo "rc $?" >> $O/parity_tail.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -40 $O/pytest.log
cat $O/bound_experiment.txt
head -12 $O/parity_tail.txt
python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print("bench", round(d["value"]), d["ms_per_step"], d["roofline"]["kernel_ms"], d["parity"], d["workload_stats"]["active_row_frac"], d["workload_stats"]["status_nonzero_frac"])
except Exception as e:
    print("bench line unreadable", e); print(open("$O/bench.err").read()[-3000:])
PY
