# round 3: sort-key A/B, counter calibration, the full -m gpu suite on the head build
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03f; mkdir -p $O
ab() { tag=$1; shift
  for v in 0 1 0 1; do
    timeout 600 python bench.py --cpu-sample 0 --steps 20 --option sort_two_ticks=$v "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag sort_two_ticks=$v', round(d['value']), 'qp ms', round(d['roofline']['kernel_ms']['usv_qp_rti'], 2), 'lin ms', round(d['roofline']['kernel_ms']['usv_linearize'], 2), 'iters', round(d['workload_stats']['qp_iter_mean'], 2))"
  done; }
ab m2 > $O/ab_sort.txt 2>&1
ab m1 --model usv_model_guidance_ca1 >> $O/ab_sort.txt 2>&1
cat $O/ab_sort.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_f -o f -- python tools/calib_traffic.py > $O/cal_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_w -o w -- python tools/calib_traffic.py > $O/cal_w.log 2>&1
python - <<PY
import csv, glob
for sub, name in (("cal_f", "FETCH_SIZE"), ("cal_w", "WRITE_SIZE")):
    for fn in glob.glob("$O/" + sub + "/**/*counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == name and "calib" in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]] = acc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        print(name, sorted(acc.values()))
PY
grep calib $O/cal_f.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
