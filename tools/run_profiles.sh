cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh $1 2>&1 | tail -30
BENCH_ARGS="--workload r01 --steps 10" bash tools/profile_round.sh $1_r01wl 2>&1 | tail -12
# counter calibration on the current plane layout
rm -rf gpurun_out/cal_$1; mkdir -p gpurun_out/cal_$1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/cal_$1/f -o f -- python tools/calib_traffic.py > gpurun_out/cal_$1/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/cal_$1/w -o w -- python tools/calib_traffic.py > gpurun_out/cal_$1/w.log 2>&1
grep calib gpurun_out/cal_$1/f.log
python - <<PY
import csv, glob
for sub,name in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    for fn in glob.glob("gpurun_out/cal_$1/"+sub+"/**/*counter_collection.csv", recursive=True):
        acc={}
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"]==name and "calib" in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]]=acc.get(r["Dispatch_Id"],0)+float(r["Counter_Value"])
        print(name, sorted(acc.items()))
PY
