# do the kernels of two handles (two streams) overlap in time?  kernel trace of tools/two_stream_probe.py with 2 handles
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03j; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python tools/two_stream_probe.py 2 > $O/log.txt 2>&1
tail -2 $O/log.txt
python - <<PY
import csv, glob
fn = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn)) if "usv_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print("columns", list(rows[0].keys()))
last = rows[-60:]
for r in last:
    print("%-22s q %s  start %9.3f ms  end %9.3f ms  dur %7.3f" % (r["Kernel_Name"].split("<")[0].replace("void ", "")[:22], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
