#!/bin/bash
# SQ occupancy / stall counters of the QP kernel (development aid). usage: tools/pmc_sq.sh <tag> [lib]
# writes gpurun_out/sq_<tag>.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; lib=${2:-}
[ -n "$lib" ] && export USVMPC_LIB=$lib
export USV_STATIC=1
out=gpurun_out/sq_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $out -o p -- python tools/quick_bench.py usv_model_pf_ca 65536 40 10 2 > $out/log.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "qp_rti" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
res = {}
for c, d in acc.items():
    vals = [sum(v) for v in d.values()]
    res[c] = vals[-1]
open("gpurun_out/sq_$tag.txt", "w").write("\n".join("%s %.4g" % kv for kv in sorted(res.items())) + "\n")
print(open("gpurun_out/sq_$tag.txt").read())
PY
tail -2 $out/log.txt
