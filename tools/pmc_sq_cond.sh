#!/bin/bash
# SQ issue / stall counters of the condensing kernel usv_qp_cond in two passes (development aid; last launch of a short bench run at
# BASELINE configs[4]'s shape, 8192 instances).  usage (GPU box): tools/pmc_sq_cond.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1
out=gpurun_out/sqc_$tag
rm -rf $out; mkdir -p $out
ARGS="--horizon 80 --obstacles 20 --moving --batch 8192 --steps 2 --warmup 1 --cpu-sample 0 --cond-N 10"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $out/a -o p -- python bench.py $ARGS > $out/log_a.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/b -o p -- python bench.py $ARGS > $out/log_b.txt 2>&1
python - <<PY
import csv, glob, collections
res = {}
for sub in "ab":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "qp_cond" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    for c, d in acc.items():
        vals = [sum(v) for v in d.values()]
        res[c + ("" if sub == "a" or c != "SQ_WAVE_CYCLES" else "_pass2")] = vals[-1]
open("gpurun_out/sqc_$tag.txt", "w").write("\n".join("%s %.4g" % kv for kv in sorted(res.items())) + "\n")
print(open("gpurun_out/sqc_$tag.txt").read())
PY
