#!/bin/bash
# SQ issue / stall counters of the QP kernel in two passes (development aid). usage: tools/pmc_sq2.sh <tag> [lib-tag]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; [ -n "$2" ] && export USVMPC_LIB=$PWD/build_ab/libusvmpc_$2.so
export USV_STATIC=1
out=gpurun_out/sq2_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $out/a -o p -- python tools/quick_bench.py usv_model_pf_ca 65536 40 10 2 > $out/log_a.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/b -o p -- python tools/quick_bench.py usv_model_pf_ca 65536 40 10 2 > $out/log_b.txt 2>&1
python - <<PY
import csv, glob, collections
res = {}
for sub in "ab":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "qp_rti" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    for c, d in acc.items():
        vals = [sum(v) for v in d.values()]
        res[c + ("" if sub == "a" or c != "SQ_WAVE_CYCLES" else "_pass2")] = vals[-1]
open("gpurun_out/sq2_$tag.txt", "w").write("\n".join("%s %.4g" % kv for kv in sorted(res.items())) + "\n")
print(open("gpurun_out/sq2_$tag.txt").read())
PY
tail -1 $out/log_a.txt; tail -1 $out/log_b.txt
