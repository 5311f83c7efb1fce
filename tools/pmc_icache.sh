#!/bin/bash
# instruction-cache counters of the QP kernel (development aid). usage: tools/pmc_icache.sh <tag> [lib-tag]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; [ -n "$2" ] && export USVMPC_LIB=$PWD/build_ab/libusvmpc_$2.so
export USV_STATIC=1
out=gpurun_out/ic_$tag
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/a -o p -- python tools/quick_bench.py usv_model_pf_ca 65536 40 10 2 > $out/log.txt 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$out/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "qp_rti" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in sorted(acc.items()):
    print(c, "%.4g" % [sum(v) for v in d.values()][-1])
PY
tail -1 $out/log.txt | cut -c1-150
