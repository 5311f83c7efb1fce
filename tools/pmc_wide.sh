#!/bin/bash
# SQ counters of the QP kernel on BASELINE configs[1] (1024 instances, N = 20, K = 3), latency mapping against throughput mapping:
# where do a lone wave's cycles go?   usage (through gpurun): tools/pmc_wide.sh <tag>   -> gpurun_out/<tag>_wide_sq_counters.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1
res=gpurun_out/${tag}_wide_sq_counters.txt
echo "SQ counters of usv_qp_rti per launch, BASELINE configs[1] (usv_model_pf_ca, 1024 instances, N = 20, K = 3), python bench.py --batch 1024 --horizon 20 --obstacles 3 --steps 6" > $res
for mode in wide throughput; do
  opt=""; [ $mode = throughput ] && opt="--option wide=0"
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    out=gpurun_out/sqw_${tag}_$mode; rm -rf $out; mkdir -p $out
    rocprofv3 --pmc $pass --output-format csv -d $out -o p -- python bench.py --cpu-sample 0 --batch 1024 --horizon 20 --obstacles 3 --steps 6 --warmup 2 $opt > $out/log.txt 2>&1
    python - >> $res <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "qp_rti" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
print("== $mode mapping")
for c, d in sorted(acc.items()):
    v = list(d.values())
    print("%-24s %.4g   (last of %d launches)" % (c, v[-1], len(v)))
PY
  done
done
cat $res
