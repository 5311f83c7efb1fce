#!/bin/bash
# memory-pipeline counters of the QP kernel (development aid). usage: tools/pmc_sq3.sh <tag> [lib-tag]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; [ -n "$2" ] && export USVMPC_LIB=$PWD/build_ab/libusvmpc_$2.so
export USV_STATIC=1
out=gpurun_out/sq3_$tag
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CYCLES SQ_CYCLES" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $out/$i -o p -- python tools/quick_bench.py usv_model_pf_ca 65536 40 10 2 > $out/log_$i.txt 2>&1
  tail -1 $out/log_$i.txt | cut -c1-200
done
python - <<PY
import csv, glob, collections
res = {}
for fn in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        if "qp_rti" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    for c, d in acc.items():
        vals = [sum(v) for v in d.values()]
        res[c] = vals[-1]
open("gpurun_out/sq3_$tag.txt", "w").write("\n".join("%s %.4g" % kv for kv in sorted(res.items())) + "\n")
print(open("gpurun_out/sq3_$tag.txt").read())
PY
