#!/bin/bash
# rocprofv3 evidence for the bench workload: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate passes.
# usage (on the GPU box): [BENCH_ARGS='--workload r01 --steps 10'] tools/profile_round.sh <tag>    -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --cpu-sample 0 --no-survey-verbatim $BENCH_ARGS > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python bench.py --cpu-sample 0 --no-survey-verbatim $BENCH_ARGS > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python bench.py --cpu-sample 0 --no-survey-verbatim $BENCH_ARGS > $out/write.log 2>&1
python - <<PY
import csv, glob, collections, json
out = "$out"
def pmc(sub, name):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in acc.items()}
f, w = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
with open(out + "/pmc_summary.csv", "w") as o:
    o.write("kernel,launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean\n")
    for k in sorted(set(f) | set(w)):
        o.write('"%s",%d,%.1f,%.1f\n' % (k, f.get(k, (0, 0))[1], f.get(k, (0, 0))[0], w.get(k, (0, 0))[0]))
print(open(out + "/pmc_summary.csv").read())
for fn in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print(open(fn).read()[:1500])
PY
tail -1 $out/trace.log | cut -c1-300
