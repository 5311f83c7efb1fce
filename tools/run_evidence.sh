cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/evidence
for f in chunk_bw tile_bw store_policy rcp_accuracy; do echo "== tools/micro/$f.hip"; timeout 300 ./build_ab/$f 2>&1 | tail -25; done > gpurun_out/evidence/microbench.txt
bash tools/pmc_sq.sh r02 > /dev/null 2>&1; cp gpurun_out/sq_r02.txt gpurun_out/evidence/
python bench.py --model usv_model_guidance_ca1 > gpurun_out/evidence/bench_m1.json 2> gpurun_out/evidence/bench_m1.err
python bench.py --batch 1024 --horizon 20 --obstacles 3 > gpurun_out/evidence/bench_cfg1.json 2> gpurun_out/evidence/bench_cfg1.err
python bench.py --model usv_model --batch 65536 --horizon 20 --obstacles 0 > gpurun_out/evidence/bench_m0.json 2> gpurun_out/evidence/bench_m0.err
for f in m1 cfg1 m0; do python -c "
import json; d=json.load(open('gpurun_out/evidence/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['workload_stats']['qp_iter_mean'], d['parity']['rel_err_per_instance'] if d['parity'] else None)"; done
cat gpurun_out/evidence/microbench.txt | head -60; cat gpurun_out/sq_r02.txt
