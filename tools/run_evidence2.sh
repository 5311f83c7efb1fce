cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/evidence2
for f in valu_f64 valu_b32 dpp_hazard mfma_f64; do echo "== tools/micro/$f.hip"; timeout 300 ./build_ab/$f 2>&1 | tail -40; done > gpurun_out/evidence2/microbench.txt
bash tools/pmc_sq2.sh final > /dev/null 2>&1; cp gpurun_out/sq2_final.txt gpurun_out/evidence2/
python bench.py --model usv_model_guidance_ca1 > gpurun_out/evidence2/bench_m1.json 2> gpurun_out/evidence2/bench_m1.err
python bench.py --batch 1024 --horizon 20 --obstacles 3 > gpurun_out/evidence2/bench_cfg1.json 2> gpurun_out/evidence2/bench_cfg1.err
python bench.py --model usv_model --batch 65536 --horizon 20 --obstacles 0 > gpurun_out/evidence2/bench_m0.json 2> gpurun_out/evidence2/bench_m0.err
timeout 900 python bench.py --horizon 80 --obstacles 20 --moving --batch 65536 --cond-N 10 --steps 10 > gpurun_out/evidence2/bench_cfg4_b65536.json 2> gpurun_out/evidence2/bench_cfg4.err
timeout 900 python bench.py --horizon 80 --obstacles 20 --moving --batch 8192 --cond-N 10 --steps 10 --cpu-sample 0 > gpurun_out/evidence2/bench_cfg4_b8192_per_gpu.json 2> gpurun_out/evidence2/bench_cfg4b.err
python bench.py --workload r01 > gpurun_out/evidence2/bench_r01wl.json 2> gpurun_out/evidence2/bench_r01wl.err
for f in m1 cfg1 m0 cfg4_b65536 cfg4_b8192_per_gpu r01wl; do python -c "
import json; d=json.load(open('gpurun_out/evidence2/bench_$f.json')); print('$f', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel_ms'], d['workload_stats']['qp_iter_mean'], d['workload_stats']['status_nonzero_frac'], d['parity']['rel_err_per_instance'] if d['parity'] else None)"; done
cat gpurun_out/sq2_final.txt
python tools/latency_probe.py 2>&1 | tail -12
