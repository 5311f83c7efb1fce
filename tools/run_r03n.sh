# round 3, head build: -m gpu suite, smoke, bench lines of every config (plain runs), rocprofv3 sets (kernel stats + PMC) for configs[2] (both
# models), configs[1] and configs[4], the shim's per-tick overhead and the small-batch latency table
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03n; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --model usv_model_guidance_ca1 > $O/bench_m1.json 2> $O/bench_m1.err
timeout 600 python bench.py --batch 1024 --horizon 20 --obstacles 3 > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 900 python bench.py --horizon 80 --obstacles 20 --moving --batch 65536 --cond-N 10 --steps 10 > $O/bench_cfg4_b65536.json 2> $O/bench_cfg4.err
timeout 900 python bench.py --horizon 80 --obstacles 20 --moving --batch 8192 --cond-N 10 --steps 10 --cpu-sample 0 > $O/bench_cfg4_b8192_per_gpu.json 2> $O/bench_cfg4b.err
timeout 600 python bench.py --model usv_model --batch 65536 --horizon 20 --obstacles 0 > $O/bench_m0.json 2> $O/bench_m0.err
for f in bench bench_m1 bench_cfg1 bench_cfg4_b65536 bench_cfg4_b8192_per_gpu bench_m0; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel_ms'], 'iters', round(d['workload_stats']['qp_iter_mean'],2), 'fail', d['workload_stats']['status_nonzero_frac'], 'active', d['workload_stats']['active_row_frac'], 'traffic', d['roofline']['traffic'], 'parity', (d['parity'] or {}).get('rel_err_per_instance'), (d['parity'] or {}).get('frac_above_1e-5'), (d['parity'] or {}).get('kkt_certified_frac'))" 2>&1 | tail -1; done
bash tools/profile_round.sh r03_c > $O/profile_r03_c.log 2>&1; tail -3 $O/profile_r03_c.log
BENCH_ARGS='--model usv_model_guidance_ca1' bash tools/profile_round.sh r03_c_m1 > $O/profile_r03_c_m1.log 2>&1; tail -3 $O/profile_r03_c_m1.log
BENCH_ARGS='--batch 1024 --horizon 20 --obstacles 3' bash tools/profile_round.sh r03_c_cfg1 > $O/profile_r03_c_cfg1.log 2>&1; tail -3 $O/profile_r03_c_cfg1.log
BENCH_ARGS='--horizon 80 --obstacles 20 --moving --batch 65536 --cond-N 10 --steps 10' bash tools/profile_round.sh r03_c_cfg4 > $O/profile_r03_c_cfg4.log 2>&1; tail -3 $O/profile_r03_c_cfg4.log
g++ -O1 -std=c++17 -Impc_collisionavoidance_amd/csrc/shim/include -o /tmp/shim_harness tests/shim_harness.cpp -Lmpc_collisionavoidance_amd/csrc/shim -lacados_ocp_solver_usv_model_guidance_ca1 -Wl,-rpath,$PWD/mpc_collisionavoidance_amd/csrc/shim && /tmp/shim_harness 40 | tail -1 > $O/shim_timing.txt 2>&1; cat $O/shim_timing.txt
timeout 900 python tools/latency_probe.py > $O/latency.txt 2>&1; cat $O/latency.txt
