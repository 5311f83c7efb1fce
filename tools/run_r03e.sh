# round 3: full -m gpu suite on the build with the aux plane in LDS, bench lines of every config, rocprof set of the headline
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03e; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --model usv_model_guidance_ca1 > $O/bench_m1.json 2> $O/bench_m1.err
timeout 600 python bench.py --batch 1024 --horizon 20 --obstacles 3 > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 900 python bench.py --horizon 80 --obstacles 20 --moving --batch 65536 --cond-N 10 --steps 10 > $O/bench_cfg4_b65536.json 2> $O/bench_cfg4.err
timeout 600 python bench.py --model usv_model --batch 65536 --horizon 20 --obstacles 0 > $O/bench_m0.json 2> $O/bench_m0.err
for f in bench bench_m1 bench_cfg1 bench_cfg4_b65536 bench_m0; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel_ms'], 'iters', round(d['workload_stats']['qp_iter_mean'],2), 'fail', d['workload_stats']['status_nonzero_frac'], 'active', d['workload_stats']['active_row_frac'], 'parity', (d['parity'] or {}).get('rel_err_per_instance'), (d['parity'] or {}).get('frac_above_1e-5'), (d['parity'] or {}).get('kkt_certified_frac'))" 2>&1 | tail -1; done
bash tools/profile_round.sh r03_a > $O/profile_r03_a.log 2>&1; tail -12 $O/profile_r03_a.log
