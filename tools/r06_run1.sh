cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run1; mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc $?" >> $out/bench_default.err
python bench.py --option hpipm_mode=3 --oracle-opt hpipm_mode=R04 > $out/bench_r04.json 2> $out/bench_r04.err; echo "rc $?" >> $out/bench_r04.err
python bench.py --oracle-opt hpipm_mode=SPEED > $out/bench_oracle_speed.json 2> $out/bench_oracle_speed.err; echo "rc $?" >> $out/bench_oracle_speed.err
python bench.py --cpu-sample 0 --model usv_model_guidance_ca1 > $out/bench_m1.json 2>/dev/null
python bench.py --cpu-sample 0 --batch 8192 > $out/bench_b8192.json 2>/dev/null
python bench.py --cpu-sample 0 --batch 32768 > $out/bench_b32768.json 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_handover.py tests/test_gpu_parity.py tests/test_gpu_closed_loop.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -15 $out/pytest.log
for f in $out/bench*.json; do python -c "import json,sys; d=json.load(open('$f')); p=d.get('parity') or {}; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), p.get('count_above_1e-5'), p.get('compared'), (p.get('rel_err_per_instance') or {}).get('max'), d.get('workload_stats',{}).get('qp_iter_mean'))"; done
