#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_run7; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cpc.py tests/test_gpu_api.py -q -x -s > $out/pytest_cpc.log 2>&1; tail -15 $out/pytest_cpc.log
python bench.py --option cond_pred_corr=1 --oracle-opt cond_pred_corr=1 > $out/bench_cpc_both_plain.json 2> $out/bench_cpc.err; echo "rc $?"
python bench.py --cpu-sample 0 > $out/bench_plain.json 2>/dev/null
for f in $out/bench*_plain.json; do python -c "
import json
d=json.load(open('$f')); w=d['workload_stats']; p=d.get('parity')
print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), 'it', round(w['qp_iter_mean'],2), 'unconv', w['qp_not_converged_frac'])
if p: print({k:p[k] for k in ('compared','count_above_1e-5','above_1e-5_without_kkt_certificate_or_beyond_5e-3','kkt_certified_frac','oracle_options')}, p['rel_err_per_instance'])"; done
