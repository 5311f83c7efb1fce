#!/bin/bash
# round 5, GPU call 2: the -m gpu suite on the contract(on) library, the outlier fixture re-taken, the two headline workloads
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_run2; mkdir -p $out
python tools/outlier_fixture.py 2048 10 $out/parity_outliers_pf_ca.npz > $out/outlier_fixture.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
python bench.py > $out/bench_plain.json 2> $out/bench_plain.err; echo "rc $?" >> $out/bench_plain.err
python bench.py --workload survey-verbatim > $out/bench_survey_verbatim_plain.json 2> $out/bench_survey_verbatim_plain.err; echo "rc $?" >> $out/bench_survey_verbatim_plain.err
python bench.py --workload survey-verbatim --model usv_model_guidance_ca1 --cpu-sample 0 > $out/bench_survey_verbatim_m1_plain.json 2> $out/bench_survey_verbatim_m1_plain.err
tail -30 $out/pytest.log; tail -5 $out/outlier_fixture.log
for f in $out/bench*_plain.json; do python -c "
import json,sys
d=json.load(open('$f')); w=d['workload_stats']
print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), d.get('ms_per_step_median'), 'fail', w['status_nonzero_frac'], 'unconv', w['qp_not_converged_frac'], 'it', w['qp_iter_mean'], 'active', w['active_row_frac'])
print(d.get('parity'))"; done
cat $out/*.err | tail -20
