#!/bin/bash
# The round's GPU evidence in one gpurun call (round 6): the -m gpu suite, the bench lines, the rocprofv3 sets of the head.
# usage (through gpurun): tools/r06_round.sh <tag, e.g. r06_a> [what: all | tests | bench | prof]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=$1; W=${2:-all}; out=gpurun_out/$R; mkdir -p $out
if [ $W = all ] || [ $W = tests ]; then
  timeout 3000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -5 $out/pytest.log
fi
if [ $W = all ] || [ $W = bench ]; then
  /usr/bin/env bash -c "time python bench.py > $out/bench_plain.json 2> $out/bench_plain.err" 2> $out/bench_plain.time; echo "rc $?" >> $out/bench_plain.err
  python bench.py --hpipm-mode R04 --spread-mode none --no-survey-verbatim > $out/bench_profile_r04_plain.json 2> $out/bench_profile_r04_plain.err; echo "rc $?" >> $out/bench_profile_r04_plain.err
  python bench.py --oracle-opt hpipm_mode=SPEED --spread-mode none --no-survey-verbatim > $out/bench_oracle_speed_plain.json 2>/dev/null
  python bench.py --workload survey-verbatim > $out/bench_survey_verbatim_plain.json 2> $out/bench_survey_verbatim_plain.err; echo "rc $?" >> $out/bench_survey_verbatim_plain.err
  python bench.py --model usv_model_guidance_ca1 > $out/bench_m1_plain.json 2>/dev/null
  python bench.py --cpu-sample 0 --model usv_model --horizon 20 > $out/bench_m0_plain.json 2>/dev/null
  python bench.py --batch 1024 --horizon 20 --obstacles 3 > $out/bench_cfg1_plain.json 2>/dev/null
  python bench.py --cpu-sample 0 --batch 1024 --horizon 20 --obstacles 3 --option wide=0 > $out/bench_cfg1_throughput_mapping_plain.json 2>/dev/null
  python bench.py --cpu-sample 0 --batch 1 --horizon 40 --obstacles 10 > $out/bench_one_instance_plain.json 2>/dev/null
  for b in 4096 8192 16384 32768; do python bench.py --cpu-sample 0 --batch $b > $out/bench_b${b}_plain.json 2>/dev/null; done
  python bench.py --cpu-sample 0 --horizon 80 --obstacles 20 --moving > $out/bench_cfg4_b65536_plain.json 2>/dev/null
  python bench.py --cpu-sample 0 --horizon 80 --obstacles 20 --moving --batch 8192 > $out/bench_cfg4_b8192_per_gpu_plain.json 2>/dev/null
  python bench.py --cpu-sample 0 --horizon 80 --obstacles 20 --moving --batch 8192 --cond-N 10 > $out/bench_cfg4_b8192_condN10_plain.json 2>/dev/null
  # in-flight set vs Infinity Cache: fewer resident waves on the headline
  for mw in 512 768 1024 1536; do python bench.py --cpu-sample 0 --no-survey-verbatim --option max_waves=$mw > $out/bench_maxwaves${mw}_plain.json 2>/dev/null; done
  for f in $out/bench*_plain.json; do python -c "import json,sys; d=json.load(open('$f')); p=d.get('parity') or {}; sv=d.get('survey_verbatim') or {}; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), 'above', p.get('count_above_1e-5'), 'of', p.get('compared'), 'max', (p.get('rel_err_per_instance') or {}).get('max'), 'spread', (p.get('profile_spread') or {}).get('max'), 'sv', sv.get('value'))"; done
fi
if [ $W = all ] || [ $W = lat ]; then
  for m in usv_model_pf_ca usv_model_guidance_ca1; do python tools/latency_probe.py $m 20 3 1,64,512,1024,2048; python tools/latency_probe.py $m 40 10 1,64,256,512; done > $out/latency_probe.txt 2>&1
  python tools/latency_probe.py usv_model_guidance_ca1 100 8 1,16,128,1024 >> $out/latency_probe.txt 2>&1   # the reference node's own shape: N = 100, K = 8
  python tools/latency_probe.py usv_model 20 0 1,64,1024,2048 >> $out/latency_probe.txt 2>&1
  python tools/latency_probe.py usv_model_pf_ca 80 20 1,64,256,512 >> $out/latency_probe.txt 2>&1
  for m in usv_model_pf_ca usv_model_guidance_ca1; do python tools/latency_probe.py $m 40 10 1; done > $out/latency_probe_take2.txt 2>&1
  python tools/latency_probe.py usv_model_guidance_ca1 100 8 1 >> $out/latency_probe_take2.txt 2>&1
  python -m pytest tests/test_shim.py -m gpu -q -s 2>&1 | grep timing >> $out/latency_probe.txt
  for m in usv_model_pf_ca usv_model_guidance_ca1; do for a in "20 3" "40 10" "40 20" "80 20" "100 8" "60 4"; do python tools/latency_probe.py $m $a 64,256,512,1024,2048,4096,8192; done; done > $out/policy_audit.txt 2>&1
  python tools/latency_probe.py usv_model 20 0 64,256,1024,2048,4096,8192 >> $out/policy_audit.txt 2>&1
  tail -4 $out/latency_probe.txt
fi
if [ $W = all ] || [ $W = prof ]; then
  tools/pmc_sq_cond.sh $R > $out/sq_cond.log 2>&1
  tools/profile_round.sh $R > $out/prof.log 2>&1
  BENCH_ARGS='--model usv_model_guidance_ca1' tools/profile_round.sh ${R}_m1 > $out/prof_m1.log 2>&1
  BENCH_ARGS='--batch 1024 --horizon 20 --obstacles 3' tools/profile_round.sh ${R}_cfg1 > $out/prof_cfg1.log 2>&1
  BENCH_ARGS='--horizon 80 --obstacles 20 --moving' tools/profile_round.sh ${R}_cfg4 > $out/prof_cfg4.log 2>&1
  BENCH_ARGS='--batch 8192' tools/profile_round.sh ${R}_b8192 > $out/prof_b8192.log 2>&1
  BENCH_ARGS='--horizon 80 --obstacles 20 --moving --batch 8192 --cond-N 10' tools/profile_round.sh ${R}_cond > $out/prof_cond.log 2>&1   # (usv_qp_cond)
  tail -3 $out/prof.log
fi
