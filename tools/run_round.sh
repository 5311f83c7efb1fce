#!/bin/bash
# The round's GPU evidence in one gpurun call: the -m gpu suite, tools/profile_round.sh for the bench lines that get a rocprofv3 set
# (BASELINE configs[2] both models, configs[1], configs[4]) and the plain (un-profiled) bench lines.
# usage (through gpurun): tools/run_round.sh <round tag, e.g. r04_b>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=$1; out=gpurun_out/$R; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tools/profile_round.sh $R > $out/prof.log 2>&1
BENCH_ARGS='--model usv_model_guidance_ca1' tools/profile_round.sh ${R}_m1 > $out/prof_m1.log 2>&1
BENCH_ARGS='--batch 1024 --horizon 20 --obstacles 3' tools/profile_round.sh ${R}_cfg1 > $out/prof_cfg1.log 2>&1
BENCH_ARGS='--horizon 80 --obstacles 20 --moving' tools/profile_round.sh ${R}_cfg4 > $out/prof_cfg4.log 2>&1
python bench.py > $out/bench_plain.json 2> $out/bench_plain.err; echo "rc $?" >> $out/bench_plain.err
python bench.py --workload survey-verbatim > $out/bench_survey_verbatim_plain.json 2> $out/bench_survey_verbatim_plain.err; echo "rc $?" >> $out/bench_survey_verbatim_plain.err
python bench.py --workload survey-verbatim --model usv_model_guidance_ca1 --cpu-sample 0 > $out/bench_survey_verbatim_m1_plain.json 2>/dev/null
python bench.py --oracle-opt itref_corr_max=2 > $out/bench_oracle_itref2_plain.json 2>/dev/null
python bench.py --oracle-opt cond_pred_corr=1 > $out/bench_oracle_cpc_device_plain_plain.json 2>/dev/null   # (exits 4: the option on ONE side only - the size of the unpinned-parity uncertainty)
python bench.py --option cond_pred_corr=1 --oracle-opt cond_pred_corr=1 > $out/bench_cpc_both_sides_plain.json 2>/dev/null
python bench.py --cpu-sample 0 --model usv_model_guidance_ca1 > $out/bench_m1_plain.json 2>/dev/null
python bench.py --cpu-sample 0 --model usv_model --horizon 20 > $out/bench_m0_plain.json 2>/dev/null
python bench.py --cpu-sample 0 --batch 1024 --horizon 20 --obstacles 3 > $out/bench_cfg1_plain.json 2>/dev/null
python bench.py --cpu-sample 0 --batch 1024 --horizon 20 --obstacles 3 --option wide=0 > $out/bench_cfg1_throughput_mapping_plain.json 2>/dev/null
python bench.py --cpu-sample 0 --batch 1 --horizon 40 --obstacles 10 > $out/bench_one_instance_plain.json 2>/dev/null
for m in usv_model_pf_ca usv_model_guidance_ca1; do python tools/latency_probe.py $m 20 3 1,64,512,1024,2048; python tools/latency_probe.py $m 40 10 1,64,256,512; done > $out/latency_probe.txt 2>&1
python tools/latency_probe.py usv_model_guidance_ca1 100 8 1,16,128,1024 >> $out/latency_probe.txt 2>&1   # the reference node's own shape: N = 100, K = 8
python tools/latency_probe.py usv_model_pf_ca 100 4 1,128,1024 >> $out/latency_probe.txt 2>&1
python tools/latency_probe.py usv_model 20 0 1,64,1024,2048 >> $out/latency_probe.txt 2>&1   # BASELINE configs[0]'s OCP (one instance) and batches of it
python tools/latency_probe.py usv_model_pf_ca 80 20 1,64,256,512 >> $out/latency_probe.txt 2>&1   # BASELINE configs[4]'s OCP (two obstacle chunks)
python tools/latency_probe.py usv_model_guidance_ca1 80 20 1,64,256 >> $out/latency_probe.txt 2>&1   # (two chunks of soft rows)
# policy audit: what the library picks by default against the forced placements (last column), 13 shapes x 7 batch sizes
for m in usv_model_pf_ca usv_model_guidance_ca1; do for a in "20 3" "40 10" "40 20" "80 20" "100 8" "60 4"; do python tools/latency_probe.py $m $a 64,256,512,1024,2048,4096,8192; done; done > $out/policy_audit.txt 2>&1
python tools/latency_probe.py usv_model 20 0 64,256,1024,2048,4096,8192 >> $out/policy_audit.txt 2>&1
# (the single-instance figures the docs quote: a second take, the worse of the two is what gets quoted - VERDICT r04 next 8)
for m in usv_model_pf_ca usv_model_guidance_ca1; do python tools/latency_probe.py $m 20 3 1; python tools/latency_probe.py $m 40 10 1; done > $out/latency_probe_take2.txt 2>&1
python tools/latency_probe.py usv_model_guidance_ca1 100 8 1 >> $out/latency_probe_take2.txt 2>&1
python -m pytest tests/test_shim.py -m gpu -q -s 2>&1 | grep timing >> $out/latency_probe.txt
python bench.py --cpu-sample 0 --horizon 80 --obstacles 20 --moving > $out/bench_cfg4_b65536_plain.json 2>/dev/null
python bench.py --cpu-sample 0 --horizon 80 --obstacles 20 --moving --batch 8192 > $out/bench_cfg4_b8192_per_gpu_plain.json 2>/dev/null
python bench.py --cpu-sample 0 --horizon 80 --obstacles 20 --moving --batch 8192 --cond-N 10 > $out/bench_cfg4_b8192_condN10_plain.json 2>/dev/null
tail -3 $out/pytest.log; for f in $out/bench*_plain.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],2))"; done
