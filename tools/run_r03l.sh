# round 3: validation of the pipelined lineariser - bit-identity on the headline workload over 60 ticks (both models), A/B timing, the -m gpu suite
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python tools/pipeline_stress.py usv_model_pf_ca 60 > $O/stress_m2.txt 2>&1; echo "rc $?" >> $O/stress_m2.txt; tail -9 $O/stress_m2.txt
timeout 900 python tools/pipeline_stress.py usv_model_guidance_ca1 60 > $O/stress_m1.txt 2>&1; echo "rc $?" >> $O/stress_m1.txt; tail -9 $O/stress_m1.txt
for rep in 1 2; do for v in 0 1; do
  timeout 600 python bench.py --cpu-sample 0 --steps 20 --option pipeline_linearize=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('m2 pipeline_linearize=$v', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'qp', round(d['roofline']['kernel_ms']['usv_qp_rti'], 2), 'lin(main stream)', round(d['roofline']['kernel_ms']['usv_linearize'], 2))"
done; done > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
