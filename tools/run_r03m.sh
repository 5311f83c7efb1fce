cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03m; mkdir -p $O
export USVMPC_LIB=$PWD/build_ab/libusvmpc_pipe.so
timeout 900 python tools/pipeline_stress.py usv_model_pf_ca 30 > $O/stress_m2.txt 2>&1; echo "rc $?" >> $O/stress_m2.txt; grep -E "pipeline|BIT|MISMATCH|rc" $O/stress_m2.txt
for rep in 1 2; do for v in 0 1; do
  timeout 600 python bench.py --cpu-sample 0 --steps 20 --option pipeline_linearize=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('m2 pipeline_linearize=$v', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'qp', round(d['roofline']['kernel_ms']['usv_qp_rti'], 2), 'lin(main stream)', round(d['roofline']['kernel_ms']['usv_linearize'], 2))"
done; done > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 600 python bench.py --cpu-sample 0 --steps 20 --model usv_model_guidance_ca1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('m1', round(d['value']), d['roofline']['kernel_ms'])"
