cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03o; mkdir -p $O
for i in 1 2 3; do timeout 900 python tools/pipeline_stress.py usv_model_pf_ca 40 2>&1 | grep -E "pipeline_linearize=1|BIT|MISMATCH"; done > $O/stress.txt 2>&1
timeout 900 python tools/pipeline_stress.py usv_model_guidance_ca1 40 2>&1 | grep -E "pipeline_linearize=1|BIT|MISMATCH" >> $O/stress.txt 2>&1
cat $O/stress.txt
for i in 1 2 3; do timeout 600 python bench.py --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel_ms'], d['roofline']['traffic'])"; done
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q -k pipelined 2>&1 | tail -2
