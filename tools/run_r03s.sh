cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do for lib in withpi nopi; do
  USVMPC_LIB=$PWD/build_ab/libusvmpc_$lib.so timeout 600 python bench.py --cpu-sample 0 --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'qp', round(d['roofline']['kernel_ms']['usv_qp_rti'], 2))"
done; done
