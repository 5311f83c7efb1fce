# round 3: HBMPACK A/B (soft-row planes of usv_model_guidance_ca1 kept as one stream in HBM) - dev build vs the shipped library
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03t; mkdir -p $O
one() { # label, lib, bench args
  label=$1; lib=$2; shift; shift
  USVMPC_LIB=$lib timeout 600 python bench.py --cpu-sample 0 --steps 10 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', round(d['value']), 'qp ms', round(d['roofline']['kernel_ms']['usv_qp_rti'], 2), 'lin ms', round(d['roofline']['kernel_ms']['usv_linearize'], 2), 'iters', round(d['workload_stats']['qp_iter_mean'], 2), 'fail', d['workload_stats']['status_nonzero_frac'], 'parity', (d.get('parity') or {}).get('rel_err_per_instance'))"
}
for r in 1 2; do
  one "m1 head   " $PWD/mpc_collisionavoidance_amd/csrc/libusvmpc.so --model usv_model_guidance_ca1
  one "m1 hbmpack" $PWD/build_ab/libusvmpc_hbmpack.so --model usv_model_guidance_ca1
done > $O/ab.txt 2>&1
one "m1k5 head   " $PWD/mpc_collisionavoidance_amd/csrc/libusvmpc.so --model usv_model_guidance_ca1 --obstacles 5 >> $O/ab.txt 2>&1
one "m1k5 hbmpack" $PWD/build_ab/libusvmpc_hbmpack.so --model usv_model_guidance_ca1 --obstacles 5 >> $O/ab.txt 2>&1
one "m2 head   " $PWD/mpc_collisionavoidance_amd/csrc/libusvmpc.so >> $O/ab.txt 2>&1
one "m2 hbmpack" $PWD/build_ab/libusvmpc_hbmpack.so >> $O/ab.txt 2>&1
cat $O/ab.txt
USVMPC_LIB=$PWD/build_ab/libusvmpc_hbmpack.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "(guidance or survey) and not config4" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
