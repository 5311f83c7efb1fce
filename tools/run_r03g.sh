# round 3: A/B of the packed-matrix load policy and the lineariser's rsqrt forms (dev builds), ref-vector test on the device
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03g; mkdir -p $O
for rep in 1 2; do for lib in base matnt; do
  USVMPC_LIB=$PWD/build_ab/libusvmpc_$lib.so timeout 600 python bench.py --cpu-sample 0 --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['value']), 'qp ms', round(d['roofline']['kernel_ms']['usv_qp_rti'], 2), 'lin ms', round(d['roofline']['kernel_ms']['usv_linearize'], 2))"
done; done > $O/ab.txt 2>&1
cat $O/ab.txt
USVMPC_LIB=$PWD/build_ab/libusvmpc_base.so timeout 900 python -m pytest tests/test_ref_vectors.py tests/test_gpu_parity.py -m gpu -q -k "not full_size and not m0_plumbing and not acados_style" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
