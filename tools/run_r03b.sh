# round 3, second GPU call: aux plane in LDS A/B (dev build), the re-gated closed-loop tests, the shim's per-tick overhead
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03b; mkdir -p $O
ab() { # tag, bench args
  tag=$1; shift
  for v in 0 1 0 1; do
    USVMPC_LIB=$PWD/build_ab/libusvmpc_aux.so timeout 600 python bench.py --cpu-sample 0 --steps 10 --option aux_in_lds=$v "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag aux_in_lds=$v', round(d['value']), 'qp ms', round(d['roofline']['kernel_ms']['usv_qp_rti'], 2), 'lin ms', round(d['roofline']['kernel_ms']['usv_linearize'], 2), 'iters', round(d['workload_stats']['qp_iter_mean'], 2), 'fail', d['workload_stats']['status_nonzero_frac'])"
  done
}
ab m2 > $O/ab_aux.txt 2>&1
ab m1 --model usv_model_guidance_ca1 >> $O/ab_aux.txt 2>&1
ab m2k9 --obstacles 9 >> $O/ab_aux.txt 2>&1
cat $O/ab_aux.txt
timeout 1200 python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_parity.py -m gpu -q -x -s -k "closed_loop or survey" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "survey parity|^usv_model|passed|failed|rc " $O/pytest.log | tail -20
g++ -O1 -std=c++17 -Impc_collisionavoidance_amd/csrc/shim/include -o /tmp/shim_harness tests/shim_harness.cpp -Lmpc_collisionavoidance_amd/csrc/shim -lacados_ocp_solver_usv_model_guidance_ca1 -Wl,-rpath,$PWD/mpc_collisionavoidance_amd/csrc/shim && /tmp/shim_harness 40 | tail -2 > $O/shim_timing.txt 2>&1
cat $O/shim_timing.txt
