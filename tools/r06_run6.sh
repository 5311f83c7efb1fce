cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run6; mkdir -p $out
timeout 900 python tools/outlier_fixture.py 2048 10 $out/parity_tail_balance.npz BALANCE > $out/fixture.log 2>&1; tail -18 $out/fixture.log
timeout 900 python -m pytest tests/test_gpu_closed_loop.py -m gpu -q -s -k "closed_loop" > $out/closed_loop.log 2>&1; grep "usv_model\|passed\|failed" $out/closed_loop.log | cut -c1-400
