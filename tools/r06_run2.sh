cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run2; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -40 $out/pytest.log
