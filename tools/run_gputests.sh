cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$1
timeout 2400 python -m pytest tests -m gpu -q ${2:-} > gpurun_out/$1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/$1/pytest.log
tail -25 gpurun_out/$1/pytest.log
