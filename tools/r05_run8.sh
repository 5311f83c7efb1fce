#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_run8; mkdir -p $out
timeout 1500 python -m pytest tests/test_sqp_options.py tests/test_gpu_wide.py tests/test_kkt_certify.py -q -x -m gpu > $out/pytest.log 2>&1; tail -15 $out/pytest.log
