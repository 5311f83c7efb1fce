#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_run3; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_handover.py -q -x -s > $out/pytest_handover.log 2>&1; echo "rc $?" >> $out/pytest_handover.log
tail -25 $out/pytest_handover.log
timeout 600 python tools/handover_probe.py usv_model_pf_ca > $out/handover_probe_m2.txt 2>&1; cat $out/handover_probe_m2.txt
timeout 600 python tools/handover_probe.py usv_model_guidance_ca1 0 16 12 8 0 > $out/handover_probe_m1.txt 2>&1; cat $out/handover_probe_m1.txt
timeout 900 python -m pytest tests/test_gpu_closed_loop.py tests/test_parity_outliers.py tests/test_gpu_wide.py -q -m gpu > $out/pytest_fixed.log 2>&1; tail -5 $out/pytest_fixed.log
