#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_run6; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_wide.py -q -x > $out/pytest_wide.log 2>&1; tail -12 $out/pytest_wide.log
for a in "usv_model_pf_ca 80 20 1,64,256,512" "usv_model_guidance_ca1 80 20 1,64,256" "usv_model_pf_ca 40 20 1,256,1024"; do timeout 600 python tools/latency_probe.py $a; done > $out/latency_two_chunks.txt 2>&1; cat $out/latency_two_chunks.txt
