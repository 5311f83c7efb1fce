cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run3; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_handover.py tests/test_shim.py -m gpu -q -s > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -12 $out/pytest.log
timeout 900 python tools/co_probe.py usv_model_pf_ca 40 10 65536 20 > $out/co_probe_65536.txt 2>&1; tail -16 $out/co_probe_65536.txt
timeout 400 python tools/co_probe.py usv_model_pf_ca 40 10 8192 40 > $out/co_probe_8192.txt 2>&1; tail -1 $out/co_probe_8192.txt
timeout 400 python tools/co_probe.py usv_model_pf_ca 40 10 32768 20 > $out/co_probe_32768.txt 2>&1; tail -1 $out/co_probe_32768.txt
timeout 400 python tools/co_probe.py usv_model_guidance_ca1 40 10 65536 20 > $out/co_probe_m1_65536.txt 2>&1; tail -1 $out/co_probe_m1_65536.txt
