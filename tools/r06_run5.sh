cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run5; mkdir -p $out
timeout 900 python tools/co_probe.py usv_model_pf_ca 40 10 65536 20 > $out/co_probe_65536.txt 2>&1; tail -1 $out/co_probe_65536.txt
timeout 600 python tools/co_probe.py usv_model_pf_ca 40 10 32768 20 > $out/co_probe_32768.txt 2>&1; tail -1 $out/co_probe_32768.txt
timeout 600 python tools/co_probe.py usv_model_pf_ca 40 10 8192 40 > $out/co_probe_8192.txt 2>&1; tail -1 $out/co_probe_8192.txt
timeout 600 python tools/co_probe.py usv_model_pf_ca 40 10 16384 30 > $out/co_probe_16384.txt 2>&1; tail -1 $out/co_probe_16384.txt
timeout 600 python tools/co_probe.py usv_model_pf_ca 40 10 4096 40 > $out/co_probe_4096.txt 2>&1; tail -1 $out/co_probe_4096.txt
