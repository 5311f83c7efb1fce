cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run4; mkdir -p $out
timeout 900 python tools/co_diag.py usv_model_pf_ca 20 3 12000 8 3 > $out/co_diag_a.txt 2>&1; cat $out/co_diag_a.txt
