#!/bin/bash
# where the tail of the headline launch goes: the main launch and the follow-up launch timed separately (rocprofv3 kernel trace) for a
# ladder of handover_iter values; handover_iter = 1 ends the main launch one pass after its queue has run dry
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_run4; mkdir -p $out
for hv in 0 1 8 16 24; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$hv -o t -- python tools/handover_probe.py usv_model_pf_ca $hv > $out/probe_$hv.txt 2>&1
  echo "== handover_iter $hv"; tail -1 $out/probe_$hv.txt
  for fn in $(find $out/trace_$hv -name "*kernel_stats.csv"); do grep -E "usv_qp_rti|usv_qp_resume|usv_linearize|Name" $fn | cut -c1-200; done
done
timeout 300 python -m pytest tests/test_gpu_handover.py -q -x 2>&1 | tail -3
