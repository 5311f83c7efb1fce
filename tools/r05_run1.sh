#!/bin/bash
# round 5, GPU call 1: do the mappings return the same bits under fp contract(on)?  and what does it cost on the headline (same-box A/B)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_run1; mkdir -p $out
python tools/bitident_probe.py > $out/bitident.txt 2>&1
for rep in 1 2; do
  for lib in build_ab/libusvmpc_r04_head.so mpc_collisionavoidance_amd/csrc/libusvmpc.so; do
    USVMPC_LIB=$PWD/$lib python build_ab/bench_r04.py --cpu-sample 0 > $out/ab_$(basename $lib .so)_$rep.json 2> $out/ab_$(basename $lib .so)_$rep.err
  done
done
for f in $out/ab_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel_ms'])"; done
tail -50 $out/bitident.txt
