#!/bin/bash
# same-box A/B: idle rows parked out of the buffer window (libusvmpc.so) against the same library without (build_ab/libusvmpc_r05_nopark.so)
# and round 4's head (build_ab/libusvmpc_r04_head.so)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_run5; mkdir -p $out
for rep in 1 2 3; do
  for lib in build_ab/libusvmpc_r05_park2.so mpc_collisionavoidance_amd/csrc/libusvmpc.so; do
    USVMPC_LIB=$PWD/$lib python build_ab/bench_r04.py --cpu-sample 0 > $out/ab_$(basename $lib .so)_$rep.json 2> $out/ab_$(basename $lib .so)_$rep.err
    USVMPC_LIB=$PWD/$lib python build_ab/bench_r04.py --cpu-sample 0 --model usv_model_guidance_ca1 > $out/abm1_$(basename $lib .so)_$rep.json 2>> $out/ab_$(basename $lib .so)_$rep.err
  done
done
for f in $out/ab*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel_ms'])"; done
timeout 1200 python -m pytest tests/test_gpu_handover.py tests/test_gpu_parity.py tests/test_gpu_api.py -q -x 2>&1 | tail -4
