#!/bin/bash
# Development aid: where the condensing kernel's time goes - builds of cond_kernels.hip that run a fixed number of IPM iterations
# (-DUSV_COND_TIMING=18) with one part switched off (-DUSV_COND_SKIP bit: 1 row pass, 2 Hessian assembly, 4 elimination, 8 reload of
# the sensitivity rows, 16 corrector backward sweep, 32 forward sweeps).  Results of such builds are garbage; only the launch time counts.
# usage (build container): tools/cond_timing.sh build ; (GPU box): tools/cond_timing.sh run
C=mpc_collisionavoidance_amd/csrc
if [ "$1" = build ]; then
  mkdir -p build_ab
  for k in 0 1 2 4 8 16 32 48; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUSV_COND_SEPARATE -DUSV_COND_TIMING=18 -DUSV_COND_SKIP=$k -I$C/gfx950 -I$C -c -o build_ab/cond_t$k.o $C/cond_kernels.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_ab/libusvmpc_t$k.so $C/build/usvmpc.o build_ab/cond_t$k.o
  done
else
  for k in 0 1 2 4 8 16 32 48; do
    USVMPC_LIB=build_ab/libusvmpc_t$k.so python bench.py --horizon 80 --obstacles 20 --moving --batch 8192 --steps 3 --warmup 1 --cpu-sample 0 --cond-N 10 2>/dev/null |
      python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip $k', j['roofline']['kernel_ms'])"
  done
fi
