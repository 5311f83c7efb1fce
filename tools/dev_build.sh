#!/bin/bash
# Development build: only the kernels the bench workloads run (usv_model_pf_ca / usv_model_guidance_ca1, K <= 16, diagonal
# Hessian, packed box rows) -> build_ab/libusvmpc_<tag>.so in ~1 min instead of ~3.  Use with USVMPC_LIB=...
# usage: tools/dev_build.sh <tag> [extra hipcc flags]
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build_ab
C=mpc_collisionavoidance_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DUSV_BENCH_ONLY -I$C/gfx950 -I$C "$@" -o build_ab/libusvmpc_$tag.so $C/usvmpc.hip
