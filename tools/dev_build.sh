#!/bin/bash
# Development build: only the kernels the bench workloads run (usv_model_pf_ca / usv_model_guidance_ca1, K <= 16, diagonal
# Hessian, packed box rows) -> build_ab/libusvmpc_<tag>.so in ~1 min instead of ~3.  Use with USVMPC_LIB=...
# usage: tools/dev_build.sh <tag> [extra hipcc flags]
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build_ab
C=mpc_collisionavoidance_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DUSV_BENCH_ONLY -I$C/gfx950 -I$C "$@" -o build_ab/libusvmpc_$tag.so $C/usvmpc.hip
# the hand-placed DPP instructions of THIS build are checked like the shipped library's (a dev library is loadable via USVMPC_LIB)
case " $* " in *USV_FUSED_DPP_FMA=0*) exit 0;; esac
python3 -m mpc_collisionavoidance_amd.dpp_check build_ab/libusvmpc_$tag.so || { rm -f build_ab/libusvmpc_$tag.so; exit 1; }
