#!/bin/bash
# Development build: only the kernels the bench workloads run (usv_model_pf_ca / usv_model_guidance_ca1, K <= 16, diagonal
# Hessian, packed box rows) -> build_ab/libusvmpc_<tag>.so; three translation units in parallel (usvmpc.hip "Build parts": the C ABI
# and one per model), ~1.5 min.  Use with USVMPC_LIB=...
# usage: tools/dev_build.sh <tag> [extra hipcc flags]
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build_ab/obj_$tag
C=mpc_collisionavoidance_amd/csrc
pids=()
for part in 0 2 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUSV_BENCH_ONLY -DUSV_PART=$part $([ $part != 0 ] && echo -DUSV_COND_SEPARATE) -I$C/gfx950 -I$C "$@" -c -o build_ab/obj_$tag/p$part.o $C/usvmpc.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "hipcc failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_ab/libusvmpc_$tag.so build_ab/obj_$tag/p0.o build_ab/obj_$tag/p2.o build_ab/obj_$tag/p4.o || exit 1
rm -rf build_ab/obj_$tag
# the hand-placed DPP instructions of THIS build are checked like the shipped library's (a dev library is loadable via USVMPC_LIB)
python3 -m mpc_collisionavoidance_amd.dpp_check build_ab/libusvmpc_$tag.so || { rm -f build_ab/libusvmpc_$tag.so; exit 1; }
