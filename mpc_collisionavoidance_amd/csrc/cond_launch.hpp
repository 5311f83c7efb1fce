// cond_launch.hpp — host interface of the partial-condensing kernels (cond_kernels.hip), so that they can be compiled as a
// translation unit of their own (the Riccati kernels of usvmpc.hip take minutes to build, these seconds).
#pragma once
#include "cond_dims.hpp"
#include <hip/hip_runtime.h>
#include <string>

namespace usv {

// sizes of the condensed QP (D.nt: the team size chosen), dynamic LDS (bytes) and resident workgroups per CU of the kernel for this model;
// 0 or a USVMPC_E_* code
int cond_prepare(int model, int kch, const DevSpec &S, int N2, CondDims &D, size_t &lds_bytes, int &blocks_per_cu, std::string &err);
// one launch (Dh: the host's copy of the sizes - team size, and the instantiation made for the shape where one exists): `teams` workgroups, each with its scratch area of D.total doubles, pulling the B instances from P.queue; 0 or -1 (no kernel)
int cond_run(int model, int kch, const CondDims &Dh, hipStream_t st, long teams, size_t lds_bytes, const DevPtrs &P, const CondDims *dD, double *scratch, int B);

} // namespace usv
