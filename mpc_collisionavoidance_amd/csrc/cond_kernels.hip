// cond_kernels.hip — the partial-condensing kernels (cond_ipm.hpp) and their launch, per model.
// Compiled on its own for the shipped library (__graft_entry__.build_hip) or included at the end of usvmpc.hip.
#include "gfx950/lanes.hpp"

#include "../../include/usvmpc.h"
#include "cond_launch.hpp"
#include "cond_ipm.hpp"
#include "models.hpp"
#ifdef USV_GEN_MODEL_HEADER
#include USV_GEN_MODEL_HEADER
#endif

namespace usv {

// Team size and register budget (profiles/r03_condensing_ab.txt): the kernel is bound by latency (LDS round trips, barriers), i.e. by how many
// waves a CU holds.  With 30-variable blocks (usv_model_pf_ca, 8 stages per block) LDS allows three teams of ~50 KB per CU: 256-thread
// teams at 168 registers (three waves per SIMD) beat 64 / 128 threads at any budget (233 against 285 .. 425 ms at 8192 instances).  With
// 13-variable blocks (usv_model_guidance_ca1, 5 stages per block) a team needs a few KB and twelve 64-thread teams fit: 188 ms against
// 378 ms with 256-thread teams at 65536 instances.  Both are built; cond_prepare takes the one with more resident waves per CU.
#ifndef USV_COND_MINWAVES // waves per SIMD the kernels with hard obstacle rows (or none) are compiled for (register budget 512 / that)
#define USV_COND_MINWAVES 4
#endif
// Round 6 (docs/rounds/r06.md section 7): with the thread index made opaque once per block (CondIpm::forget_tid) the kernels want 155 - 173
// registers instead of 386, LDS per team of the 30-variable blocks went from 53 to 40 KB, and FOUR 256-thread teams per CU at 128 registers
// (11 spilled) beat three at 168: 126 against 171 ms at 8192 instances (205 ms before the round).  The soft-row kernels (usv_model_guidance_ca1:
// twelve or more 64-thread teams per CU) spill 150 registers at 128 and stay at three waves per SIMD.
template <bool SOFT> constexpr int cond_minwaves() { return SOFT ? 3 : USV_COND_MINWAVES; }

// One instance per workgroup of NT threads, the condensed block's matrices in LDS, the instance's condensed QP in the
// workgroup's scratch area in HBM; workgroups pull further instances from the queue as they finish.
// MB, NXR: stages per block and touched states the instantiation is made for (0, 0: any - CondBlkSizes).
template <class M, int KCH, bool SOFT, int NT, int MB, int NXR>
__global__ void __launch_bounds__(NT, cond_minwaves<SOFT>()) usv_qp_cond(DevPtrs P, const CondDims *Dp, double *scratch, int nB, int queue0)
{
    extern __shared__ double cond_lds[];
    __shared__ int nxt;
    CondIpm<M, KCH, SOFT, CondTeam<NT>, MB, NXR> q(P, *Dp, scratch + (long)blockIdx.x * Dp->total, cond_lds);
    long g = blockIdx.x;
    while (g < nB) {
        q.solve(g);
        if (threadIdx.x == 0) nxt = queue0 + atomicAdd(P.queue, 1);
        __syncthreads();
        g = nxt;
        __syncthreads();
    }
}

namespace {

// the shape a 256-thread instantiation with compile-time sizes exists for (0: none): usv_model_pf_ca with 8 stages per block and its 7 touched
// states (5 bounded + the position) - BASELINE configs[4] with qp_cond_N = 10
template <class M, bool SOFT> struct CondFixed { static constexpr int MB = 0, NXR = 0; };
#if !defined(USV_GEN_ONLY)
template <> struct CondFixed<ModelM2, false> { static constexpr int MB = 8, NXR = 7; };
#endif

template <class M, int KCH, bool SOFT, int NT>
int occupancy_of(const DevSpec &S, int N2, CondDims &D, size_t &lds, int &nb)
{
    auto kern = &usv_qp_cond<M, KCH, SOFT, NT, 0, 0>; // (the fixed-shape instantiations use the same LDS and no more registers)
    nb = 0;
    if (!cond_dims(S, M::NX, M::NU, M::IPX, M::IPY, N2, NT, SOFT, D)) return USVMPC_E_ARG;
    lds = (size_t)D.lds_doubles * sizeof(double);
    if (lds > 160u * 1024u) return 0; // (nb = 0: does not fit)
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NT, lds) != hipSuccess)
        return USVMPC_E_HIP;
    return 0;
}

template <class M, int KCH, bool SOFT>
int prepare_for(const DevSpec &S, int N2, CondDims &D, size_t &lds, int &nb, std::string &err)
{
    CondDims D64, D256;
    size_t l64 = 0, l256 = 0;
    int n64 = 0, n256 = 0;
    const int r64 = occupancy_of<M, KCH, SOFT, 64>(S, N2, D64, l64, n64), r256 = occupancy_of<M, KCH, SOFT, 256>(S, N2, D256, l256, n256);
    if (r64 == USVMPC_E_ARG || r256 == USVMPC_E_ARG) { err = "qp_cond_N must lie in 1..N-1 and a condensed stage may have at most 64 variables (nx + ceil(N / qp_cond_N) nu)"; return USVMPC_E_ARG; }
    if (r64 || r256) { err = "partial condensing: the kernel cannot be launched with this much LDS"; return USVMPC_E_HIP; }
    if (n64 < 1 && n256 < 1) { err = "partial condensing: the condensed block does not fit in LDS (block too large)"; return USVMPC_E_ARG; }
    // (about as many resident waves with the small teams - at least three quarters: more instances in flight, cheaper barriers - measured 2x on M1)
    if (4 * n64 >= 3 * 4 * n256) { D = D64; lds = l64; nb = n64; }
    else { D = D256; lds = l256; nb = n256; }
    return 0;
}

template <class M, int KCH, bool SOFT>
int run_for(const CondDims &Dh, hipStream_t st, long teams, size_t lds, const DevPtrs &P, const CondDims *dD, double *scratch, int B)
{
    const int nt = Dh.nt;
    if (nt == 64) hipLaunchKernelGGL((usv_qp_cond<M, KCH, SOFT, 64, 0, 0>), dim3((unsigned)teams), dim3(64), lds, st, P, dD, scratch, B, (int)teams);
    else if (CondFixed<M, SOFT>::MB > 0 && Dh.Mb == CondFixed<M, SOFT>::MB && Dh.nxr == CondFixed<M, SOFT>::NXR) {
        auto kern = &usv_qp_cond<M, KCH, SOFT, 256, CondFixed<M, SOFT>::MB, CondFixed<M, SOFT>::NXR>;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        hipLaunchKernelGGL(kern, dim3((unsigned)teams), dim3(256), lds, st, P, dD, scratch, B, (int)teams);
    } else hipLaunchKernelGGL((usv_qp_cond<M, KCH, SOFT, 256, 0, 0>), dim3((unsigned)teams), dim3(256), lds, st, P, dD, scratch, B, (int)teams);
    return 0;
}

} // namespace

// (usv_model has no obstacle rows, usv_model_pf_ca hard ones, usv_model_guidance_ca1 soft ones)
#define USV_COND_DISPATCH(CALL)                                                                                      \
    switch (model) {                                                                                                 \
    USV_COND_BUILTIN(CALL)                                                                                           \
    USV_COND_GENERATED(CALL)                                                                                         \
    }

#if defined(USV_GEN_ONLY)
#define USV_COND_BUILTIN(CALL)
#elif defined(USV_BENCH_ONLY)
#define USV_COND_BUILTIN(CALL) case USVMPC_MODEL_PF_CA: if (kch <= 1) return CALL(ModelM2, 1, false); break; case USVMPC_MODEL_GUIDANCE_CA1: if (kch <= 1) return CALL(ModelM1, 1, true); break;
#else
#define USV_COND_BUILTIN(CALL)                                                                                       \
    case USVMPC_MODEL_USV: return CALL(ModelM0, 0, false);                                                           \
    case USVMPC_MODEL_GUIDANCE_CA1: return kch <= 1 ? CALL(ModelM1, 1, true) : CALL(ModelM1, 2, true);               \
    case USVMPC_MODEL_PF_CA: return kch <= 1 ? CALL(ModelM2, 1, false) : CALL(ModelM2, 2, false);
#endif
#if defined(USV_GEN_MODEL_HEADER) && !defined(USV_BENCH_ONLY)
#define USV_COND_GENERATED(CALL) case USVMPC_MODEL_GENERATED: return CALL(ModelGen, USV_GEN_KCH, (USV_GEN_SOFT != 0));
#else
#define USV_COND_GENERATED(CALL)
#endif

int cond_prepare(int model, int kch, const DevSpec &S, int N2, CondDims &D, size_t &lds_bytes, int &blocks_per_cu, std::string &err)
{
#define USV_COND_PREP(M, K, SF) prepare_for<M, K, SF>(S, N2, D, lds_bytes, blocks_per_cu, err)
    USV_COND_DISPATCH(USV_COND_PREP)
#undef USV_COND_PREP
    err = "partial condensing: no kernel for this model in this library";
    return USVMPC_E_ARG;
}

int cond_run(int model, int kch, const CondDims &Dh, hipStream_t st, long teams, size_t lds_bytes, const DevPtrs &P, const CondDims *dD, double *scratch, int B)
{
#define USV_COND_RUN(M, K, SF) run_for<M, K, SF>(Dh, st, teams, lds_bytes, P, dD, scratch, B)
    USV_COND_DISPATCH(USV_COND_RUN)
#undef USV_COND_RUN
    return -1;
}

} // namespace usv

#if defined(USV_COND_TIMING) // development build only (tools/cond_timing.py): cycles of a team's first thread per phase of cond_ipm.hpp
extern "C" int usvmpc_debug_cond_ticks(unsigned long long *out32, int reset)
{
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(usv::usv_cond_ticks), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(usv::usv_cond_ticks), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
