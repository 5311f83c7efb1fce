// usvmpc.hip — gfx950 kernels + the C ABI of include/usvmpc.h.
//
// Two kernels per SQP-RTI iteration (and per iteration of the full SQP):
//   usv_linearize<M,KCH,..>   one 16-lane group per (instance, stage): ERK4 + forward VDE, GN gradient, the
//                             stage matrix packed into the workspace planes   (linearize.hpp)
//   usv_qp_rti<M,KCH,SOFT,..> one 16-lane group per instance: obstacle-row linearisation, Riccati-IPM QP,
//                             RTI / SQP step, NLP residual test for the full SQP  (qp_ipm.hpp)
// Both are FP64 VALU + DPP kernels (lane gathers go through the LDS crossbar, no LDS memory, no MFMA: the
// blocks are at most 16x16).
//
// Build parts.  The QP kernel templates are what takes minutes to compile, and one (model, obstacle chunks) pair has nothing in common
// with another: __graft_entry__.build() therefore compiles THIS file several times in parallel, -DUSV_PART=0 for the C ABI, the handle
// bookkeeping and the small kernels, -DUSV_PART=1 .. 5 for one launch_pair / export_pair instantiation each (explicit instantiation
// there, extern template everywhere else), and links the objects.  Without USV_PART everything is one translation unit (the generated-
// model libraries of genbuild.py, tools/dev_build.sh).
#ifndef USV_PART
#define USV_PART -1
#endif
#define USV_MAIN (USV_PART <= 0)

#include "gfx950/lanes.hpp"

#include "guidance.hpp"
#include "host_spec.hpp"
#include "linearize.hpp"
#include "models.hpp"
#include "qp_ipm.hpp"
#include "cond_launch.hpp"
#ifdef USV_GEN_MODEL_HEADER // a model generated from a symbolic definition (codegen.py): struct ModelGen
#include USV_GEN_MODEL_HEADER
#endif

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace usv;

// ---------------------------------------------------------------------------------- kernels
#ifndef USV_LIN_BLOCKS
#define USV_LIN_BLOCKS 2 // 256-thread blocks per CU the lineariser is compiled for (2: 2 waves per SIMD, 256 registers)
#endif
// MODE: 0 the whole batch; 1 speculative for the next tick beside a running QP launch; 2 fix-up of what mode 1 skipped (linearize.hpp)
template <class M, int KCH, bool SOFT, bool MULTI, int MODE = 0>
__global__ void __launch_bounds__(256, USV_LIN_BLOCKS) usv_linearize(DevPtrs P, long ngroups)
{
    const long gid = lanes::group_linear();
    if (gid >= ngroups) return; // ngroups is a multiple of 4: whole waves leave together
    Linearize<M, KCH, SOFT, MULTI, MODE>::run(P, gid);
}

#ifndef USV_QP_WAVES
#define USV_QP_WAVES 2 // waves per SIMD the QP kernel is compiled for (register budget 512 / USV_QP_WAVES)
#endif
// Two obstacle chunks (K > 16) or soft state bounds keep twice the row state in flight: those instantiations get the whole
// register file of a SIMD (one wave, 512 registers) instead of spilling 100-280 registers at two waves - the kernel is
// issue-bound, a second wave buys little and scratch traffic costs a lot.
template <int KCH, bool SOFTBOX>
constexpr int qp_waves() { return (KCH >= 2 || SOFTBOX) ? 1 : USV_QP_WAVES; }

// rows: instances a wave starts on (4; fewer when the workspace lives in LDS and only that many fit: LDSWS) - row r of block b
// starts on group b * rows + r, surplus rows stay idle.
// AUXLDS: the aux plane of the rows' instances in the wave's LDS instead of HBM (qp_ipm.hpp)
// WIDE: the latency mapping - one instance per wave (rows = 1: rows 1 - 3 are handed row 0's group and share its LDS region; in the
// sweeps they take over the row work of the neighbouring stages: qp_ipm.hpp)
// WW > 1 (with WIDE): a workgroup of WW waves per instance (qp_ipm.hpp)
// CPC: with HPIPM's conditional predictor-corrector built in (qp_ipm.hpp; option "cond_pred_corr" switches the test) - every kernel of the library
template <class M, int KCH, bool SOFT, bool HDIAG, bool PACK, bool SOFTBOX, bool LDSWS = false, bool MERGE = false, bool AUXLDS = false, bool WIDE = false,
          int WW = 1, bool CPC = true>
__global__ void __launch_bounds__(64 * WW, ((LDSWS || WIDE) ? 1 : qp_waves<KCH, SOFTBOX>())) usv_qp_rti(DevPtrs P, long ngroups, int phase, int queue0, int rows)
{
    const int row = (int)(threadIdx.x >> 4);
    const long g0 = (long)blockIdx.x * rows;
    if (g0 >= ngroups) return;
    // (hand-over with a co-resident follow-up kernel: that kernel goes to work only once every workgroup of THIS launch is on the device)
    if constexpr (!WIDE && !LDSWS) { if (P.co_ctl != nullptr && threadIdx.x == 0) lanes::count_one(P.co_ctl + 2); }
    const bool has = row < rows;
    QpIpm<M, KCH, SOFT, HDIAG, PACK, SOFTBOX, LDSWS, MERGE, AUXLDS, WIDE, WW, CPC> q(P, has ? g0 + row : g0, has ? row : -1);
    q.solve(phase, queue0);
}
// The wide instantiations: packed layouts with one or two obstacle chunks, and the layouts with the box rows in planes of their own (UNPACKED:
// no obstacle rows, obstacle rows that leave the box rows no idle lanes, or - SOFTBOX - soft state bounds).
// (LDSWS: the solver's planes in LDS - false: in HBM, for horizons that do not fit a CU's LDS and for the launches of a full SQP)
template <class M, int KCH, bool SOFT, bool MERGE, bool LDSWS = true, int WW = 1, bool SOFTBOX = false, bool UNPACKED = false>
constexpr auto wide_kernel()
{
    // (unpacked rows beside obstacle rows and soft state bounds: one wave per instance; four waves are built for the packed layouts - one or two
    // obstacle chunks, K = 17 .. 32 being BASELINE configs[4]'s OCP - and the layout without obstacle rows)
    if constexpr (SOFTBOX || (UNPACKED && KCH > 0)) {
        if constexpr (WW == 1 && !MERGE) return &usv_qp_rti<M, KCH, SOFT, true, false, SOFTBOX, LDSWS, false, false, true, 1>;
        else return (decltype(&usv_qp_rti<M, KCH, SOFT, true, false, SOFTBOX, true, false>))nullptr;
    } else if constexpr (KCH == 1 || KCH == 2) return &usv_qp_rti<M, KCH, SOFT, true, true, false, LDSWS, MERGE, false, true, WW>;
    else if constexpr (KCH == 0 && !MERGE) return &usv_qp_rti<M, KCH, SOFT, true, false, false, LDSWS, false, false, true, WW>; // (no obstacle rows: box rows in their own planes)
    else return (decltype(&usv_qp_rti<M, KCH, SOFT, true, (KCH > 0), false, true, MERGE>))nullptr;
}
// The follow-up of a launch that handed long runners over (QpIpm::suspend, option "handover_iter"): one wave per suspended instance,
// on the latency mapping - over the planes in HBM the suspending row was working on, or (LDSWS, horizons that fit) after copying them into
// LDS, where a pass costs half.  Workgroup i takes entries i, i + gridDim, ... of the list; with an empty list the launch is a few microseconds.
template <class M, int KCH, bool SOFT, bool MERGE, bool LDSWS>
__global__ void __launch_bounds__(64, 1) usv_qp_resume(DevPtrs P)
{
    const int n = lanes::uniform(*P.susp_count);
    const bool co = P.co_ctl != nullptr; // (the co-resident kernel may have taken the entry - or be about to: one compare-and-swap decides)
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
        int g = P.susp_list[i];
        if (co) {
            if (threadIdx.x == 0) g = (g >= 0 && lanes::claim(P.susp_list + i, g)) ? g : -1;
            g = lanes::wave_first_i(g);
            if (g < 0) continue;
        }
        QpIpm<M, KCH, SOFT, true, (KCH > 0), false, LDSWS, MERGE, false, true, 1> q(P, (long)g, (threadIdx.x >> 4) == 0 ? 0 : -1);
        q.solve(3, -1);
    }
}
// The same follow-up BESIDE the draining launch (option "handover_co"): enqueued on a stream of its own together with the main launch, its
// workgroups come onto the device as main wavefronts leave it (75 KB of LDS and a SIMD's registers each), take a ticket each and wait - a
// bounded wait - for the list entry of that number to appear; the instance behind it is finished here while the main launch is still
// draining, instead of behind it.  A workgroup leaves when the main launch has ended and its ticket's entry never came (the list is
// complete then), when its wait runs into the spin limit, or - at once - when it finds itself on the device before every workgroup of the
// main launch is (it must not hold what a persistent launch still waits for).  Whatever is left is done by usv_qp_resume behind the main
// launch; an entry is taken by one of the two (compare-and-swap).  Scheduling only: the same sweeps over the same planes.
template <class M, int KCH, bool SOFT, bool MERGE>
__global__ void __launch_bounds__(64, 1) usv_qp_resume_co(DevPtrs P, int main_wgs, int cap, int spin_limit)
{
    int *ctl = P.co_ctl;
    {
        int ok = 0;
        if (threadIdx.x == 0) ok = lanes::observe(ctl + 2) >= main_wgs;
        if (!lanes::wave_first_i(ok)) return;
    }
    for (;;) {
        int t = 0;
        if (threadIdx.x == 0) t = atomicAdd(ctl, 1);
        t = lanes::wave_first_i(t);
        if (t >= cap) return;
        // Every lane polls the same word: one access for the wave, and NO loop inside a single-lane region.  (The first form of this kernel had
        // lane 0 alone run the ticket / poll / claim loops: on this toolchain every instance a workgroup finished after its first came out
        // wrong - with or without the main launch beside it, with or without the fences; the sweeps' code was the follow-up launch's.  The
        // single-lane regions left are an atomic each.)
        int e, spins = 0;
        for (;;) {
            e = lanes::observe(P.susp_list + t);
            if (e != -1) break;
            if (lanes::observe(ctl + 1) != 0) { e = lanes::observe(P.susp_list + t); break; } // (the main launch has ended: the list is final)
            if (++spins > spin_limit) break;
            __builtin_amdgcn_s_sleep(64);
        }
        e = lanes::wave_first_i(e);
        if (e == -1) { // nothing will come for this number (entries are dense), or the wait was given up: what is left is the launch's behind the main one
            if (spins > spin_limit && threadIdx.x == 0) lanes::count_one(ctl + 4);
            return;
        }
        int mine = 0;
        if (threadIdx.x == 0) mine = (e >= 0 && lanes::claim(P.susp_list + t, e)) ? 1 : 0;
        if (!lanes::wave_first_i(mine)) continue;
        lanes::acquire_agent(); // (the suspending wave released at agent scope before the entry went out)
        {
            QpIpm<M, KCH, SOFT, true, (KCH > 0), false, true, MERGE, false, true, 1> q(P, (long)e, (threadIdx.x >> 4) == 0 ? 0 : -1);
            q.solve(3, -1);
        }
        if (threadIdx.x == 0) lanes::count_one(ctl + 3);
    }
}
static __global__ void usv_co_done(int *ctl) { lanes::publish(ctl + 1, 1); }
using qp_resume_t = void (*)(DevPtrs);
using qp_resume_co_t = void (*)(DevPtrs, int, int, int);
template <class M, int KCH, bool SOFT, bool MERGE, bool LDSWS = false>
constexpr qp_resume_t resume_kernel()
{
    if constexpr (KCH == 1 || (KCH == 0 && !MERGE)) return &usv_qp_resume<M, KCH, SOFT, MERGE, LDSWS>;
    else return nullptr;
}
template <class M, int KCH, bool SOFT, bool MERGE>
constexpr qp_resume_co_t resume_co_kernel()
{
    if constexpr (KCH == 1 || (KCH == 0 && !MERGE)) return &usv_qp_resume_co<M, KCH, SOFT, MERGE>;
    else return nullptr;
}
// the wide kernels of one layout: [planes in LDS, planes in HBM] x [one wave, four waves per instance], and the follow-up kernel
// (nplw: planes per stage an instance keeps in LDS; ex_lds / ex_hbm: planes of the exchange area - qp_ipm.hpp NPLW, EX_N)
using qp_kernel_t = void (*)(DevPtrs, long, int, int, int);
struct WideSet { qp_kernel_t lds1, hbm1, lds4, hbm4; qp_resume_t resume, resume_lds; qp_resume_co_t resume_co; int nplw, ex_lds, ex_hbm; };
constexpr WideSet NO_WIDE = WideSet{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
template <class M, int KCH, bool SOFT, bool MERGE, bool SOFTBOX = false, bool UNPACKED = false>
constexpr WideSet wide_set()
{
    using WL = WsLayout<M, KCH, SOFT, SOFTBOX>;
    constexpr bool packed = KCH > 0 && !SOFTBOX && !UNPACKED; // (the packed layouts leave the four box planes out of the LDS map)
    return WideSet{wide_kernel<M, KCH, SOFT, MERGE, true, 1, SOFTBOX, UNPACKED>(), wide_kernel<M, KCH, SOFT, MERGE, false, 1, SOFTBOX, UNPACKED>(),
                   wide_kernel<M, KCH, SOFT, MERGE, true, 4, SOFTBOX, UNPACKED>(), wide_kernel<M, KCH, SOFT, MERGE, false, 4, SOFTBOX, UNPACKED>(),
                   (SOFTBOX || (UNPACKED && KCH > 0)) ? nullptr : resume_kernel<M, KCH, SOFT, MERGE>(),
                   (SOFTBOX || (UNPACKED && KCH > 0)) ? nullptr : resume_kernel<M, KCH, SOFT, MERGE, true>(),
                   (SOFTBOX || (UNPACKED && KCH > 0)) ? nullptr : resume_co_kernel<M, KCH, SOFT, MERGE>(),
                   WL::P_RB0 - (packed ? 4 : 0) + (SOFTBOX ? 6 : 0), wide_ex_planes(KCH, SOFTBOX), wide_ex_planes_hbm(KCH, SOFTBOX)};
}

// Multiplier read-back (usvmpc_get "lam" / "t"): the inequality multipliers and slacks of every instance's last QP, from the
// group-indexed workspace planes into instance-major arrays in acados' row order (QpIpm::export_rows).  Run on demand.
template <class M, int KCH, bool SOFT, bool PACK, bool SOFTBOX>
__global__ void __launch_bounds__(64) usv_qp_export(DevPtrs P, long ngroups)
{
    const long g0 = (long)blockIdx.x * 4;
    if (g0 >= ngroups) return;
    QpIpm<M, KCH, SOFT, true, PACK, SOFTBOX> q(P, g0 + (long)(threadIdx.x >> 4));
    q.export_rows();
}

// Instances of a launch whose QP did not converge to the IPM tolerances (qp_status != 0): counted by a kernel of its own behind the QP
// launch, per workgroup in LDS and one atomic per workgroup (usvmpc_unconverged_counts; SURVEY.md 8(d) counts converged solves).  Not inside
// QpIpm::finish(): one more counter there moved the headline kernel's register allocation (12 -> 17 spilled registers, +1.3 % per launch in a
// same-box A/B).
static __global__ void __launch_bounds__(256) usv_count_unconverged(const int *qp_status, int B, int *count, unsigned long long *total)
{
    __shared__ int loc;
    if (threadIdx.x == 0) loc = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B && qp_status[i] != 0) atomicAdd(&loc, 1);
    __syncthreads();
    if (threadIdx.x == 0 && loc != 0) { atomicAdd(count, loc); atomicAdd(total, (unsigned long long)loc); } // (total: the handle's running sum)
}

#if USV_MAIN
// full SQP bookkeeping: start of a call (everything running) and end (still running = max iterations)
__global__ void usv_sqp_begin(DevPtrs P, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { P.sqp_state[i] = -1; P.sqp_iter[i] = 0; }
}
__global__ void usv_sqp_end(DevPtrs P, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) P.status[i] = P.sqp_state[i] < 0 ? 2 : P.sqp_state[i];
}

// Closed-loop hand-over between two ticks, as the reference's callers do it on the host
// (x0 = get(1,"x"); set(0,"lbx",x0): scripts/usv_guidance_ca1/main.py:169-175): the next initial
// state is the predicted x_1 plus an optional Gaussian disturbance (the commented "Add noise"
// hooks of scripts/usv_pf_ca/main.py:181-183).  No trajectory shift, as in the reference.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void usv_advance(DevPtrs P, int nx, double sigma, unsigned long long seed, unsigned mask)
{
    const DevSpec &S = *P.spec;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)S.B * nx) return;
    const long b = i / nx;
    const int j = (int)(i - b * nx);
    double v = P.x[(b * (S.N + 1) + 1) * nx + j];
    if (sigma != 0.0 && ((mask >> j) & 1u)) {
        const unsigned long long h1 = splitmix64(seed ^ (unsigned long long)(2 * i));
        const unsigned long long h2 = splitmix64(seed ^ (unsigned long long)(2 * i + 1));
        const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740993.0);
        const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
        v += sigma * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
    const_cast<double *>(P.x0)[i] = v;
}

// Debug / test entry points: the model functions exactly as the lineariser calls them (M::fjvp: f and one Jacobian
// column per call) and the obstacle-row geometry exactly as the QP kernel evaluates it (obs_dist), on caller-supplied
// points, so that a GPU test can compare the device transcription with vectors derived from the reference's own
// model files (tests/golden/ref_model_*.npz).  One thread per (point, column).
template <class M>
__global__ void usv_debug_model(int n, const double *x, const double *u, double *f, double *J)
{
    constexpr int NX = M::NX, NU = M::NU, NZ = NX + NU;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * NZ) return;
    const int pt = i / NZ, c = i - pt * NZ;
    double xx[NX], uu[NU > 0 ? NU : 1], s[NX], su[NU > 0 ? NU : 1], ff[NX], js[NX];
    for (int j = 0; j < NX; j++) { xx[j] = x[pt * NX + j]; s[j] = (c == NU + j) ? 1.0 : 0.0; }
    for (int j = 0; j < NU; j++) { uu[j] = u[pt * NU + j]; su[j] = (c == j) ? 1.0 : 0.0; }
    M::fjvp(xx, uu, s, su, ff, js);
    for (int j = 0; j < NX; j++) {
        J[((long)pt * NX + j) * NZ + c] = js[j];   // d f_j / d z_c, z = [u; x]
        if (c == 0) f[pt * NX + j] = ff[j];
    }
}

__global__ void usv_debug_obstacle(int n, int K, const double *pos, const double *p, double *h, double *grad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int pt = i / K, k = i - pt * K;
    double d, ux, uy;
    usv::obs_dist(pos[2 * pt] - p[(long)pt * 2 * K + 2 * k], pos[2 * pt + 1] - p[(long)pt * 2 * K + 2 * k + 1], d, ux, uy);
    h[i] = d; grad[2 * i] = ux; grad[2 * i + 1] = uy;
}

// Traffic calibration for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters: streams a KNOWN number of
// workspace planes with exactly the access instruction of the solver kernels (one
// buffer_load_dwordx2 per lane per plane, 16 lanes of a group contiguous) and writes one plane.
__global__ void __launch_bounds__(64) usv_calib_stream(DevPtrs P, long ngroups, int nread)
{
    const long g = lanes::group_linear();
    if (g >= ngroups) return;
    const lanes::Planes W(P.ws, (unsigned)(ngroups * (nread + 1) * 128), lanes::Planes::lane_offset(g, nread + 1, lanes::lane()));
    double acc = 0.0;
    for (int i = 0; i < nread; i++) acc += W.ld(i);
    W.st(nread, acc);
}
#endif // USV_MAIN

// Difficulty binning: a wave carries four instances and runs until the slowest one converges, so
// instances are grouped by the IPM iteration count of their previous solve (a counting sort on the
// device, hardest first so that long-running waves start early).  Only the group -> instance map
// changes; the arithmetic of an instance does not depend on its neighbours.
constexpr int SORT_BINS = 64;

// (histogram and ranks are formed per workgroup in LDS; a workgroup then touches each global bin once - 65 536 threads
// hammering 64 global counters took 0.19 ms per kernel)
// The sort key of an instance: its IPM iteration count in the last solve - or, option "sort_two_ticks", the larger of the last TWO.
// What the launch must avoid is an instance that runs long being handed out late (it then ends with a few waves finishing it while
// the device idles: a tenth of the launch on the bench workload); one tick's count predicts the next with correlation 0.53 only.
// The two-tick key looked better in schedule simulations and measured the same on the device (profiles/r03_tail.txt): off by default.
__device__ __forceinline__ int sort_key(const int *qp_iter, const int *qp_iter_prev, int i)
{
    return min(max(max(qp_iter[i], qp_iter_prev[i]), 0), SORT_BINS - 1);
}

static __global__ void __launch_bounds__(256) usv_sort_hist(const int *qp_iter, const int *qp_iter_prev, int B, int *hist)
{
    __shared__ int loc[SORT_BINS];
    if (threadIdx.x < SORT_BINS) loc[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) atomicAdd(&loc[sort_key(qp_iter, qp_iter_prev, i)], 1);
    __syncthreads();
    if (threadIdx.x < SORT_BINS && loc[threadIdx.x] != 0) atomicAdd(&hist[threadIdx.x], loc[threadIdx.x]);
}

static __global__ void usv_sort_scan(int *hist, int *cursor)
{
    if (threadIdx.x == 0) {
        int pos = 0;
        for (int b = SORT_BINS - 1; b >= 0; b--) { // descending difficulty
            cursor[b] = pos;
            pos += hist[b];
            hist[b] = 0;
        }
    }
}

static __global__ void __launch_bounds__(256) usv_sort_scatter(const int *qp_iter, const int *qp_iter_prev, int B, int *cursor, int *perm)
{
    __shared__ int loc[SORT_BINS], base[SORT_BINS];
    if (threadIdx.x < SORT_BINS) loc[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int bin = 0, rank = 0;
    if (i < B) {
        bin = sort_key(qp_iter, qp_iter_prev, i);
        rank = atomicAdd(&loc[bin], 1);
    }
    __syncthreads();
    if (threadIdx.x < SORT_BINS && loc[threadIdx.x] != 0) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], loc[threadIdx.x]);
    __syncthreads();
    if (i < B) perm[base[bin] + rank] = i;
}

// ---------------------------------------------------------------------------------- handle
struct usvmpc_handle {
    usvmpc_desc desc;
    DevSpec spec;
    DevPtrs ptrs;
    int nx, nu, nz, ny, ny_e, N, K, B, Bp, kch;
    bool soft;
    int device;
    hipStream_t stream;
    bool own_stream;
    static constexpr int RING = 64;      // per-solve event triples, newest at (nsolves-1) % RING
    hipEvent_t ev[RING][4];   // start | lineariser done | QP phase done | main QP launch done (the follow-up launch of a hand-over comes after it)
    long nsolves;
    DevSpec *d_spec;
    int *d_perm, *d_hist, *d_cursor, *d_iter_prev;
    GuidancePtrs gd;          // device buffers of the guidance front end (allocated on first use)
    bool gd_ready;
    int gd_npts_cap;
    double *gd_psi;
    double *gd_world;   // [B][n_world][3] world obstacles of the last usvmpc_guidance_sense
    size_t gd_world_cap;
    bool sort_enabled;
    bool sort_two;            // sort key: the larger of the last two iteration counts instead of the last one (option, default off)
    bool merge_rows;
    bool dynamic_rows;        // QP kernel as a persistent launch whose rows pull instances from a queue (option "dynamic_rows")
    int lds_mode;             // workspace of the QP kernel in LDS: -1 when the batch is small enough (default), 0 never, 1 whenever it fits
    long lds_cap;             // waves an LDS-workspace launch holds at once (0: not yet known)
    int wide_mode;            // the latency mapping (one instance per wave, QpIpm WIDE): -1 for small batches (default), 0 never, 1 whenever it applies
    long wide_cap;            // waves a launch of the wide kernel holds at once (0: not yet known, -1: does not fit)
    long wide_hbm_cap;        // the same for the wide kernel over planes in HBM (horizons that do not fit LDS)
    int wide_waves;           // waves per instance of the latency mapping: -1 (default) four for soft-row OCPs and two obstacle chunks while the batch is at most one instance per CU, else one; 1; 4
    long wide4_cap, wide4_hbm_cap; // workgroups of four waves a launch holds at once (0: not yet known, -1: does not fit)
    int last_wide;            // the last RTI launch ran on the wide kernel
    int handover_iter;        // option "handover_iter": IPM iterations after which a row of a drained launch hands its instance to the follow-up launch (0: never)
    int *d_susp_count, *d_susp_list; // [RING] instances each of the last launches handed over / [B] their groups
    double *d_susp_rec;       // [B][4] (DevPtrs::susp_rec)
    long resume_cap;          // workgroups of the follow-up launch (0: not yet known, -1: the kernel cannot be launched)
    // the follow-up kernel beside the draining launch (usv_qp_resume_co; option "handover_co": -1 = when the follow-up works in LDS, 0 never, 1 the same)
    int handover_co;
    int co_spin_limit;        // polls of ~1 us a co-resident workgroup waits for its entry before it gives up (option "handover_co_spin")
    long co_wgs;              // workgroups of the co-resident launch (option "handover_co_wgs"; 0: one per CU - launch_qp says why)
    hipStream_t co_stream;    // nullptr until first used
    hipEvent_t ev_co_pre, ev_co_end;
    int *d_co_ctl;            // [RING][8] DevPtrs::co_ctl of the last launches
    bool ev3_set[RING];       // ev[.][3] was recorded for that solve
    bool resume_lds, handover_lds; // the follow-up launch copies the planes into LDS (chosen with resume_cap) / option "handover_lds"
    long max_waves;           // cap on the persistent waves of the QP kernel (0: as many as the device holds)
    int ncu;                  // compute units of the device
    long qp_cap;              // groups a full-occupancy launch of the QP kernel holds at once (0: not yet known)
    bool aux_lds;             // option "aux_in_lds": the aux plane of an RTI solve in the waves' LDS when the horizon fits (default on)
    long aux_cap;             // the same for the aux-in-LDS instantiation (0: not yet known, -1: does not fit / would cost a wave)
    bool map_changed;         // the group -> instance map differs from the one the workspace's multipliers were written under
    unsigned noise_mask;      // states usvmpc_advance disturbs (option "disturbance_mask"; default: all)
    int *d_fail_ring;         // [RING] instances with status != 0, one slot per solve
    int *d_unconv_ring;       // [RING] instances whose QP did not converge to the tolerances (qp_status != 0), one slot per solve
    unsigned long long *d_unconv_total; // [1] ... summed over every RTI solve of the handle (usvmpc_unconverged_total)
    // Caller-visible arrays live in ONE device arena, in the order [x | u | status | x0 | yref | yref_e | p | lh] (256-byte aligned
    // pieces).  Small handles (the single-instance drop-in faces: AcadosOcpSolver, the acados C shim) also keep a pinned host
    // MIRROR of it: usvmpc_set then writes the mirror and marks the field dirty - no HIP call, no synchronisation - and the next
    // launch uploads what is dirty (the reference protocol's 3N+4 setters per tick become ONE asynchronous copy: the dirty
    // fields are adjacent); after a solve [x | u | status] comes back in ONE copy and usvmpc_get "x" / "u" is served from it.
    enum { F_X = 0, F_U, F_STATUS, F_X0, F_YREF, F_YREF_E, F_P, F_LH, F_COUNT };
    char *arena;              // device
    char *mirror;             // pinned host copy of the arena, or nullptr (arena larger than MIRROR_MAX)
    size_t arena_bytes, f_off[F_COUNT + 1], f_len[F_COUNT];
    size_t dirty_lo[F_COUNT], dirty_hi[F_COUNT]; // byte range inside each field that the mirror holds newer than the device (lo >= hi: none)
    // one stage of several instances set at a time (the reference protocol on a small multi-instance handle): the rows of that stage are not
    // contiguous, so they are remembered per (field, stage) and go up at the next launch as 2-D copies of runs of consecutive stages
    std::vector<char> dirty_stage[F_COUNT];
    bool inflight;            // a copy between mirror and arena may still be running on the stream
    bool out_valid;           // the mirror's [x | u | status] is what the device holds (or newer)
    bool extern_access;       // a device pointer was handed out: the device arrays may change behind the mirror
    // Pipelined lineariser (option "pipeline_linearize"): the lineariser of tick t + 1 is launched on a second stream right behind the
    // QP launch of tick t; its workgroups are dispatched as that launch's persistent waves leave, i.e. it runs in the launch's tail
    // (profiles/r03_tail.txt: the last ~14 ms of a 76 ms launch run on a device that is being vacated).  An instance is linearised
    // there only when its own results AND those of the instance still owning the target planes are final (DevPtrs::epoch); the few
    // that were not are redone by a fix-up pass in front of the next QP launch.  The queue order of a tick is then fixed one tick
    // earlier (from the counts of two solves back).  Scheduling only: results are bit-identical.
    bool pipeline;            // option; used for RTI solves of handles without a host mirror
    hipStream_t aux_stream;   // nullptr until first used
    hipEvent_t ev_pre, ev_spec;
    int *d_epoch, *d_redo, *d_perm2;
    long spec_for;            // solve number the outstanding / finished speculative linearisation was made for (-1: none)
    bool spec_valid;          // ... and nothing it read has been changed by the caller since
    bool spec_outstanding;    // the second stream may still be writing the lineariser's planes
    int spec_quiet;           // RTI solves in a row whose ahead-of-time linearisation (had there been one) no caller write invalidated: the
                              // lineariser runs ahead only from SPEC_QUIET_MIN on - a caller that sets yref / x / u every tick (the reference's
                              // protocol: scripts/usv_guidance_ca1/main.py:123-130) never pays for a speculative pass that is thrown away
    long spec_hits, spec_misses; // ahead-of-time linearisations used / discarded (usvmpc_pipeline_stats)
    const int *spec_perm;     // the group -> instance map it used (= the map the solve spec_for must use)
    // partial condensing (option "qp_cond_N"): RTI solves condense the QP to cond_N2 stages first (0: off - the Riccati sweep over the N stages)
    int cond_N2;
    CondDims cond_dims;       // sizes of the condensed QP (valid when d_cond_dims is set)
    CondDims *d_cond_dims;
    double *d_cond_scratch;   // [cond_teams][cond_dims.total]
    long cond_teams;
    size_t cond_lds;          // dynamic LDS of the condensing kernel, bytes
    long export_at;           // nsolves the multiplier read-back buffers (ptrs.lam_out / t_out) were filled at; -1: never
    bool last_cond;           // the last launch solved the partially condensed QP (its rows live in per-team scratch, not in the workspace)
    bool layout_dirty;        // the row layout option changed after the last solve: the workspace cannot be read back
    size_t bytes;
    std::string err;
    std::vector<void *> allocs;
};

namespace {

thread_local std::string g_create_err;

#define HIP_TRY(h, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return USVMPC_E_HIP;                                                                  \
        }                                                                                         \
    } while (0)

template <class T>
int dev_alloc(usvmpc_handle *h, T **p, size_t count, bool zero)
{
    const size_t nbytes = (count ? count : 1) * sizeof(T);
    HIP_TRY(h, hipMalloc((void **)p, nbytes));
    h->allocs.push_back((void *)*p);
    h->bytes += nbytes;
    // on the handle's stream: it is non-blocking, i.e. NOT ordered against the legacy default stream hipMemset uses
    if (zero) HIP_TRY(h, hipMemsetAsync((void *)*p, 0, nbytes, h->stream));
    return 0;
}

// release one buffer obtained from dev_alloc (growing a guidance buffer replaces it)
void dev_free(usvmpc_handle *h, void *p, size_t nbytes)
{
    if (!p) return;
    for (size_t i = 0; i < h->allocs.size(); i++)
        if (h->allocs[i] == p) { h->allocs.erase(h->allocs.begin() + (long)i); break; }
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(p);
    h->bytes -= nbytes;
}

// Forget what the occupancy queries said about the QP kernels: whatever changes WHICH kernel a launch takes (row layout, mapping,
// wave cap, workspace placement) makes launch_qp ask again - for every instantiation, with its dynamic LDS size set afresh.
void reset_caps(usvmpc_handle *h)
{
    h->qp_cap = 0; h->lds_cap = 0; h->aux_cap = 0;
    h->wide_cap = 0; h->wide_hbm_cap = 0; h->wide4_cap = 0; h->wide4_hbm_cap = 0; h->resume_cap = 0;
}

#if USV_MAIN
struct Field {
    double *base;   // device pointer
    int n;          // per-stage length
    int stages;     // number of stages in the array
    int stage_off;  // caller stage that maps to slot 0
};

// caller-visible field table; `set` = writable by the caller
int lookup(usvmpc_handle *h, const char *f, int stage, bool set, Field &o)
{
    const std::string s(f ? f : "");
    DevPtrs &P = h->ptrs;
    const int N = h->N;
    if (s == "x") o = {P.x, h->nx, N + 1, 0};
    else if (s == "u") o = {P.u, h->nu, N, 0};
    else if (s == "x0") o = {const_cast<double *>(P.x0), h->nx, 1, 0};
    else if (s == "yref") {
        if (stage == N) o = {const_cast<double *>(P.yref_e), h->ny_e, 1, N};
        else o = {const_cast<double *>(P.yref), h->ny, N, 0};
    } else if (s == "yref_e") o = {const_cast<double *>(P.yref_e), h->ny_e, 1, 0};
    else if (s == "p") o = {const_cast<double *>(P.p), 2 * h->K, N + 1, 0};
    else if (s == "lh") o = {const_cast<double *>(P.lh), h->K, N, 0};
    else if (!set && s == "pi") o = {P.pi, h->nx, N, 1};
    else if (!set && s == "sl") o = {P.sl, h->K, N, 0};
    else if (!set && s == "su") o = {P.su, h->K, N, 0};
    else if (!set && s == "res") o = {P.res, 4, 1, 0};
    else if (!set && s == "obs_tmin") o = {P.obs_tmin, 1, 1, 0};
    else if (!set && s == "nlp_res") o = {P.nlp_res, 4, 1, 0};
    else if (!set && s == "lam" && P.lam_out) o = {P.lam_out, P.nlam, N + 1, 0};
    else if (!set && s == "t" && P.t_out) o = {P.t_out, P.nlam, N + 1, 0};
    else {
        h->err = "unknown field '" + s + "'";
        return USVMPC_E_FIELD;
    }
    return 0;
}

int ensure_export(usvmpc_handle *h); // (below: needs the kernel dispatch)

// the speculative lineariser of the second stream: wait for it and forget what it made (something it read or wrote is about to change)
int spec_cancel(usvmpc_handle *h)
{
    h->spec_valid = false;
    h->spec_quiet = 0;
    if (h->spec_outstanding) {
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipStreamSynchronize(h->aux_stream));
        h->spec_outstanding = false;
    }
    return 0;
}

constexpr size_t MIRROR_MAX = 1u << 20; // arenas up to 1 MiB are mirrored on the host

int field_index(const std::string &s, int stage, int N)
{
    if (s == "x") return usvmpc_handle::F_X;
    if (s == "u") return usvmpc_handle::F_U;
    if (s == "x0") return usvmpc_handle::F_X0;
    if (s == "yref") return stage == N ? usvmpc_handle::F_YREF_E : usvmpc_handle::F_YREF;
    if (s == "yref_e") return usvmpc_handle::F_YREF_E;
    if (s == "p") return usvmpc_handle::F_P;
    if (s == "lh") return usvmpc_handle::F_LH;
    return -1;
}

// wait for a mirror <-> arena copy that may still be running before the host touches the mirror
int mirror_quiesce(usvmpc_handle *h)
{
    if (h->inflight) {
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->inflight = false;
    }
    return 0;
}

#endif // USV_MAIN

// upload what the mirror holds newer than the device: consecutive dirty fields go as one copy
int mirror_flush(usvmpc_handle *h)
{
    if (!h->mirror) return 0;
    for (int fi = 0; fi < usvmpc_handle::F_COUNT; fi++) { // stage-wise sets of a multi-instance handle: runs of consecutive dirty stages
        std::vector<char> &ds = h->dirty_stage[fi];
        if (ds.empty()) continue;
        const size_t stages = ds.size(), pitch = h->f_len[fi] / (size_t)h->B, row = pitch / stages;
        for (size_t a = 0; a < stages;) {
            if (!ds[a]) { a++; continue; }
            size_t b = a;
            while (b + 1 < stages && ds[b + 1]) b++;
            // (a whole-field range that is dirty as well is uploaded below and covers these rows: same bytes, the order does not matter)
            HIP_TRY(h, hipMemcpy2DAsync(h->arena + h->f_off[fi] + a * row, pitch, h->mirror + h->f_off[fi] + a * row, pitch, (b - a + 1) * row, (size_t)h->B,
                                        hipMemcpyHostToDevice, h->stream));
            h->inflight = true;
            for (size_t i = a; i <= b; i++) ds[i] = 0;
            a = b + 1;
        }
        ds.clear();
    }
    int f = 0;
    while (f < usvmpc_handle::F_COUNT) {
        if (h->dirty_lo[f] >= h->dirty_hi[f]) { f++; continue; }
        size_t lo = h->f_off[f] + h->dirty_lo[f], hi = h->f_off[f] + h->dirty_hi[f];
        // extend over the following fields while this one is dirty to its end and the next from its start
        int g = f;
        while (g + 1 < usvmpc_handle::F_COUNT && h->dirty_hi[g] == h->f_len[g] && h->dirty_lo[g + 1] == 0 && h->dirty_hi[g + 1] > 0) {
            g++;
            hi = h->f_off[g] + h->dirty_hi[g];
        }
        HIP_TRY(h, hipMemcpyAsync(h->arena + lo, h->mirror + lo, hi - lo, hipMemcpyHostToDevice, h->stream));
        h->inflight = true;
        for (int i = f; i <= g; i++) { h->dirty_lo[i] = 1; h->dirty_hi[i] = 0; }
        f = g + 1;
    }
    return 0;
}

#if USV_MAIN
int copy_field(usvmpc_handle *h, const char *field, int stage, double *host, size_t n, bool set)
{
    if (!h) return USVMPC_E_ARG;
    if (!host) { h->err = "null buffer"; return USVMPC_E_ARG; }
    if (!set && (std::string(field ? field : "") == "lam" || std::string(field ? field : "") == "t")) {
        const int rce = ensure_export(h);
        if (rce) return rce;
    }
    Field f;
    if (std::string(field ? field : "") == "res" || std::string(field ? field : "") == "nlp_res" ||
        std::string(field ? field : "") == "obs_tmin" ||
        std::string(field ? field : "") == "x0" ||
        std::string(field ? field : "") == "yref_e")
        stage = stage < 0 ? -1 : 0;
    int rc = lookup(h, field, stage, set, f);
    if (rc) return rc;
    if ((int)n != f.n) {
        h->err = std::string("mismatching dimension for field '") + field + "': expected " + std::to_string(f.n) +
                 ", got " + std::to_string(n);
        return USVMPC_E_SIZE;
    }
    if (f.n == 0) return 0;
    const size_t B = (size_t)h->B;
    const int fi = h->mirror ? field_index(std::string(field ? field : ""), stage, h->N) : -1;
    if (fi >= 0 && (set || ((fi == usvmpc_handle::F_X || fi == usvmpc_handle::F_U) && h->out_valid && !h->extern_access))) {
        // ---- through the pinned mirror: no HIP call unless a copy is still in flight
        const bool whole = stage < 0 || f.stages == 1;
        if (!whole && (stage - f.stage_off < 0 || stage - f.stage_off >= f.stages)) {
            h->err = std::string("stage ") + std::to_string(stage) + " out of range for field '" + field + "'";
            return USVMPC_E_STAGE;
        }
        if (whole && stage >= 0 && stage - f.stage_off != 0) { h->err = "stage out of range"; return USVMPC_E_STAGE; }
        rc = mirror_quiesce(h);
        if (rc) return rc;
        char *m = h->mirror + h->f_off[fi];
        const size_t row = (size_t)f.n * sizeof(double), pitch = (size_t)f.stages * row;
        const size_t first = whole ? 0 : (size_t)(stage - f.stage_off) * row;
        if (whole) {
            if (set) std::memcpy(m, host, B * pitch); else std::memcpy(host, m, B * pitch);
        } else {
            for (size_t b = 0; b < B; b++) {
                if (set) std::memcpy(m + b * pitch + first, (const char *)host + b * row, row);
                else std::memcpy((char *)host + b * row, m + b * pitch + first, row);
            }
        }
        if (set && !whole && B > 1) {
            // one stage of several instances: the rows are not contiguous, and a dirty RANGE over them would later upload the mirror's
            // copy of everything in between - stale where a device-side writer (the guidance kernels, a caller holding a device pointer)
            // has written behind the mirror.  The stage is remembered; its rows go up with the next launch (mirror_flush), as one 2-D copy
            // per run of consecutive dirty stages - no HIP call here (the immediate copy made the NEXT set synchronise)
            // A caller that holds device pointers (usvmpc_get_device_ptr: extern_access) orders its own kernels against this set by the
            // stream: for it the rows go up NOW, as before (a deferred upload would land on top of what the caller's kernel
            // wrote to the same stage after the set, and a caller kernel reading the field before the solve would see the old rows).
            if (h->extern_access) {
                HIP_TRY(h, hipSetDevice(h->device));
                HIP_TRY(h, hipMemcpy2DAsync(h->arena + h->f_off[fi] + first, pitch, m + first, pitch, row, B, hipMemcpyHostToDevice, h->stream));
                h->inflight = true;
            } else {
                std::vector<char> &ds = h->dirty_stage[fi];
                if (ds.size() != (size_t)f.stages) ds.assign((size_t)f.stages, 0);
                ds[(size_t)(stage - f.stage_off)] = 1;
            }
        } else if (set) {
            const size_t lo = first, hi = whole ? B * pitch : first + row;
            if (h->dirty_lo[fi] >= h->dirty_hi[fi]) { h->dirty_lo[fi] = lo; h->dirty_hi[fi] = hi; }
            else { h->dirty_lo[fi] = std::min(h->dirty_lo[fi], lo); h->dirty_hi[fi] = std::max(h->dirty_hi[fi], hi); }
        }
        if (set && (fi == usvmpc_handle::F_X || fi == usvmpc_handle::F_U || fi == usvmpc_handle::F_YREF || fi == usvmpc_handle::F_YREF_E))
            h->spec_quiet = 0;
        return 0;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    if (set) { // (a lineariser that ran ahead may have read what is being replaced - it reads x, u, yref: the next solve linearises again)
        const std::string fs(field ? field : "");
        if (fs == "x" || fs == "u" || fs == "yref" || fs == "yref_e") { h->spec_valid = false; h->spec_quiet = 0; }
    }
    if (!set) { rc = mirror_flush(h); if (rc) return rc; } // (a get of a field with pending writes sees them)
    if (stage < 0 || f.stages == 1) {
        if (stage >= 0 && f.stages == 1 && stage - f.stage_off != 0) {
            h->err = "stage out of range"; return USVMPC_E_STAGE;
        }
        const size_t nbytes = B * f.stages * f.n * sizeof(double);
        if (set) HIP_TRY(h, hipMemcpyAsync(f.base, host, nbytes, hipMemcpyHostToDevice, h->stream));
        else HIP_TRY(h, hipMemcpyAsync(host, f.base, nbytes, hipMemcpyDeviceToHost, h->stream));
    } else {
        const int slot = stage - f.stage_off;
        if (slot < 0 || slot >= f.stages) {
            h->err = std::string("stage ") + std::to_string(stage) + " out of range for field '" + field + "'";
            return USVMPC_E_STAGE;
        }
        const size_t row = (size_t)f.n * sizeof(double), pitch = (size_t)f.stages * row;
        double *d = f.base + (size_t)slot * f.n;
        if (set) HIP_TRY(h, hipMemcpy2DAsync(d, pitch, host, row, row, B, hipMemcpyHostToDevice, h->stream));
        else HIP_TRY(h, hipMemcpy2DAsync(host, row, d, pitch, row, B, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->inflight = false;
    return 0;
}

#endif // USV_MAIN

// the condensed QP solve of an RTI iteration (after the lineariser): buffers on first use; the kernels live in cond_kernels.hip
int launch_cond(usvmpc_handle *h)
{
    if (!h->d_cond_dims) {
        CondDims D;
        size_t lds = 0;
        int nb = 0;
        const int rcp = cond_prepare(h->desc.model, h->kch, h->spec, h->cond_N2, D, lds, nb, h->err);
        if (rcp) return rcp;
        if (nb < 1 || h->ncu < 1) { h->err = "partial condensing: the kernel cannot be launched with this much LDS"; return USVMPC_E_HIP; }
        long teams = std::min<long>(h->B, (long)nb * h->ncu);
        if (h->max_waves > 0) teams = std::min<long>(teams, h->max_waves);
        if (dev_alloc(h, &h->d_cond_dims, 1, false)) return USVMPC_E_HIP;
        if (dev_alloc(h, &h->d_cond_scratch, (size_t)teams * (size_t)D.total, true)) return USVMPC_E_HIP;
        HIP_TRY(h, hipMemcpyAsync(h->d_cond_dims, &D, sizeof(CondDims), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream)); // (D is a local)
        h->cond_dims = D; h->cond_teams = teams; h->cond_lds = lds;
    }
    HIP_TRY(h, hipMemsetAsync(h->ptrs.queue, 0, sizeof(int), h->stream));
    if (h->ptrs.lam_out) { // multiplier read-back: this solve fills the buffers (rows a stage does not have read 0)
        const size_t nbytes = (size_t)h->B * (h->N + 1) * (size_t)(h->ptrs.nlam > 0 ? h->ptrs.nlam : 1) * sizeof(double);
        HIP_TRY(h, hipMemsetAsync(h->ptrs.lam_out, 0, nbytes, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->ptrs.t_out, 0, nbytes, h->stream));
        h->export_at = h->nsolves + 1;
    }
    if (cond_run(h->desc.model, h->kch, h->cond_dims, h->stream, h->cond_teams, h->cond_lds, h->ptrs, h->d_cond_dims, h->d_cond_scratch, h->B)) {
        h->err = "partial condensing: no kernel for this model in this library";
        return USVMPC_E_ARG;
    }
    h->map_changed = true; // (the group-indexed workspace holds no multipliers of this QP: a later full SQP starts afresh)
    return 0;
}

void cond_release(usvmpc_handle *h)
{
    if (h->d_cond_scratch) dev_free(h, h->d_cond_scratch, (size_t)h->cond_teams * (size_t)h->cond_dims.total * sizeof(double));
    if (h->d_cond_dims) dev_free(h, h->d_cond_dims, sizeof(CondDims));
    h->d_cond_scratch = nullptr; h->d_cond_dims = nullptr; h->cond_teams = 0;
}

} // namespace

// (external linkage: a split build defines each instantiation in a translation unit of its own - see "Build parts" at the top)
template <class M, int KCH, bool SOFT>
int launch_pair(usvmpc_handle *h, int phase)
{
    {   // pending host writes go up first; the results come back below
        const int rcf = mirror_flush(h);
        if (rcf) return rcf;
    }
    const long lin_groups = (long)(h->N + 1) * h->Bp;
    const long qp_groups = h->Bp;
    const int lin_block = 256, qp_block = 64;
    const long lin_grid = (lin_groups * LANES + lin_block - 1) / lin_block;
    hipEvent_t *ev = h->ev[h->nsolves % usvmpc_handle::RING];
    const int B = h->B;
    // counting sort of the previous solve's iteration counts into a group -> instance map
    auto sort_into = [&](int *dst) {
        const int *prev2 = h->sort_two ? h->d_iter_prev : h->ptrs.qp_iter; // (option "sort_two_ticks" = 0: the last count alone)
        hipLaunchKernelGGL(usv_sort_hist, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->ptrs.qp_iter, prev2, B, h->d_hist);
        hipLaunchKernelGGL(usv_sort_scan, dim3(1), dim3(64), 0, h->stream, h->d_hist, h->d_cursor);
        hipLaunchKernelGGL(usv_sort_scatter, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->ptrs.qp_iter, prev2, B, h->d_cursor, dst);
    };
    auto lin_launch = [&](auto mode, hipStream_t st, const DevPtrs &P) {
        constexpr int MODE = decltype(mode)::value;
        if (h->spec.sim_steps > 1)
            hipLaunchKernelGGL((usv_linearize<M, KCH, SOFT, true, MODE>), dim3((unsigned)lin_grid), dim3(lin_block), 0, st, P, lin_groups);
        else
            hipLaunchKernelGGL((usv_linearize<M, KCH, SOFT, false, MODE>), dim3((unsigned)lin_grid), dim3(lin_block), 0, st, P, lin_groups);
    };
    // Pipelined lineariser (see usvmpc_handle): RTI solves of large handles
    const bool cond = phase == 0 && h->cond_N2 > 0; // RTI solve on the partially condensed QP (cond_ipm.hpp)
    const bool pipe = phase == 0 && h->pipeline && !h->mirror && !h->extern_access && h->dynamic_rows && h->B >= 16384 && !cond;
    if (pipe && !h->aux_stream) {
        // (lowest priority: when this stream's lineariser and the main stream's QP launch become eligible together, the QP
        // launch's workgroups are placed first and the lineariser gets the compute units that launch vacates)
        int prio_least = 0, prio_greatest = 0;
        HIP_TRY(h, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIP_TRY(h, hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, prio_least));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_pre, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_spec, hipEventDisableTiming));
        if (dev_alloc(h, &h->d_epoch, (size_t)B, false) || dev_alloc(h, &h->d_redo, (size_t)B * ((h->N + 32) / 32), true) ||
            dev_alloc(h, &h->d_perm2, (size_t)B, true))
            return USVMPC_E_HIP;
        HIP_TRY(h, hipMemsetAsync(h->d_epoch, 0xff, (size_t)B * sizeof(int), h->stream)); // -1: nothing is final yet
    }
    // whatever this solve does with the lineariser's planes comes after a speculative linearisation that may still be writing them
    if (h->spec_outstanding) {
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_spec, 0));
        h->spec_outstanding = false;
    }
    const bool had_spec = pipe && h->spec_for == h->nsolves;
    const bool use_spec = had_spec && h->spec_valid;
    if (had_spec) (use_spec ? h->spec_hits : h->spec_misses)++;
    h->spec_valid = false;
    if (use_spec) {
        // the map this tick was linearised under (made one tick ago from the counts of the solve before)
        h->ptrs.perm = h->spec_perm;
        h->map_changed = true;
    } else if (h->sort_enabled && h->nsolves > 0 && phase != 2) {
        // (the later iterations of a full SQP read the multipliers the previous launch left in the group-indexed workspace: the
        // group -> instance map must not change inside one SQP call; a phase-1 launch re-sorts BEFORE its QP writes the multipliers:
        // map and workspace stay consistent)
        if (phase == 0) h->map_changed = true;
        sort_into(h->d_perm);
        HIP_TRY(h, hipGetLastError());
        h->ptrs.perm = h->d_perm;
    }
    constexpr int SPEC_QUIET_MIN = 2;
    if (pipe) h->spec_quiet++; // (reset by every caller write that would invalidate a linearisation made ahead of time)
    const bool spec_next = pipe && h->spec_quiet >= SPEC_QUIET_MIN; // the next tick's lineariser runs ahead, beside this solve's QP launch
    h->ptrs.epoch = spec_next ? h->d_epoch : nullptr;
    h->ptrs.redo = pipe ? h->d_redo : nullptr;
    h->ptrs.redo_words = (h->N + 32) / 32;
    h->ptrs.perm_cur = nullptr;
    h->ptrs.tick = (int)h->nsolves;
    h->ev3_set[h->nsolves % usvmpc_handle::RING] = false;
    HIP_TRY(h, hipEventRecord(ev[0], h->stream));
    if (use_spec) lin_launch(std::integral_constant<int, 2>{}, h->stream, h->ptrs); // only what the speculative pass had to skip
    else lin_launch(std::integral_constant<int, 0>{}, h->stream, h->ptrs);
    HIP_TRY(h, hipGetLastError());
    if (pipe) HIP_TRY(h, hipMemsetAsync(h->d_redo, 0, (size_t)B * ((h->N + 32) / 32) * sizeof(int), h->stream));
    HIP_TRY(h, hipEventRecord(ev[1], h->stream));
    h->ptrs.fail_count = h->d_fail_ring + h->nsolves % usvmpc_handle::RING;
    HIP_TRY(h, hipMemsetAsync(h->ptrs.fail_count, 0, sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_unconv_ring + h->nsolves % usvmpc_handle::RING, 0, sizeof(int), h->stream));
    if (h->d_susp_count) HIP_TRY(h, hipMemsetAsync(h->d_susp_count + h->nsolves % usvmpc_handle::RING, 0, sizeof(int), h->stream)); // (hand-over count of this launch)
    const int *next_perm = nullptr;
    if (spec_next) {
        // the NEXT tick's map, from the counts this launch is about to overwrite, into the buffer this tick does not use
        if (h->sort_enabled) {
            int *dst = (h->ptrs.perm == h->d_perm) ? h->d_perm2 : h->d_perm;
            sort_into(dst);
            HIP_TRY(h, hipGetLastError());
            next_perm = dst;
        }
        HIP_TRY(h, hipEventRecord(h->ev_pre, h->stream));
    }
    // (the counts of the solve before this one: the second half of the next sort key - copied after every sort of THIS solve has read
    // the pair (qp_iter, d_iter_prev), the pipelined map's included, and before the QP launch overwrites qp_iter)
    if (h->sort_enabled && h->sort_two && phase == 0)
        HIP_TRY(h, hipMemcpyAsync(h->d_iter_prev, h->ptrs.qp_iter, (size_t)h->B * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    constexpr bool CANPACK = KCH > 0;
    const bool pack = CANPACK && h->spec.boxpack != 0;
    if (h->spec.npt != (h->spec.any_bsoft ? WsLayout<M, KCH, SOFT, true>::NPT : WsLayout<M, KCH, SOFT, false>::NPT)) {
        h->err = "workspace layout mismatch between host and kernels";
        return USVMPC_E_ARG;
    }
    // An RTI solve is ONE launch of as many waves as the device holds at once; their rows start on the first groups and
    // pull the remaining ones from a queue as they finish (qp_ipm.hpp).  The full SQP keeps one group per row: its later
    // iterations find their multipliers in the group's part of the workspace.
    // Small batches: the planes of every instance in flight fit in LDS (160 KB per CU), and a solve whose sweeps wait for
    // HBM at every stage - nothing else runs on the CU to hide it - becomes a solve on LDS.  rows_lds instances per wave
    // (as many whole horizons as fit), one wave per CU at a time; further instances come through the same queue.
    auto launch_qp = [&](auto kern, decltype(kern) kern_lds, decltype(kern) kern_aux = nullptr, WideSet wide = NO_WIDE) -> int {
        const long lds_inst = (long)(h->N + 1) * h->spec.npt * 128;
        h->last_wide = 0;
        h->ptrs.susp_count = nullptr; h->ptrs.susp_list = nullptr; h->ptrs.susp_rec = nullptr; h->ptrs.handover_iter = 0; h->ptrs.co_ctl = nullptr; // (set by the path that hands over)
        const qp_kernel_t kern_wide = wide.lds1, kern_wide_hbm = wide.hbm1;
        // Four waves per instance (qp_ipm.hpp, WW): a workgroup = a whole CU shares out the row work of 16 consecutive stages - for the
        // single instance and batches of at most one instance per CU.
        if (wide.lds4 != nullptr && phase == 0 && h->wide_mode != 0 && h->wide_waves != 1 && h->ncu > 0) {
            const size_t pl = (size_t)(h->N + 1) * (size_t)wide.nplw * 128;
            const size_t b4 = pl + (size_t)16 * wide.ex_lds * 128 + 128, x4 = (size_t)16 * wide.ex_hbm * 128 + 128;
            const long win_bytes = (long)std::min(h->N + 1, 16) * h->Bp * h->spec.npt * 128; // (the window of a block of 16 stages: 32-bit offsets)
            if (h->wide4_cap == 0) {
                int nb = 0;
                hipFuncAttributes fa;
                if (b4 <= 160u * 1024u && hipFuncGetAttributes(&fa, (const void *)wide.lds4) == hipSuccess && fa.sharedSizeBytes + b4 <= 160u * 1024u &&
                    hipFuncSetAttribute((const void *)wide.lds4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b4) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wide.lds4, 4 * qp_block, b4) == hipSuccess && nb > 0)
                    h->wide4_cap = (long)h->ncu;
                else
                    h->wide4_cap = -1;
            }
            if (h->wide4_cap < 0 && h->wide4_hbm_cap == 0) {
                int nb = 0;
                h->wide4_hbm_cap = (wide.hbm4 != nullptr && win_bytes < (1L << 32) &&
                                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wide.hbm4, 4 * qp_block, x4) == hipSuccess && nb > 0) ? (long)h->ncu : -1;
            }
            const bool lds = h->wide4_cap > 0;
            long cap = lds ? h->wide4_cap : h->wide4_hbm_cap;
            if (cap > 0 && h->max_waves > 0) cap = std::max<long>(1, std::min(cap, h->max_waves / 4)); // option "max_waves" counts wavefronts
            // default: where the row work is the larger share - the soft-row OCPs and two obstacle chunks (measured, one instance / 256 instances
            // per tick: usv_model_guidance_ca1 N = 100 / K = 8 1.78 -> 1.59 / 5.6 -> 5.0 ms, N = 40 / K = 10 0.94 -> 0.86 / 2.05 -> 1.87, N = 80 / K = 20
            // 4.00 -> 3.00 / 8.1 -> 6.2; usv_model_pf_ca N = 80 / K = 20 6.95 -> 6.22 / 10.4 -> 9.6, with ONE chunk of hard rows 0 - 7 % SLOWER: there
            // the recursion dominates and pays the barriers)
            // Up to one instance per CU; with the queue and a horizon of 40 or more up to two (tools/latency_probe.py over 13 shapes x 7 batch sizes,
            // profiles/r05_f_policy_audit.txt: 512 instances 6 - 8 % under one wave each; at N = 20 the second round costs more than the row work saves)
            const long reach = (h->dynamic_rows && h->N >= 40) ? 2 * cap : cap;
            // (round 6, profiles/r06_b_policy_audit.txt: ONE chunk of hard rows also gains 2 - 4 % from four waves when the rows are many and the
            // horizon long - usv_model_pf_ca N = 40 / K = 10: one instance 2.50 -> 2.40 ms, 64: 5.84 -> 5.63, 256: 4.03 -> 3.94; N = 100 / K = 8,
            // 64: 9.62 -> 9.34; with K = 3 or 4 it loses - up to one instance per CU)
            const bool hard_many = !SOFT && KCH == 1 && h->K >= 8 && h->N >= 40 && (long)h->B <= cap;
            if (cap > 0 && (h->wide_waves == 4 || ((SOFT || KCH == 2) && (long)h->B <= reach) || hard_many)) {
                long nw = (long)h->B;
                int q0 = -1;
                if (h->dynamic_rows && nw > cap) { nw = cap; q0 = (int)nw; }
                if (q0 >= 0) HIP_TRY(h, hipMemsetAsync(h->ptrs.queue, 0, sizeof(int), h->stream));
                hipLaunchKernelGGL(lds ? wide.lds4 : wide.hbm4, dim3((unsigned)nw), dim3(4 * qp_block), lds ? b4 : x4, h->stream, h->ptrs, nw, phase, q0, 1);
                h->last_wide = 4;
                return 0;
            }
        }
        // The latency mapping: ONE instance per wave (qp_ipm.hpp, WIDE) - planes in LDS, the four rows share out the stage-local row
        // work.  A wave then finishes an instance 1.4x (hard rows) to 1.8x (soft rows) sooner and the device holds a quarter of the instances at once: it pays while
        // the batch leaves SIMDs idle anyway (a solve of the batch then lasts as long as its hardest instance on a lone wave).
        if (kern_wide != nullptr && h->wide_mode != 0 && h->ncu > 0) {
          if (phase == 0) { // (the launches of a full SQP find their multipliers in the group's planes in HBM: the variant over planes in HBM below)
            // (in LDS: the planes the solve writes - WsLayout's up to L_zu less the four box planes the packed layouts leave unused)
            const size_t bytes = (size_t)(h->N + 1) * (size_t)wide.nplw * 128 + (size_t)4 * wide.ex_lds * 128;
            if (h->wide_cap == 0) {
                int nb = 0;
                hipFuncAttributes fa;
                if (bytes <= 160u * 1024u && hipFuncGetAttributes(&fa, (const void *)kern_wide) == hipSuccess && fa.sharedSizeBytes + bytes <= 160u * 1024u &&
                    hipFuncSetAttribute((const void *)kern_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern_wide, qp_block, bytes) == hipSuccess && nb > 0)
                    h->wide_cap = (long)std::min(nb, 4) * h->ncu;   // (one wave per SIMD at most: the point is a lone wave's issue rate)
                else
                    h->wide_cap = -1;
                if (h->wide_cap > 0 && h->max_waves > 0) h->wide_cap = std::min(h->wide_cap, h->max_waves); // option "max_waves"
            }
            // default: while the batch fits the SIMDs twice over (the queue hands the second half to the waves that finish first)
            const bool take = h->wide_cap > 0 && (h->wide_mode > 0 || (long)h->B <= 2 * h->wide_cap);
            if (take) {
                long nw = (long)h->B;
                int q0 = -1;
                if (h->dynamic_rows && nw > h->wide_cap) { nw = h->wide_cap; q0 = (int)nw; }
                if (!h->dynamic_rows && nw > h->wide_cap) { /* without the queue every instance needs its wave at launch: still correct, later workgroups wait */ }
                if (q0 >= 0) HIP_TRY(h, hipMemsetAsync(h->ptrs.queue, 0, sizeof(int), h->stream));
                hipLaunchKernelGGL(kern_wide, dim3((unsigned)nw), dim3(qp_block), bytes, h->stream, h->ptrs, nw, phase, q0, 1);
                h->last_wide = 1;
                return 0;
            }
          }
            // The horizon's planes do not fit a CU's LDS (the reference node's own N = 100: nmpc_guidance_ca1.cpp:64), or the launch belongs to a
            // full SQP: the same sweeps over the planes in HBM / L2 - the four rows of a wave address the four stages of a block through one
            // window, the next block's row planes and the next stage's recursion planes are in flight ahead of their use.
            const long win_bytes = (long)std::min(h->N + 1, 4) * h->Bp * h->spec.npt * 128; // (the window of a block of four stages: 32-bit offsets)
            if ((h->wide_cap < 0 || phase != 0) && kern_wide_hbm != nullptr && win_bytes < (1L << 32)) {
                const size_t xbytes = (size_t)4 * wide.ex_hbm * 128;
                if (h->wide_hbm_cap == 0) {
                    int nb = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern_wide_hbm, qp_block, xbytes) == hipSuccess && nb > 0)
                        h->wide_hbm_cap = (long)std::min(nb, 4) * h->ncu;
                    else
                        h->wide_hbm_cap = -1;
                    if (h->wide_hbm_cap > 0 && h->max_waves > 0) h->wide_hbm_cap = std::min(h->wide_hbm_cap, h->max_waves);
                }
                // default: an RTI solve while the batch fits the resident waves twice over, as with the planes in LDS (measured at 2 048 instances, N = 80 / 100:
                // 1.2 - 1.3x the throughput mapping; at 4 096 the throughput mapping is ahead); the launches of a full SQP once (no queue there)
                const long reach = (phase == 0 && h->dynamic_rows) ? 2 * h->wide_hbm_cap : h->wide_hbm_cap;
                if (h->wide_hbm_cap > 0 && (h->wide_mode > 0 || (long)h->B <= reach)) {
                    long nw = (long)h->B;
                    int q0 = -1;
                    // (full SQP: one group per workgroup for the whole call - its multipliers persist in the group's planes)
                    if (h->dynamic_rows && phase == 0 && nw > h->wide_hbm_cap) { nw = h->wide_hbm_cap; q0 = (int)nw; }
                    if (q0 >= 0) HIP_TRY(h, hipMemsetAsync(h->ptrs.queue, 0, sizeof(int), h->stream));
                    hipLaunchKernelGGL(kern_wide_hbm, dim3((unsigned)nw), dim3(qp_block), xbytes, h->stream, h->ptrs, nw, phase, q0, 1);
                    h->last_wide = 1;
                    return 0;
                }
            }
        }
        // (the kernel's own static LDS - exchange area, parked constants - comes out of the same 160 KB)
        long lds_static = 0;
        if (kern_lds != nullptr) {
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, (const void *)kern_lds) == hipSuccess) lds_static = (long)fa.sharedSizeBytes;
        }
        const int rows_lds = (int)std::min<long>(4, (160L * 1024 - lds_static) / lds_inst);
        bool use_lds = phase == 0 && h->lds_mode != 0 && kern_lds != nullptr && rows_lds >= 1 && h->ncu > 0;
        // by default only while one round of workgroups covers the batch: measured on usv_model_pf_ca, N = 20 / K = 3, the solve
        // of 512 instances takes 5.9 ms with the planes in LDS against 6.5 ms in HBM, at 1024 (two rounds) 7.5 against 7.1
        if (use_lds && h->lds_mode < 0) use_lds = (long)h->B <= (long)rows_lds * h->ncu;
        if (use_lds) {
            const size_t bytes = (size_t)rows_lds * lds_inst;
            if (h->lds_cap == 0) {
                int nb = 0;
                if (hipFuncSetAttribute((const void *)kern_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern_lds, qp_block, bytes) == hipSuccess && nb > 0)
                    h->lds_cap = (long)nb * h->ncu;
                else
                    h->lds_cap = -1;
            }
            if (h->lds_cap > 0) {
                long nw = ((long)h->B + rows_lds - 1) / rows_lds;
                int q0 = -1;
                if (h->dynamic_rows && nw > h->lds_cap) { nw = h->lds_cap; q0 = (int)(nw * rows_lds); }
                if (q0 >= 0) HIP_TRY(h, hipMemsetAsync(h->ptrs.queue, 0, sizeof(int), h->stream));
                hipLaunchKernelGGL(kern_lds, dim3((unsigned)nw), dim3(qp_block), bytes, h->stream, h->ptrs, nw * rows_lds, phase, q0, rows_lds);
                return 0;
            }
        }
        long ng = qp_groups;
        int q0 = -1;
        if (h->dynamic_rows && phase == 0 && h->qp_cap == 0) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, qp_block, 0) == hipSuccess && nb > 0 && h->ncu > 0)
                h->qp_cap = 4L * nb * h->ncu;
            else
                h->qp_cap = -1;
        }
        // The aux plane in LDS (qp_ipm.hpp, AUXLDS): 4 rows x (N + 1) stages x at most ten values beside the kernel's static LDS - taken
        // when it does not cost a resident wave (usv_model_pf_ca at N = 40, K = 10: 13.1 KB + 6.7 KB of the 20 KB a wave may have)
        size_t aux_bytes = 0;
        if (phase == 0 && kern_aux != nullptr && h->aux_lds && h->dynamic_rows && h->qp_cap > 0) {
            aux_bytes = (size_t)4 * (h->N + 1) * (size_t)(h->spec.aux_dense4 + (h->kch > 0 ? 2 : 0) + 2 * h->nu) * sizeof(double);
            if (h->aux_cap == 0) {
                int nb = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern_aux, qp_block, aux_bytes) == hipSuccess && 4L * nb * h->ncu >= h->qp_cap)
                    h->aux_cap = 4L * nb * h->ncu;
                else
                    h->aux_cap = -1;
            }
            if (h->aux_cap < 0) aux_bytes = 0;
        }
        if (aux_bytes) kern = kern_aux;
        if (h->dynamic_rows && phase == 0) {
            long cap = aux_bytes ? std::min(h->aux_cap, h->qp_cap) : h->qp_cap;
            if (h->max_waves > 0 && 4L * h->max_waves < cap) cap = 4L * h->max_waves; // option "max_waves": fewer resident waves
            if (cap > 0 && cap < qp_groups) { ng = cap; q0 = (int)ng; }
        }
        if (q0 >= 0) HIP_TRY(h, hipMemsetAsync(h->ptrs.queue, 0, sizeof(int), h->stream));
        // Hand-over of long runners (qp_ipm.hpp, QpIpm::suspend): a launch that refills from the queue ends with a few rows finishing
        // instances of 30 - 50 iterations on an idling device; past "handover_iter" iterations those go to a follow-up launch on the
        // latency mapping (one instance per wave over the same planes: 1.6x per pass for usv_model_pf_ca at N = 40).  Scheduling only.
        bool hand = false;
        size_t xbytes = (size_t)4 * wide.ex_hbm * 128;
        qp_resume_t kern_resume = wide.resume;
        int hand_it = 0;
        if (phase == 0 && h->handover_iter != 0 && wide.resume != nullptr && (long)std::min(h->N + 1, 4) * h->Bp * h->spec.npt * 128 < (1L << 32)) {
            if (h->resume_cap == 0) {
                // (planes in LDS when the horizon fits - option "handover_lds", default on -, else over the planes in HBM)
                int nb = 0;
                hipFuncAttributes fa;
                const size_t lbytes = (size_t)(h->N + 1) * (size_t)wide.nplw * 128 + (size_t)4 * wide.ex_lds * 128;
                h->resume_lds = false;
                if (h->handover_lds && wide.resume_lds != nullptr && lbytes <= 160u * 1024u && hipFuncGetAttributes(&fa, (const void *)wide.resume_lds) == hipSuccess &&
                    fa.sharedSizeBytes + lbytes <= 160u * 1024u &&
                    hipFuncSetAttribute((const void *)wide.resume_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lbytes) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wide.resume_lds, qp_block, lbytes) == hipSuccess && nb > 0 && h->ncu > 0) {
                    h->resume_cap = (long)std::min(nb, 4) * h->ncu;
                    h->resume_lds = true;
                } else {
                    h->resume_cap = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wide.resume, qp_block, xbytes) == hipSuccess && nb > 0 && h->ncu > 0)
                                        ? (long)std::min(nb, 4) * h->ncu : -1;
                }
            }
            if (h->resume_lds) { kern_resume = wide.resume_lds; xbytes = (size_t)(h->N + 1) * (size_t)wide.nplw * 128 + (size_t)4 * wide.ex_lds * 128; }
            if (h->resume_cap > 0 && !h->d_susp_list) {
                if (dev_alloc(h, &h->d_susp_count, (size_t)usvmpc_handle::RING, true) || dev_alloc(h, &h->d_susp_list, (size_t)h->B, false) ||
                    dev_alloc(h, &h->d_susp_rec, (size_t)h->B * 4, false))
                    return USVMPC_E_HIP;
            }
            // default (-1): past 20 iterations when the follow-up works in LDS AND the batch is at most three times what the device holds at once
            // (re-measured in round 6 under the default QP solver profile, whose solves are shorter - profiles/r06_handover_co.txt: with the
            // follow-up kernel beside the launch -18 % per tick at 4 096 instances, -13 % at 8 192, -4 % at 16 384, 0 at 32 768, +1 % at
            // 65 536; with it only behind the launch nothing is gained any more at any size), never when it would run over the planes in HBM (a
            // loss: profiles/r05_handover.txt)
            const bool small = h->qp_cap > 0 && (long)h->B <= 3 * h->qp_cap;
            hand_it = h->handover_iter > 0 ? h->handover_iter : ((h->resume_lds && small) ? 20 : 0);
            hand = h->resume_cap > 0 && hand_it > 0;
        }
        h->ptrs.susp_count = hand ? h->d_susp_count + h->nsolves % usvmpc_handle::RING : nullptr;
        h->ptrs.susp_list = hand ? h->d_susp_list : nullptr;
        h->ptrs.susp_rec = hand ? h->d_susp_rec : nullptr;
        h->ptrs.handover_iter = hand ? hand_it : 0;
        const dim3 qg((unsigned)((ng * LANES + qp_block - 1) / qp_block)), qb(qp_block);
        // The follow-up kernel BESIDE the draining launch (usv_qp_resume_co): on a stream of its own, eligible together with the main launch;
        // what it does not get to is done by the follow-up launch behind the main one.  With the planes copied into LDS only (the form that pays).
        const bool co = hand && h->handover_co != 0 && h->resume_lds && wide.resume_co != nullptr && h->own_stream;
        h->ptrs.co_ctl = nullptr;
        if (co) {
            if (!h->co_stream) {
                int prio_least = 0, prio_greatest = 0;
                HIP_TRY(h, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
                HIP_TRY(h, hipStreamCreateWithPriority(&h->co_stream, hipStreamNonBlocking, prio_least));
                HIP_TRY(h, hipEventCreateWithFlags(&h->ev_co_pre, hipEventDisableTiming));
                HIP_TRY(h, hipEventCreateWithFlags(&h->ev_co_end, hipEventDisableTiming));
                if (dev_alloc(h, &h->d_co_ctl, (size_t)usvmpc_handle::RING * 8, true)) return USVMPC_E_HIP;
            }
            HIP_TRY(h, hipFuncSetAttribute((const void *)wide.resume_co, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xbytes)); // (per kernel, not per handle: handles differ in horizon)
            h->ptrs.co_ctl = h->d_co_ctl + 8 * (h->nsolves % usvmpc_handle::RING);
            HIP_TRY(h, hipMemsetAsync(h->ptrs.co_ctl, 0, 8 * sizeof(int), h->stream));
            HIP_TRY(h, hipMemsetAsync(h->d_susp_list, 0xff, (size_t)h->B * sizeof(int), h->stream)); // (-1: no entry yet)
            HIP_TRY(h, hipEventRecord(h->ev_co_pre, h->stream));
        }
        hipLaunchKernelGGL(kern, qg, qb, aux_bytes, h->stream, h->ptrs, ng, phase, q0, 4);
        if (co) {
            hipLaunchKernelGGL(usv_co_done, dim3(1), dim3(1), 0, h->stream, h->ptrs.co_ctl);
            HIP_TRY(h, hipStreamWaitEvent(h->co_stream, h->ev_co_pre, 0));
            // One follow-up workgroup per CU unless the caller asks otherwise (option "handover_co_wgs"): what finds room BESIDE the main launch's
            // workgroups at once (75 KB of LDS next to their eight times 10 KB).  With two per CU - what fits once the main launch has left - some
            // of them wait to be placed while the main launch runs, and about one tick in 1 500 then stalled until their waits ran out: the main
            // launch took 410 ms instead of 9 (tools/co_soak.py, docs/rounds/r06.md section 8: 0 stalls in 16 000 ticks with one per CU, same pace).
            long nco = std::min<long>(h->resume_cap, (long)h->B);
            nco = std::min<long>(nco, h->co_wgs > 0 ? (long)h->co_wgs : (long)std::max(h->ncu, 1));
            hipLaunchKernelGGL(wide.resume_co, dim3((unsigned)nco), dim3(qp_block), xbytes, h->co_stream, h->ptrs, (int)qg.x, (int)h->B, h->co_spin_limit);
            HIP_TRY(h, hipEventRecord(h->ev_co_end, h->co_stream));
        }
        if (hand) {
            HIP_TRY(h, hipEventRecord(ev[3], h->stream));
            h->ev3_set[h->nsolves % usvmpc_handle::RING] = true;
            const long nwg = std::min<long>(h->resume_cap, (long)h->B);
            hipLaunchKernelGGL(kern_resume, dim3((unsigned)nwg), dim3(qp_block), xbytes, h->stream, h->ptrs);
        }
        if (co) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_co_end, 0)); // (the tick's QPs are solved when both kernels are through)
        return 0;
    };
    int rcq = 0;
    if (cond) {
        rcq = launch_cond(h);
    } else {
#ifdef USV_BENCH_ONLY // development builds (tools/dev_build.sh): only the instantiation the bench workload runs
    if (!(h->spec.hdiag && pack && !h->spec.any_bsoft)) { h->err = "development build: bench instantiation only"; return USVMPC_E_ARG; }
    // (one row pass when every box row rides in a slot lane: qp_ipm.hpp, MERGE)
    if (h->merge_rows && !h->spec.box_dense)
        rcq = launch_qp(&usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, false, CANPACK>, &usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, true, CANPACK>,
                        &usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, false, CANPACK, true>, wide_set<M, KCH, SOFT, CANPACK>());
    else
        rcq = launch_qp(&usv_qp_rti<M, KCH, SOFT, true, CANPACK, false>, &usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, true>,
                        &usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, false, false, true>);
#else
    if (h->spec.any_bsoft) { // soft state bounds: rows with slacks, ten planes of their own
        rcq = h->spec.hdiag ? launch_qp(&usv_qp_rti<M, KCH, SOFT, true, false, true>, nullptr, nullptr, wide_set<M, KCH, SOFT, false, true>())
                            : launch_qp(&usv_qp_rti<M, KCH, SOFT, false, false, true>, nullptr);
    } else if (h->spec.hdiag) { // (every OCP of the reference: the only instantiations that also come with the workspace in LDS)
        // (one row pass when every box row rides in a slot lane: qp_ipm.hpp, MERGE)
        // (the packed layouts - every OCP of the reference, the bench workloads - also come with the aux plane in LDS)
        if (pack && h->merge_rows && !h->spec.box_dense)
            rcq = launch_qp(&usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, false, CANPACK>, &usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, true, CANPACK>,
                            &usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, false, CANPACK, true>, wide_set<M, KCH, SOFT, CANPACK>());
        else
            rcq = pack ? launch_qp(&usv_qp_rti<M, KCH, SOFT, true, CANPACK, false>, &usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, true>,
                                   &usv_qp_rti<M, KCH, SOFT, true, CANPACK, false, false, false, true>, wide_set<M, KCH, SOFT, false>())
                       : launch_qp(&usv_qp_rti<M, KCH, SOFT, true, false, false>, &usv_qp_rti<M, KCH, SOFT, true, false, false, true>, nullptr,
                                   wide_set<M, KCH, SOFT, false, false, true>());   // (box rows in planes of their own)
    } else {
        rcq = pack ? launch_qp(&usv_qp_rti<M, KCH, SOFT, false, CANPACK, false>, nullptr) : launch_qp(&usv_qp_rti<M, KCH, SOFT, false, false, false>, nullptr);
    }
#endif
    }
    if (rcq) return rcq;
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(ev[2], h->stream));
    if (phase == 0) { // (an RTI solve: one QP per instance)
        hipLaunchKernelGGL(usv_count_unconverged, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->ptrs.qp_status, B,
                           h->d_unconv_ring + h->nsolves % usvmpc_handle::RING, h->d_unconv_total);
        HIP_TRY(h, hipGetLastError());
    }
    if (spec_next) {
        // the next tick's lineariser, behind this launch on the second stream, under the map made above.  (Its stream has the lowest
        // priority: when both become eligible the QP launch's workgroups are placed first; groups it reaches before their instance is
        // final are marked and redone by the fix-up pass - measured on the bench workload: 0.35 of 7.4 ms, ~5 % of the groups.)
        HIP_TRY(h, hipStreamWaitEvent(h->aux_stream, h->ev_pre, 0));
        DevPtrs Pn = h->ptrs;
        Pn.perm = next_perm;
        Pn.perm_cur = h->ptrs.perm;
        lin_launch(std::integral_constant<int, 1>{}, h->aux_stream, Pn);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(h->ev_spec, h->aux_stream));
        h->spec_for = h->nsolves + 1;
        h->spec_valid = true;
        h->spec_outstanding = true;
        h->spec_perm = next_perm;
    }
    h->nsolves++;
    h->layout_dirty = false;
    h->last_cond = cond;
    if (h->mirror) { // [x | u | status] of this solve, one copy
        HIP_TRY(h, hipMemcpyAsync(h->mirror, h->arena, h->f_off[usvmpc_handle::F_X0], hipMemcpyDeviceToHost, h->stream));
        h->inflight = true;
        h->out_valid = true;
    }
    return 0;
}

// ---- multiplier read-back: buffers on first use, one kernel per solve they are asked for
template <class M, int KCH, bool SOFT>
int export_pair(usvmpc_handle *h)
{
    constexpr bool CANPACK = KCH > 0;
    const bool pack = CANPACK && h->spec.boxpack != 0;
    const dim3 grid((unsigned)((h->Bp + 3) / 4)), block(64);
    if (h->spec.any_bsoft) hipLaunchKernelGGL((usv_qp_export<M, KCH, SOFT, false, true>), grid, block, 0, h->stream, h->ptrs, (long)h->Bp);
    else if (pack) hipLaunchKernelGGL((usv_qp_export<M, KCH, SOFT, CANPACK, false>), grid, block, 0, h->stream, h->ptrs, (long)h->Bp);
    else hipLaunchKernelGGL((usv_qp_export<M, KCH, SOFT, false, false>), grid, block, 0, h->stream, h->ptrs, (long)h->Bp);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

// The instantiations of the stock library, one per build part
#if USV_PART >= 0 && !defined(USV_GEN_ONLY)
#if USV_PART == 1
#define USV_PAIR_1(M, KCH, SOFT) template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#else
#define USV_PAIR_1(M, KCH, SOFT) extern template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); extern template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#endif
#if USV_PART == 2
#define USV_PAIR_2(M, KCH, SOFT) template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#else
#define USV_PAIR_2(M, KCH, SOFT) extern template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); extern template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#endif
#if USV_PART == 3
#define USV_PAIR_3(M, KCH, SOFT) template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#else
#define USV_PAIR_3(M, KCH, SOFT) extern template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); extern template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#endif
#if USV_PART == 4
#define USV_PAIR_4(M, KCH, SOFT) template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#else
#define USV_PAIR_4(M, KCH, SOFT) extern template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); extern template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#endif
#if USV_PART == 5
#define USV_PAIR_5(M, KCH, SOFT) template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#else
#define USV_PAIR_5(M, KCH, SOFT) extern template int launch_pair<M, KCH, SOFT>(usvmpc_handle *, int); extern template int export_pair<M, KCH, SOFT>(usvmpc_handle *);
#endif
#ifndef USV_BENCH_ONLY
USV_PAIR_1(ModelM0, 0, false)
USV_PAIR_3(ModelM1, 2, true)
USV_PAIR_5(ModelM2, 2, false)
#endif
USV_PAIR_2(ModelM1, 1, true)
USV_PAIR_4(ModelM2, 1, false)
#endif

#if USV_MAIN
namespace {

// planes per stage of the packed [B A] for this model (MatPack)
int model_mat_planes(int model)
{
    switch (model) {
#ifndef USV_GEN_ONLY
    case USVMPC_MODEL_USV: return MatPack<ModelM0>::NPK;
    case USVMPC_MODEL_GUIDANCE_CA1: return MatPack<ModelM1>::NPK;
    case USVMPC_MODEL_PF_CA: return MatPack<ModelM2>::NPK;
#endif
#ifdef USV_GEN_MODEL_HEADER
    case USVMPC_MODEL_GENERATED: return MatPack<ModelGen>::NPK;
#endif
    }
    return 0;
}

int launch(usvmpc_handle *h, int phase = 0)
{
    switch (h->desc.model) {
#ifdef USV_BENCH_ONLY
    case USVMPC_MODEL_PF_CA: if (h->kch <= 1) return launch_pair<ModelM2, 1, false>(h, phase); break;
    case USVMPC_MODEL_GUIDANCE_CA1: if (h->kch <= 1) return launch_pair<ModelM1, 1, true>(h, phase); break;
#elif !defined(USV_GEN_ONLY)
    case USVMPC_MODEL_USV: return launch_pair<ModelM0, 0, false>(h, phase);
    case USVMPC_MODEL_GUIDANCE_CA1:
        return h->kch <= 1 ? launch_pair<ModelM1, 1, true>(h, phase) : launch_pair<ModelM1, 2, true>(h, phase);
    case USVMPC_MODEL_PF_CA:
        return h->kch <= 1 ? launch_pair<ModelM2, 1, false>(h, phase) : launch_pair<ModelM2, 2, false>(h, phase);
#endif
#if defined(USV_GEN_MODEL_HEADER) && !defined(USV_BENCH_ONLY)
    case USVMPC_MODEL_GENERATED: return launch_pair<ModelGen, USV_GEN_KCH, (USV_GEN_SOFT != 0)>(h, phase);
#endif
    }
    h->err = "unknown model";
    return USVMPC_E_ARG;
}

int ensure_export(usvmpc_handle *h)
{
    if (h->nsolves == 0) { h->err = "no QP has been solved yet: nothing to read back"; return USVMPC_E_ARG; }
    if (h->layout_dirty) { h->err = "the row layout option changed after the last solve: solve again before reading multipliers"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    DevPtrs &P = h->ptrs;
    if (!P.lam_out) {
        P.nlam = lam_len(h->spec, h->soft);
        const size_t cnt = (size_t)h->B * (h->N + 1) * (size_t)(P.nlam > 0 ? P.nlam : 1);
        if (dev_alloc(h, &P.lam_out, cnt, false)) return USVMPC_E_HIP;
        if (dev_alloc(h, &P.t_out, cnt, false)) { P.lam_out = nullptr; return USVMPC_E_HIP; }
        h->export_at = -1;
    }
    if (h->export_at == h->nsolves) return 0;
    if (h->last_cond) {
        // the partially condensed solve keeps its rows in per-workgroup scratch: it writes "lam" / "t" itself, when the buffers exist
        h->err = "\"lam\" / \"t\" of a partially condensed solve (qp_cond_N) are written by the solve itself: the buffers exist from now on "
                 "(option \"keep_multipliers\" = 1 creates them up front) - solve again";
        return USVMPC_E_ARG;
    }
    const size_t nbytes = (size_t)h->B * (h->N + 1) * (size_t)(P.nlam > 0 ? P.nlam : 1) * sizeof(double);
    HIP_TRY(h, hipMemsetAsync(P.lam_out, 0, nbytes, h->stream));
    HIP_TRY(h, hipMemsetAsync(P.t_out, 0, nbytes, h->stream));
    int rc = USVMPC_E_ARG;
    switch (h->desc.model) {
#ifdef USV_BENCH_ONLY
    case USVMPC_MODEL_PF_CA: if (h->kch <= 1) rc = export_pair<ModelM2, 1, false>(h); break;
    case USVMPC_MODEL_GUIDANCE_CA1: if (h->kch <= 1) rc = export_pair<ModelM1, 1, true>(h); break;
#elif !defined(USV_GEN_ONLY)
    case USVMPC_MODEL_USV: rc = export_pair<ModelM0, 0, false>(h); break;
    case USVMPC_MODEL_GUIDANCE_CA1: rc = h->kch <= 1 ? export_pair<ModelM1, 1, true>(h) : export_pair<ModelM1, 2, true>(h); break;
    case USVMPC_MODEL_PF_CA: rc = h->kch <= 1 ? export_pair<ModelM2, 1, false>(h) : export_pair<ModelM2, 2, false>(h); break;
#endif
#if defined(USV_GEN_MODEL_HEADER) && !defined(USV_BENCH_ONLY)
    case USVMPC_MODEL_GENERATED: rc = export_pair<ModelGen, USV_GEN_KCH, (USV_GEN_SOFT != 0)>(h); break;
#endif
    }
    if (rc) { if (rc == USVMPC_E_ARG) h->err = "multiplier read-back: no kernel for this model in this library"; return rc; }
    h->export_at = h->nsolves;
    return 0;
}

} // namespace

// ---------------------------------------------------------------------------------- C ABI
extern "C" {

int usvmpc_model_dims(int model, int *nx, int *nu)
{
    int a, b;
    if (model_dims(model, a, b)) return USVMPC_E_ARG;
    if (nx) *nx = a;
    if (nu) *nu = b;
    return 0;
}

void usvmpc_default_options(usvmpc_desc *d)
{
    if (d) default_options(*d);
}

int usvmpc_hpipm_profile(usvmpc_desc *d, int mode)
{
    return (d && hpipm_profile(*d, mode)) ? 0 : USVMPC_E_ARG;
}

int usvmpc_create(const usvmpc_desc *d, usvmpc_handle **out)
{
    if (!d || !out) return USVMPC_E_ARG;
    *out = nullptr;
    DevSpec S;
    const std::string err = build_spec(*d, S);
    if (!err.empty()) {
        g_create_err = err;
        std::fprintf(stderr, "usvmpc_create: %s\n", err.c_str());
        return USVMPC_E_ARG;
    }
    if (d->model != USVMPC_MODEL_GENERATED && (d->model == USVMPC_MODEL_GUIDANCE_CA1) != (d->soft != 0) && d->K > 0) {
        std::fprintf(stderr, "usvmpc_create: obstacle rows are soft for model 1 and hard for model 2\n");
        return USVMPC_E_ARG;
    }
#ifdef USV_GEN_MODEL_HEADER
    if (d->model == USVMPC_MODEL_GENERATED &&
        ((S.K + LANES - 1) / LANES != USV_GEN_KCH || (S.K > 0 && (d->soft != 0) != (USV_GEN_SOFT != 0)))) {
        std::fprintf(stderr, "usvmpc_create: this library was generated for %d obstacle chunk(s), soft = %d\n", USV_GEN_KCH, USV_GEN_SOFT);
        return USVMPC_E_ARG;
    }
#endif
#ifdef USV_GEN_ONLY
    if (d->model != USVMPC_MODEL_GENERATED) {
        std::fprintf(stderr, "usvmpc_create: this library only holds the generated model\n");
        return USVMPC_E_ARG;
    }
#endif
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || d->device < 0 || d->device >= ndev) {
        std::fprintf(stderr, "usvmpc_create: no usable HIP device (count %d, requested %d); there is no CPU fallback\n",
                     ndev, d->device);
        return USVMPC_E_NODEVICE;
    }
    usvmpc_handle *h = new usvmpc_handle();
    h->desc = *d;
    h->spec = S;
    model_dims(d->model, h->nx, h->nu);
    h->nz = h->nx + h->nu; h->ny = S.ny; h->ny_e = S.ny_e;
    h->N = S.N; h->K = S.K; h->B = S.B; h->Bp = S.Bp;
    h->kch = d->model == USVMPC_MODEL_USV ? 0 : ((S.K + LANES - 1) / LANES > 1 ? 2 : 1);
    h->soft = d->soft != 0 || d->model == USVMPC_MODEL_GUIDANCE_CA1;
#ifdef USV_GEN_MODEL_HEADER
    if (d->model == USVMPC_MODEL_GENERATED) { h->kch = USV_GEN_KCH; h->soft = USV_GEN_SOFT != 0; }
#endif
    h->device = d->device;
    h->bytes = 0; h->nsolves = 0; h->own_stream = true;
    h->map_changed = false;
    h->export_at = -1;
    h->layout_dirty = false;
    h->last_cond = false;
    h->pipeline = true;
    h->aux_stream = nullptr; h->ev_pre = nullptr; h->ev_spec = nullptr;
    h->d_epoch = nullptr; h->d_redo = nullptr; h->d_perm2 = nullptr;
    h->spec_for = -1; h->spec_valid = false; h->spec_outstanding = false; h->spec_perm = nullptr;
    h->spec_quiet = 0; h->spec_hits = 0; h->spec_misses = 0;
    h->noise_mask = ~0u;
    h->cond_N2 = 0; h->d_cond_dims = nullptr; h->d_cond_scratch = nullptr; h->cond_teams = 0; h->cond_lds = 0;
    h->dynamic_rows = true;
    h->qp_cap = 0;
    h->aux_lds = true;
    h->aux_cap = 0;
    h->lds_mode = -1;
    h->lds_cap = 0;
    h->wide_mode = -1; h->wide_cap = 0; h->wide_hbm_cap = 0; h->last_wide = 0;
    h->wide_waves = -1; h->wide4_cap = 0; h->wide4_hbm_cap = 0;
    h->handover_co = -1; h->co_spin_limit = 200000; h->co_wgs = 0; h->co_stream = nullptr; h->ev_co_pre = nullptr; h->ev_co_end = nullptr; h->d_co_ctl = nullptr;
    h->handover_iter = -1; for (bool &e : h->ev3_set) e = false; h->resume_lds = false; h->handover_lds = true; h->d_susp_count = nullptr; h->d_susp_list = nullptr; h->d_susp_rec = nullptr; h->resume_cap = 0;
    h->max_waves = 0;
    {
        hipDeviceProp_t prop;
        h->ncu = (hipGetDeviceProperties(&prop, d->device) == hipSuccess) ? prop.multiProcessorCount : 0;
    }
    std::memset(&h->ptrs, 0, sizeof(h->ptrs));
    auto fail = [&](int rc) {
        std::fprintf(stderr, "usvmpc_create: %s\n", h->err.c_str());
        for (void *a : h->allocs) (void)hipFree(a);
        if (h->mirror) (void)hipHostFree(h->mirror);
        delete h;
        return rc;
    };
#define TRY_C(x) do { int rc_ = (x); if (rc_) return fail(rc_); } while (0)
#define HIP_C(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hipGetErrorString(e_); return fail(USVMPC_E_HIP); } } while (0)
    HIP_C(hipSetDevice(h->device));
    HIP_C(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (int r = 0; r < usvmpc_handle::RING; r++)
        for (int i = 0; i < 4; i++) HIP_C(hipEventCreate(&h->ev[r][i]));
    const size_t B = h->B, N = h->N, K = h->K;
    const size_t stride = (size_t)h->Bp * LANES;
    const size_t kch = h->kch ? h->kch : 1;
    DevPtrs &P = h->ptrs;
    h->spec.npt = ws_planes(h->nx, h->nu, h->kch, h->soft, model_mat_planes(h->desc.model), h->spec.any_bsoft != 0);
    // one stage of the workspace is addressed through ONE 32-bit buffer window (lanes::Planes: group * npt * 128 + ...)
    if ((size_t)h->Bp * (size_t)h->spec.npt * 128u > (size_t)UINT32_MAX - 16384u) { // (16 KB of headroom: a parked row addresses just past the window - qp_ipm.hpp)
        h->err = "batch too large for one handle: batch * " + std::to_string(h->spec.npt * 128) + " bytes per stage exceed the 4 GiB "
                 "buffer window (at most " + std::to_string((size_t)UINT32_MAX / ((size_t)h->spec.npt * 128u) / 4 * 4) + " instances); use several handles";
        return fail(USVMPC_E_ARG);
    }
    TRY_C(dev_alloc(h, &h->d_spec, 1, false));
    HIP_C(hipMemcpy(h->d_spec, &h->spec, sizeof(DevSpec), hipMemcpyHostToDevice));
    P.spec = h->d_spec;
    {   // the caller-visible arrays: one arena [x | u | status | x0 | yref | yref_e | p | lh]
        const size_t len[usvmpc_handle::F_COUNT] = {B * (N + 1) * h->nx * sizeof(double), B * N * h->nu * sizeof(double), B * sizeof(int),
                                                    B * h->nx * sizeof(double), B * N * h->ny * sizeof(double), B * h->ny_e * sizeof(double),
                                                    B * (N + 1) * 2 * K * sizeof(double), B * N * K * sizeof(double)};
        size_t off = 0;
        for (int f = 0; f < usvmpc_handle::F_COUNT; f++) {
            h->f_off[f] = off; h->f_len[f] = len[f];
            off += (len[f] + 255) / 256 * 256;
            h->dirty_lo[f] = 1; h->dirty_hi[f] = 0;
        }
        h->f_off[usvmpc_handle::F_COUNT] = off;
        h->arena_bytes = off;
        TRY_C(dev_alloc(h, &h->arena, off, true));
        char *a = h->arena;
        P.x = (double *)(a + h->f_off[usvmpc_handle::F_X]);
        P.u = (double *)(a + h->f_off[usvmpc_handle::F_U]);
        P.status = (int *)(a + h->f_off[usvmpc_handle::F_STATUS]);
        P.x0 = (double *)(a + h->f_off[usvmpc_handle::F_X0]);
        P.yref = (double *)(a + h->f_off[usvmpc_handle::F_YREF]);
        P.yref_e = (double *)(a + h->f_off[usvmpc_handle::F_YREF_E]);
        P.p = (double *)(a + h->f_off[usvmpc_handle::F_P]);
        P.lh = (double *)(a + h->f_off[usvmpc_handle::F_LH]);
        h->mirror = nullptr;
        if (off <= MIRROR_MAX) {
            HIP_C(hipHostMalloc((void **)&h->mirror, off, hipHostMallocDefault));
            std::memset(h->mirror, 0, off);
        }
        h->inflight = false; h->out_valid = false; h->extern_access = false;
    }
    TRY_C(dev_alloc(h, &P.sl, B * N * K, true));
    TRY_C(dev_alloc(h, &P.su, B * N * K, true));
    TRY_C(dev_alloc(h, &P.pi, B * N * h->nx, true));
    TRY_C(dev_alloc(h, &P.qp_iter, B, true));
    TRY_C(dev_alloc(h, &P.qp_status, B, true));
    TRY_C(dev_alloc(h, &P.res, B * 4, true));
    TRY_C(dev_alloc(h, &P.obs_tmin, B, true));
    TRY_C(dev_alloc(h, &h->d_fail_ring, usvmpc_handle::RING, true));
    TRY_C(dev_alloc(h, &h->d_unconv_ring, usvmpc_handle::RING, true));
    TRY_C(dev_alloc(h, &h->d_unconv_total, 1, true));
    TRY_C(dev_alloc(h, &P.queue, 1, true));
    TRY_C(dev_alloc(h, &P.nlp_res, B * 4, true));
    TRY_C(dev_alloc(h, &P.sqp_iter, B, true));
    TRY_C(dev_alloc(h, &P.sqp_state, B, true));
    TRY_C(dev_alloc(h, &P.sqp_running, 1, true));
    TRY_C(dev_alloc(h, &h->d_perm, B, true));
    TRY_C(dev_alloc(h, &h->d_iter_prev, B, true));
    TRY_C(dev_alloc(h, &h->d_hist, SORT_BINS, true));
    TRY_C(dev_alloc(h, &h->d_cursor, SORT_BINS, true));
    h->sort_enabled = true;
    h->sort_two = false;
    h->merge_rows = true;
    h->gd_ready = false; h->gd_npts_cap = 0; h->gd_psi = nullptr; h->gd_world = nullptr; h->gd_world_cap = 0;
    std::memset(&h->gd, 0, sizeof(h->gd));
    TRY_C(dev_alloc(h, &P.ws, (N + 1) * (size_t)ws_planes(h->nx, h->nu, h->kch, h->soft, model_mat_planes(h->desc.model), h->spec.any_bsoft != 0) * stride, true));
    HIP_C(hipDeviceSynchronize());
#undef TRY_C
#undef HIP_C
    *out = h;
    return 0;
}

int usvmpc_destroy(usvmpc_handle *h)
{
    if (!h) return USVMPC_E_ARG;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->aux_stream) {
        (void)hipStreamSynchronize(h->aux_stream);
        (void)hipStreamDestroy(h->aux_stream);
        (void)hipEventDestroy(h->ev_pre);
        (void)hipEventDestroy(h->ev_spec);
    }
    if (h->co_stream) {
        (void)hipStreamSynchronize(h->co_stream);
        (void)hipStreamDestroy(h->co_stream);
        (void)hipEventDestroy(h->ev_co_pre);
        (void)hipEventDestroy(h->ev_co_end);
    }
    for (void *a : h->allocs) (void)hipFree(a);
    if (h->mirror) (void)hipHostFree(h->mirror);
    for (int r = 0; r < usvmpc_handle::RING; r++)
        for (int i = 0; i < 4; i++) (void)hipEventDestroy(h->ev[r][i]);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int usvmpc_set(usvmpc_handle *h, const char *field, int stage, const double *v, size_t n)
{
    return copy_field(h, field, stage, const_cast<double *>(v), n, true);
}

int usvmpc_get(usvmpc_handle *h, const char *field, int stage, double *out, size_t n)
{
    return copy_field(h, field, stage, out, n, false);
}

int usvmpc_get_int(usvmpc_handle *h, const char *field, int *out)
{
    if (!h || !out) return USVMPC_E_ARG;
    const std::string s(field ? field : "");
    const int *src = s == "status" ? h->ptrs.status : s == "qp_iter" ? h->ptrs.qp_iter : s == "qp_status" ? h->ptrs.qp_status
                     : s == "sqp_iter" ? h->ptrs.sqp_iter : nullptr;
    if (!src) { h->err = "unknown integer field '" + s + "'"; return USVMPC_E_FIELD; }
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(out, src, (size_t)h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return 0;
}

int usvmpc_solve_async(usvmpc_handle *h)
{
    if (!h) return USVMPC_E_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    return launch(h);
}

int usvmpc_sync(usvmpc_handle *h)
{
    if (!h) return USVMPC_E_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->spec_outstanding) { // (the lineariser that runs a tick ahead is part of the work enqueued so far)
        HIP_TRY(h, hipStreamSynchronize(h->aux_stream));
        h->spec_outstanding = false;
    }
    h->inflight = false;
    return 0;
}

int usvmpc_solve(usvmpc_handle *h, int *status)
{
    int rc = usvmpc_solve_async(h);
    if (rc) return rc;
    rc = usvmpc_sync(h);
    if (rc) return rc;
    int worst = 0;
    if (status) {
        if (h->mirror && h->out_valid) std::memcpy(status, h->mirror + h->f_off[usvmpc_handle::F_STATUS], (size_t)h->B * sizeof(int)); // (came back with x | u)
        else {
            rc = usvmpc_get_int(h, "status", status);
            if (rc) return rc;
        }
        for (int b = 0; b < h->B; b++) worst = status[b] > worst ? status[b] : worst;
    }
    return worst;
}

int usvmpc_solve_sqp(usvmpc_handle *h, int *status)
{
    if (!h) return USVMPC_E_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    const int B = h->B;
    const int max_iter = h->desc.nlp_max_iter > 0 ? h->desc.nlp_max_iter : 100;
    {
        const int rcs = spec_cancel(h);
        if (rcs) return rcs;
    }
    hipLaunchKernelGGL(usv_sqp_begin, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->ptrs, B);
    HIP_TRY(h, hipGetLastError());
    for (int it = 0; it < max_iter; it++) {
        HIP_TRY(h, hipMemsetAsync(h->ptrs.sqp_running, 0, sizeof(int), h->stream));
        // the very first launch of a handle has no multipliers yet; afterwards the workspace holds those of the
        // last QP of every instance (acados likewise keeps nlp_out between calls)
        // ... unless the group -> instance map has changed since (difficulty binning re-sorted or switched off): the
        // workspace is group-indexed, so the multipliers there belong to other instances and the call starts from
        // zero multipliers like a first one
        const bool fresh = it == 0 && (h->nsolves == 0 || h->map_changed);
        if (it == 0) h->map_changed = false;
        const int rc = launch(h, fresh ? 1 : 2);
        if (rc) return rc;
        int running = 0;
        HIP_TRY(h, hipMemcpyAsync(&running, h->ptrs.sqp_running, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (running == 0) break; // every instance has converged or failed
    }
    hipLaunchKernelGGL(usv_sqp_end, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->ptrs, B);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int worst = 0;
    if (status) {
        const int rc = usvmpc_get_int(h, "status", status);
        if (rc) return rc;
        for (int b = 0; b < B; b++) worst = status[b] > worst ? status[b] : worst;
    }
    return worst;
}

int usvmpc_get_device_ptr(usvmpc_handle *h, const char *field, void **dptr)
{
    if (!h || !dptr) return USVMPC_E_ARG;
    const std::string s(field ? field : "");
    const DevPtrs &P = h->ptrs;
    const void *p = s == "x" ? (const void *)P.x : s == "u" ? (const void *)P.u : s == "x0" ? (const void *)P.x0
                  : s == "yref" ? (const void *)P.yref : s == "yref_e" ? (const void *)P.yref_e
                  : s == "p" ? (const void *)P.p : s == "lh" ? (const void *)P.lh : s == "pi" ? (const void *)P.pi
                  : s == "sl" ? (const void *)P.sl : s == "su" ? (const void *)P.su
                  : s == "status" ? (const void *)P.status : s == "qp_iter" ? (const void *)P.qp_iter
                  : s == "qp_status" ? (const void *)P.qp_status : s == "res" ? (const void *)P.res
                  : s == "obs_tmin" ? (const void *)P.obs_tmin
                  : s == "nlp_res" ? (const void *)P.nlp_res : s == "sqp_iter" ? (const void *)P.sqp_iter : nullptr;
    if (!p) { h->err = "unknown field '" + s + "'"; return USVMPC_E_FIELD; }
    {   // the caller is about to read (or write) device memory directly: pending host writes go up, and the host mirror no longer
        // vouches for the device's x / u
        HIP_TRY(h, hipSetDevice(h->device));
        const int rcf = mirror_flush(h);
        if (rcf) return rcf;
        h->extern_access = true;
        const int rcs = spec_cancel(h); // (and no lineariser runs ahead on arrays the caller may write behind the library's back)
        if (rcs) return rcs;
    }
    *dptr = const_cast<void *>(p);
    return 0;
}

int usvmpc_kernel_ms(usvmpc_handle *h, int n, float *linearize_ms, float *qp_ms)
{
    if (!h || n < 1) return USVMPC_E_ARG;
    if (h->nsolves < n || n > usvmpc_handle::RING) { h->err = "fewer solves recorded than requested"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    for (int i = 0; i < n; i++) { // oldest of the last n first
        hipEvent_t *ev = h->ev[(h->nsolves - n + i) % usvmpc_handle::RING];
        HIP_TRY(h, hipEventSynchronize(ev[2]));
        float a = 0, b = 0;
        HIP_TRY(h, hipEventElapsedTime(&a, ev[0], ev[1]));
        HIP_TRY(h, hipEventElapsedTime(&b, ev[1], ev[2]));
        if (linearize_ms) linearize_ms[i] = a;
        if (qp_ms) qp_ms[i] = b;
    }
    return 0;
}

int usvmpc_followup_ms(usvmpc_handle *h, int n, float *ms)
{
    if (!h || n < 1 || !ms) return USVMPC_E_ARG;
    if (h->nsolves < n || n > usvmpc_handle::RING) { h->err = "fewer solves recorded than requested"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    for (int i = 0; i < n; i++) { // oldest of the last n first
        const int r = (h->nsolves - n + i) % usvmpc_handle::RING;
        ms[i] = 0.0f;
        if (!h->ev3_set[r]) continue;
        HIP_TRY(h, hipEventSynchronize(h->ev[r][2]));
        HIP_TRY(h, hipEventElapsedTime(&ms[i], h->ev[r][3], h->ev[r][2]));
    }
    return 0;
}

int usvmpc_tick_ms(usvmpc_handle *h, int n, float *ms)
{
    if (!h || n < 2 || !ms) return USVMPC_E_ARG;
    if (h->nsolves < n || n > usvmpc_handle::RING) { h->err = "fewer solves recorded than requested"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    for (int i = 0; i + 1 < n; i++) { // oldest of the last n first
        hipEvent_t *ev = h->ev[(h->nsolves - n + i) % usvmpc_handle::RING], *evn = h->ev[(h->nsolves - n + i + 1) % usvmpc_handle::RING];
        HIP_TRY(h, hipEventSynchronize(evn[0]));
        HIP_TRY(h, hipEventElapsedTime(&ms[i], ev[0], evn[0]));
    }
    return 0;
}

int usvmpc_debug_model_eval(int model, int device, int n, const double *x, const double *u, double *f, double *J)
{
    int nx, nu;
    if (model_dims(model, nx, nu) || n < 1 || !x || !f || !J || (nu > 0 && !u)) return USVMPC_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return USVMPC_E_NODEVICE;
    const int nz = nx + nu;
    double *dx = nullptr, *du = nullptr, *df = nullptr, *dJ = nullptr;
    int rc = 0;
    if (hipMalloc((void **)&dx, sizeof(double) * n * nx) != hipSuccess || hipMalloc((void **)&du, sizeof(double) * n * (nu ? nu : 1)) != hipSuccess ||
        hipMalloc((void **)&df, sizeof(double) * n * nx) != hipSuccess || hipMalloc((void **)&dJ, sizeof(double) * n * nx * nz) != hipSuccess)
        rc = USVMPC_E_HIP;
    if (!rc) {
        (void)hipMemcpy(dx, x, sizeof(double) * n * nx, hipMemcpyHostToDevice);
        if (nu) (void)hipMemcpy(du, u, sizeof(double) * n * nu, hipMemcpyHostToDevice);
        const dim3 grid((unsigned)((n * nz + 63) / 64)), block(64);
        switch (model) {
#ifndef USV_GEN_ONLY
        case USVMPC_MODEL_USV: hipLaunchKernelGGL(usv_debug_model<ModelM0>, grid, block, 0, 0, n, dx, du, df, dJ); break;
        case USVMPC_MODEL_GUIDANCE_CA1: hipLaunchKernelGGL(usv_debug_model<ModelM1>, grid, block, 0, 0, n, dx, du, df, dJ); break;
        case USVMPC_MODEL_PF_CA: hipLaunchKernelGGL(usv_debug_model<ModelM2>, grid, block, 0, 0, n, dx, du, df, dJ); break;
#endif
#ifdef USV_GEN_MODEL_HEADER
        case USVMPC_MODEL_GENERATED: hipLaunchKernelGGL(usv_debug_model<ModelGen>, grid, block, 0, 0, n, dx, du, df, dJ); break;
#endif
        default: rc = USVMPC_E_ARG;
        }
        if (!rc && (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess)) rc = USVMPC_E_HIP;
        if (!rc) {
            (void)hipMemcpy(f, df, sizeof(double) * n * nx, hipMemcpyDeviceToHost);
            (void)hipMemcpy(J, dJ, sizeof(double) * n * nx * nz, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(dx); (void)hipFree(du); (void)hipFree(df); (void)hipFree(dJ);
    return rc;
}

int usvmpc_debug_obstacle_eval(int device, int n, int K, const double *pos, const double *p, double *h, double *grad)
{
    if (n < 1 || K < 1 || !pos || !p || !h || !grad) return USVMPC_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return USVMPC_E_NODEVICE;
    double *dpos = nullptr, *dp = nullptr, *dh = nullptr, *dg = nullptr;
    int rc = 0;
    if (hipMalloc((void **)&dpos, sizeof(double) * n * 2) != hipSuccess || hipMalloc((void **)&dp, sizeof(double) * n * 2 * K) != hipSuccess ||
        hipMalloc((void **)&dh, sizeof(double) * n * K) != hipSuccess || hipMalloc((void **)&dg, sizeof(double) * n * K * 2) != hipSuccess)
        rc = USVMPC_E_HIP;
    if (!rc) {
        (void)hipMemcpy(dpos, pos, sizeof(double) * n * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(dp, p, sizeof(double) * n * 2 * K, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(usv_debug_obstacle, dim3((unsigned)((n * K + 63) / 64)), dim3(64), 0, 0, n, K, dpos, dp, dh, dg);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = USVMPC_E_HIP;
        if (!rc) {
            (void)hipMemcpy(h, dh, sizeof(double) * n * K, hipMemcpyDeviceToHost);
            (void)hipMemcpy(grad, dg, sizeof(double) * n * K * 2, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(dpos); (void)hipFree(dp); (void)hipFree(dh); (void)hipFree(dg);
    return rc;
}

int usvmpc_fail_counts(usvmpc_handle *h, int n, int *counts)
{
    if (!h || n < 1 || !counts) return USVMPC_E_ARG;
    if (h->nsolves < n || n > usvmpc_handle::RING) { h->err = "fewer solves recorded than requested"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    int ring[usvmpc_handle::RING];
    HIP_TRY(h, hipMemcpyAsync(ring, h->d_fail_ring, sizeof(ring), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++) counts[i] = ring[(h->nsolves - n + i) % usvmpc_handle::RING];
    return 0;
}

int usvmpc_unconverged_counts(usvmpc_handle *h, int n, int *counts)
{
    if (!h || n < 1 || !counts) return USVMPC_E_ARG;
    if (h->nsolves < n || n > usvmpc_handle::RING) { h->err = "fewer solves recorded than requested"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    int ring[usvmpc_handle::RING];
    HIP_TRY(h, hipMemcpyAsync(ring, h->d_unconv_ring, sizeof(ring), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++) counts[i] = ring[(h->nsolves - n + i) % usvmpc_handle::RING];
    return 0;
}

int usvmpc_unconverged_total(usvmpc_handle *h, long long *total)
{
    if (!h || !total) return USVMPC_E_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    unsigned long long v = 0;
    HIP_TRY(h, hipMemcpyAsync(&v, h->d_unconv_total, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *total = (long long)v;
    return 0;
}

int usvmpc_handover_counts(usvmpc_handle *h, int n, int *counts)
{
    if (!h || n < 1 || !counts) return USVMPC_E_ARG;
    if (h->nsolves < n || n > usvmpc_handle::RING) { h->err = "fewer solves recorded than requested"; return USVMPC_E_ARG; }
    if (!h->d_susp_count) { for (int i = 0; i < n; i++) counts[i] = 0; return 0; } // (no launch of this handle has handed anything over)
    HIP_TRY(h, hipSetDevice(h->device));
    int ring[usvmpc_handle::RING];
    HIP_TRY(h, hipMemcpyAsync(ring, h->d_susp_count, sizeof(ring), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++) counts[i] = ring[(h->nsolves - n + i) % usvmpc_handle::RING];
    return 0;
}

int usvmpc_handover_co_counts(usvmpc_handle *h, int n, int *finished, int *timeouts)
{
    if (!h || n < 1 || !finished) return USVMPC_E_ARG;
    if (h->nsolves < n || n > usvmpc_handle::RING) { h->err = "fewer solves recorded than requested"; return USVMPC_E_ARG; }
    for (int i = 0; i < n; i++) { finished[i] = 0; if (timeouts) timeouts[i] = 0; }
    if (!h->d_co_ctl) return 0; // (no launch of this handle had a co-resident follow-up kernel)
    HIP_TRY(h, hipSetDevice(h->device));
    int ring[usvmpc_handle::RING * 8];
    HIP_TRY(h, hipMemcpyAsync(ring, h->d_co_ctl, sizeof(ring), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++) {
        const int r = (h->nsolves - n + i) % usvmpc_handle::RING;
        finished[i] = ring[8 * r + 3];
        if (timeouts) timeouts[i] = ring[8 * r + 4];
    }
    return 0;
}

int usvmpc_pipeline_stats(usvmpc_handle *h, long *used, long *discarded)
{
    if (!h) return USVMPC_E_ARG;
    if (used) *used = h->spec_hits;
    if (discarded) *discarded = h->spec_misses;
    return 0;
}

int usvmpc_last_mapping(usvmpc_handle *h, int *mapping)
{
    if (!h || !mapping) return USVMPC_E_ARG;
    *mapping = h->last_wide;
    return 0;
}

int usvmpc_last_kernel_ms(usvmpc_handle *h, float *linearize_ms, float *qp_ms)
{
    return usvmpc_kernel_ms(h, 1, linearize_ms, qp_ms);
}

int usvmpc_advance(usvmpc_handle *h, double sigma, unsigned long long seed)
{
    if (!h) return USVMPC_E_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    {
        const int rcf = mirror_flush(h);
        if (rcf) return rcf;
    }
    const long n = (long)h->B * h->nx;
    hipLaunchKernelGGL(usv_advance, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->ptrs, h->nx, sigma, seed, h->noise_mask);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int usvmpc_set_option(usvmpc_handle *h, const char *name, double value)
{
    if (!h) return USVMPC_E_ARG;
    const std::string s(name ? name : "");
    {
        const int rcs = spec_cancel(h); // (options change maps, layouts or launches: a lineariser that ran ahead is not trusted across them)
        if (rcs) return rcs;
    }
    if (s == "pipeline_linearize") { h->pipeline = value != 0.0; return 0; }
    if (s == "sort_by_difficulty") {
        h->sort_enabled = value != 0.0;
        if (!h->sort_enabled && h->ptrs.perm) { h->ptrs.perm = nullptr; h->map_changed = true; }
        return 0;
    }
    if (s == "max_waves") { h->max_waves = (long)value; reset_caps(h); cond_release(h); return 0; }
    if (s == "keep_multipliers") { // create the "lam" / "t" buffers now (a partially condensed solve fills them only if they exist)
        if (value == 0.0) return 0;
        DevPtrs &P = h->ptrs;
        if (!P.lam_out) {
            HIP_TRY(h, hipSetDevice(h->device));
            P.nlam = lam_len(h->spec, h->soft);
            const size_t cnt = (size_t)h->B * (h->N + 1) * (size_t)(P.nlam > 0 ? P.nlam : 1);
            if (dev_alloc(h, &P.lam_out, cnt, false)) return USVMPC_E_HIP;
            if (dev_alloc(h, &P.t_out, cnt, false)) { P.lam_out = nullptr; return USVMPC_E_HIP; }
            h->export_at = -1;
        }
        return 0;
    }
    if (s == "qp_cond_N") { // acados qp_solver_cond_N: stages of the partially condensed QP; 0 or N: no condensing (the default)
        const int n2 = (int)value;
        if (n2 < 0 || n2 > h->N) { h->err = "qp_cond_N must lie in 0..N"; return USVMPC_E_ARG; }
        const int want = (n2 == 0 || n2 == h->N) ? 0 : n2;
        if (want > 0) {
            // (blocks as HPIPM partitions them: N / N2 stages each, the first N mod N2 one more; one block's variables must fit a wave)
            if (h->nx + ((h->N + want - 1) / want) * h->nu > 64) { h->err = "qp_cond_N: a condensed stage may have at most 64 variables (nx + ceil(N / qp_cond_N) nu)"; return USVMPC_E_ARG; }
            if (h->spec.any_bsoft) { h->err = "partial condensing (qp_cond_N) is not built for soft state bounds"; return USVMPC_E_ARG; }
        }
        if (want != h->cond_N2) { cond_release(h); h->cond_N2 = want; h->map_changed = true; }
        return 0;
    }
    if (s == "sort_two_ticks") { h->sort_two = value != 0.0; return 0; }
    if (s == "aux_in_lds") { h->aux_lds = value != 0.0; reset_caps(h); return 0; }
    if (s == "lds_workspace") { // -1: when the batch is small (default), 0: never, 1: whenever an instance's planes fit in LDS
        h->lds_mode = value < 0.0 ? -1 : (value > 0.0 ? 1 : 0);
        reset_caps(h);
        return 0;
    }
    if (s == "wide") { // the latency mapping, one instance per wave: -1 for batches that leave SIMDs idle (default), 0 never, 1 whenever it applies
        h->wide_mode = value < 0.0 ? -1 : (value > 0.0 ? 1 : 0);
        reset_caps(h);
        return 0;
    }
    if (s == "wide_waves") { // waves per instance of the latency mapping: -1 (default) four for soft-row OCPs / two obstacle chunks up to one instance per CU, else one; 1; 4
        h->wide_waves = value < 0.0 ? -1 : (value >= 4.0 ? 4 : 1);
        reset_caps(h);
        return 0;
    }
    if (s == "dynamic_rows") { // 0: one group per row for the whole launch (the rows of a wave wait for its slowest)
        h->dynamic_rows = value != 0.0;
        reset_caps(h);
        return 0;
    }
    if (s == "cond_pred_corr" || s == "cpc_factor" || s == "hpipm_mode") {
        // HPIPM's conditional predictor-corrector (d_ocp_qp_ipm_arg.cond_pred_corr: on in every mode acados picks from - DESIGN.md section 2): a
        // corrected step that leaves the duality measure above cpc_factor (2) x the predictor's mu_aff is replaced by the centring-only step.
        // Changes the iteration path, i.e. results inside the exit tolerance ball.  Built into every kernel; the option switches the test.
        // "hpipm_mode": the profile's values of mu0, alpha_min and cond_pred_corr (host_spec.hpp, hpipm_profile).
        if (s == "cpc_factor") { if (!(value > 0.0)) { h->err = "cpc_factor must be positive"; return USVMPC_E_ARG; } h->spec.cpc_factor = value; }
        else if (s == "hpipm_mode") {
            usvmpc_desc d;
            if (value != (double)(int)value || !hpipm_profile(d, (int)value)) { h->err = "hpipm_mode must be one of USVMPC_HPIPM_*"; return USVMPC_E_ARG; }
            h->spec.mu0 = d.mu0; h->spec.alpha_min = d.alpha_min; h->spec.cpc = d.cond_pred_corr; h->spec.cpc_factor = d.cpc_factor;
        } else {
            h->spec.cpc = value != 0.0 ? 1 : 0;
        }
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpyAsync(h->d_spec, &h->spec, sizeof(DevSpec), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return 0;
    }
    if (s == "handover_iter") { // IPM iterations after which a row of a drained launch hands its instance over to the follow-up launch; 0: never
        if (value > 1e6) { h->err = "handover_iter out of range"; return USVMPC_E_ARG; }
        h->handover_iter = value < 0.0 ? -1 : (int)value;   // (-1: the default - launch_qp: past 20 for small batches when the follow-up works in LDS, else never)
        return 0;
    }
    if (s == "handover_lds") { h->handover_lds = value != 0.0; reset_caps(h); return 0; }
    if (s == "handover_co") { h->handover_co = value < 0.0 ? -1 : (value != 0.0 ? 1 : 0); return 0; } // the follow-up kernel beside the draining launch (default) or only behind it
    if (s == "handover_co_spin") { if (!(value >= 1.0 && value <= 2e9)) { h->err = "handover_co_spin out of range"; return USVMPC_E_ARG; } h->co_spin_limit = (int)value; return 0; }
    if (s == "handover_co_wgs") { if (!(value >= 0.0 && value <= 1e6)) { h->err = "handover_co_wgs out of range"; return USVMPC_E_ARG; } h->co_wgs = (long)value; return 0; } // the follow-up launch with the planes copied into LDS when the horizon fits (default), or always over the planes in HBM
    if (s == "disturbance_mask") { // bit j: usvmpc_advance adds its noise to state j
        h->noise_mask = (unsigned)value;
        return 0;
    }
    if (s == "merge_box_rows") { // 1 (default): box rows processed in their slot lanes when all of them ride there
        h->merge_rows = value != 0.0;   // (another set of kernels: merged / two-pass instantiations, wide ones included)
        reset_caps(h);
        return 0;
    }
    if (s == "host_mirror") { // 0: drop the pinned host mirror of the caller-visible arrays (every set / get then goes to the device)
        if (value != 0.0) { if (!h->mirror) { h->err = "host_mirror cannot be switched on again"; return USVMPC_E_ARG; } return 0; }
        if (!h->mirror) return 0;
        HIP_TRY(h, hipSetDevice(h->device));
        const int rcf = mirror_flush(h);
        if (rcf) return rcf;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->inflight = false; h->out_valid = false;
        (void)hipHostFree(h->mirror);
        h->mirror = nullptr;
        return 0;
    }
    if (s == "static_obstacles" || s == "pack_box_rows") {
        if (s == "static_obstacles") h->spec.p_static = value != 0.0;
        else { h->spec.boxpack = (value != 0.0 && h->spec.boxpack_ok) ? 1 : 0; reset_caps(h); h->layout_dirty = true; }
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpyAsync(h->d_spec, &h->spec, sizeof(DevSpec), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return 0;
    }
    h->err = "unknown option '" + s + "'";
    return USVMPC_E_FIELD;
}

// ---- guidance front end (M1 only): see guidance.hpp
static int guidance_alloc(usvmpc_handle *h, int npts)
{
    if (h->desc.model != USVMPC_MODEL_GUIDANCE_CA1) { h->err = "the guidance front end belongs to usv_model_guidance_ca1"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t B = h->B;
    GuidancePtrs &G = h->gd;
    if (!h->gd_ready) {
        double *d; int *i; float *f;
        if (dev_alloc(h, &d, B * 2, true)) return USVMPC_E_HIP; G.vel = d;
        if (dev_alloc(h, &d, B * 3, true)) return USVMPC_E_HIP; G.pose = d;
        if (dev_alloc(h, &d, B * GUIDANCE_LMAX * 3, true)) return USVMPC_E_HIP; G.obs = d;
        if (dev_alloc(h, &i, B, true)) return USVMPC_E_HIP; G.nobs = i;
        if (dev_alloc(h, &G.k, B, true)) return USVMPC_E_HIP;
        if (dev_alloc(h, &f, B, true)) return USVMPC_E_HIP; G.past_psied = f;
        if (dev_alloc(h, &G.ak, B, true)) return USVMPC_E_HIP;
        if (dev_alloc(h, &G.ye, B, true)) return USVMPC_E_HIP;
        if (dev_alloc(h, &G.active, B, true)) return USVMPC_E_HIP;
        if (dev_alloc(h, &G.heading, B, true)) return USVMPC_E_HIP;
        if (dev_alloc(h, &G.rdes, B, true)) return USVMPC_E_HIP;
        if (dev_alloc(h, &G.speed, B, true)) return USVMPC_E_HIP;
        if (dev_alloc(h, &h->gd_psi, B, true)) return USVMPC_E_HIP;
        G.lmax = GUIDANCE_LMAX;
        h->gd_ready = true;
    }
    if (npts > h->gd_npts_cap) { // grow: the old list is released, its slot in the allocation table with it
        dev_free(h, const_cast<double *>(G.wp), B * 2 * (size_t)h->gd_npts_cap * sizeof(double));
        double *d;
        if (dev_alloc(h, &d, B * 2 * (size_t)npts, true)) return USVMPC_E_HIP;
        G.wp = d;
        h->gd_npts_cap = npts;
    }
    return 0;
}

int usvmpc_guidance_reset(usvmpc_handle *h, const double *waypoints, int npts, const double *psi)
{
    if (!h || !waypoints || !psi || npts < 2) return USVMPC_E_ARG;
    int rc = guidance_alloc(h, npts);
    if (rc) return rc;
    GuidancePtrs &G = h->gd;
    G.npts = npts;
    const size_t B = h->B;
    HIP_TRY(h, hipMemcpyAsync(const_cast<double *>(G.wp), waypoints, B * 2 * npts * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->gd_psi, psi, B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(usv_guidance_reset, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, h->stream, G, h->gd_psi, (int)B);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!h->spec.p_static) return usvmpc_set_option(h, "static_obstacles", 1.0);
    return 0;
}

int usvmpc_guidance_prepare(usvmpc_handle *h, const double *vel_uv, const double *pose, const double *obstacles,
                            const int *n_obstacles, int lmax)
{
    // n_obstacles == NULL: keep the body-frame lists usvmpc_guidance_sense left on the device
    if (!h || !vel_uv || !pose || lmax < 0 || lmax > GUIDANCE_LMAX) return USVMPC_E_ARG;
    if (!h->gd_ready || h->gd.npts < 2) { h->err = "usvmpc_guidance_reset must be called first"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    {
        const int rcf = mirror_flush(h);
        if (rcf) return rcf;
    }
    GuidancePtrs &G = h->gd;
    const size_t B = h->B;
    HIP_TRY(h, hipMemcpyAsync(const_cast<double *>(G.vel), vel_uv, B * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(const_cast<double *>(G.pose), pose, B * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (n_obstacles)
        HIP_TRY(h, hipMemcpyAsync(const_cast<int *>(G.nobs), n_obstacles, B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (n_obstacles && lmax > 0) {
        if (!obstacles) return USVMPC_E_ARG;
        // [B][lmax][3] -> [B][GUIDANCE_LMAX][3]
        HIP_TRY(h, hipMemcpy2DAsync(const_cast<double *>(G.obs), GUIDANCE_LMAX * 3 * sizeof(double), obstacles,
                                    (size_t)lmax * 3 * sizeof(double), (size_t)lmax * 3 * sizeof(double), B,
                                    hipMemcpyHostToDevice, h->stream));
    }
    hipLaunchKernelGGL(usv_guidance_pre, dim3((unsigned)((B + 127) / 128)), dim3(128), 0, h->stream, h->ptrs, G);
    HIP_TRY(h, hipGetLastError());
    // the copies above read caller memory: it must be reusable when this call returns (pinned buffers make
    // hipMemcpyAsync truly asynchronous)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return 0;
}

int usvmpc_guidance_sense(usvmpc_handle *h, const double *pose, const double *world, int n_world, double max_radius,
                          double *obstacles, int *n_obstacles)
{
    if (!h || !pose || n_world < 0 || (n_world > 0 && !world)) return USVMPC_E_ARG;
    if (!h->gd_ready) { h->err = "usvmpc_guidance_reset must be called first"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    GuidancePtrs &G = h->gd;
    const size_t B = h->B;
    if ((size_t)n_world > h->gd_world_cap) {
        dev_free(h, h->gd_world, B * h->gd_world_cap * 3 * sizeof(double));
        h->gd_world = nullptr;
        if (dev_alloc(h, &h->gd_world, B * (size_t)n_world * 3, true)) return USVMPC_E_HIP;
        h->gd_world_cap = (size_t)n_world;
    }
    HIP_TRY(h, hipMemcpyAsync(const_cast<double *>(G.pose), pose, B * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (n_world > 0)
        HIP_TRY(h, hipMemcpyAsync(h->gd_world, world, B * (size_t)n_world * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(usv_obstacle_sim, dim3((unsigned)((B + 127) / 128)), dim3(128), 0, h->stream, G, h->gd_world, n_world,
                       max_radius, (int)B);
    HIP_TRY(h, hipGetLastError());
    if (obstacles)
        HIP_TRY(h, hipMemcpyAsync(obstacles, G.obs, B * GUIDANCE_LMAX * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (n_obstacles)
        HIP_TRY(h, hipMemcpyAsync(n_obstacles, G.nobs, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (obstacles || n_obstacles) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return 0;
}

int usvmpc_guidance_publish(usvmpc_handle *h, double *heading, double *r_des, double *speed, double *ye, int *active)
{
    if (!h) return USVMPC_E_ARG;
    if (!h->gd_ready) { h->err = "usvmpc_guidance_reset must be called first"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    GuidancePtrs &G = h->gd;
    const size_t B = h->B;
    hipLaunchKernelGGL(usv_guidance_post, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, h->stream, h->ptrs, G);
    HIP_TRY(h, hipGetLastError());
    if (heading) HIP_TRY(h, hipMemcpyAsync(heading, G.heading, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (r_des) HIP_TRY(h, hipMemcpyAsync(r_des, G.rdes, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (speed) HIP_TRY(h, hipMemcpyAsync(speed, G.speed, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (ye) HIP_TRY(h, hipMemcpyAsync(ye, G.ye, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (active) HIP_TRY(h, hipMemcpyAsync(active, G.active, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return 0;
}

int usvmpc_guidance_state(usvmpc_handle *h, int *wp_index, float *past_psied)
{
    if (!h) return USVMPC_E_ARG;
    if (!h->gd_ready) { h->err = "usvmpc_guidance_reset must be called first"; return USVMPC_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t B = h->B;
    if (wp_index) HIP_TRY(h, hipMemcpyAsync(wp_index, h->gd.k, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (past_psied) HIP_TRY(h, hipMemcpyAsync(past_psied, h->gd.past_psied, B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return 0;
}

int usvmpc_calibrate_traffic(usvmpc_handle *h, int nplanes, double *bytes_read, double *bytes_written)
{
    if (!h || nplanes < 1) return USVMPC_E_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    {
        const int rcs = spec_cancel(h);
        if (rcs) return rcs;
    }
    const long stride = (long)h->Bp * LANES;
    const long avail = (long)(h->N + 1) * ws_planes(h->nx, h->nu, h->kch, h->soft, model_mat_planes(h->desc.model), h->spec.any_bsoft != 0);
    if (nplanes + 1 > avail) { h->err = "nplanes exceeds the workspace"; return USVMPC_E_ARG; }
    if ((size_t)h->Bp * (size_t)(nplanes + 1) * 128u > (size_t)UINT32_MAX) { // (usv_calib_stream addresses it all through one 32-bit window)
        h->err = "nplanes * batch exceeds the 4 GiB buffer window of the calibration kernel";
        return USVMPC_E_ARG;
    }
    const long groups = h->Bp;
    hipLaunchKernelGGL(usv_calib_stream, dim3((unsigned)((groups * LANES + 63) / 64)), dim3(64), 0, h->stream, h->ptrs, groups, nplanes);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (bytes_read) *bytes_read = (double)nplanes * stride * 8.0;
    if (bytes_written) *bytes_written = (double)stride * 8.0;
    return 0;
}

int usvmpc_set_stream(usvmpc_handle *h, void *stream)
{
    if (!h) return USVMPC_E_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    {
        const int rcf = mirror_flush(h);
        if (rcf) return rcf;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->inflight = false;
    {
        const int rcs = spec_cancel(h);
        if (rcs) return rcs;
        h->pipeline = false; // (the caller's stream carries the caller's own ordering: nothing of this handle runs beside it)
        h->handover_co = 0;
    }
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return 0;
}

size_t usvmpc_device_bytes(usvmpc_handle *h) { return h ? h->bytes : 0; }

const char *usvmpc_last_error(usvmpc_handle *h) { return h ? h->err.c_str() : "null handle"; }

} // extern "C"
#endif // USV_MAIN

#ifndef USV_COND_SEPARATE // (one translation unit by default - generated-model and development builds; the shipped library compiles it on its own)
#include "cond_kernels.hip"
#endif
