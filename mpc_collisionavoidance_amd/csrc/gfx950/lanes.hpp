// lanes.hpp (gfx950) — cross-lane primitives for the "16 lanes per OCP instance" layout.
//
// One OCP instance is owned by one DPP row (16 consecutive lanes of a wave64), so a wave carries
// four instances.  Lane r of a group owns row r of every small matrix (variable ordering [u;x])
// and entry r of every vector; other lanes' entries are fetched with DPP `row_newbcast`
// (broadcast lane K of each row to the whole row), reductions use DPP `row_ror` butterflies.
// No LDS, no shuffles through memory, no MFMA.
//
// Control flow around these primitives must be wave-uniform: a DPP read from an EXEC-disabled
// lane is undefined, so kernels freeze finished instances with selects instead of branching.
#pragma once
#include <hip/hip_runtime.h>

#define USV_DEV __device__ __forceinline__

namespace lanes {

constexpr int GROUP = 16;

USV_DEV int lane() { return (int)(threadIdx.x & 15u); }
// global index of this 16-lane group (one group = one OCP instance or one (instance,stage) pair)
USV_DEV long group_linear() { return (long)blockIdx.x * (long)(blockDim.x >> 4) + (long)(threadIdx.x >> 4); }

template <int CTRL>
USV_DEV double dpp_mov(double v)
{
    // bound_ctrl:1 - every lane of the row is written and no source lane is ever masked off (control
    // flow around these ops is wave-uniform), so there is no `old` value to preserve.
    // 64-bit operand: row_newbcast is a legal DP-ALU DPP control on gfx950 and becomes ONE v_mov_b64_dpp;
    // the rotations are not and are split by the compiler into two v_mov_b32_dpp.
    const long l = __builtin_bit_cast(long, v);
    const long r = __builtin_amdgcn_update_dpp(0L, l, CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, r);
}

// value held by lane K of this group, delivered to all 16 lanes (DPP row_newbcast:K)
template <int K>
USV_DEV double bcast(double v)
{
    static_assert(K >= 0 && K < 16, "lane index");
    return dpp_mov<0x150 + K>(v);
}

// the same for a 32-bit integer
template <int K>
USV_DEV int bcast_i(int v)
{
    static_assert(K >= 0 && K < 16, "lane index");
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, true);
}

// USV_DPP_PAD (generated-model libraries only: genbuild.py defines it when the kernels of a user's model trip the hazard check): two
// wait states in front of every group - the hazard cannot occur whatever the compiler schedules around the asm; same arithmetic, a
// few per cent slower.  The stock library is never built with it: a hazard there is fixed at its site (lanes::settle).
#ifdef USV_DPP_PAD
#define USV_DPP_HEAD "s_nop 1\n\t"
#else
#define USV_DPP_HEAD ""
#endif
// c += bcast<K>(b_remote) * a_own
template <int K>
USV_DEV void fma_bc(double &c, double b_remote, double a_own)
{
    // v_fmac_f64_dpp with row_newbcast is the one DP-ALU DPP form gfx950 has and costs what a plain v_fma_f64 costs
    // (profiles/r02_microbench.txt), but hipcc does not select it from builtins.  Inside asm the compiler cannot pad the
    // one hazard it has (a VALU write of the DPP source b_remote within the 2 preceding wait states - measured on the
    // part: tools/micro/dpp_hazard.hip; accumulator and plain source are forwarded normally), so the BINARY is checked
    // instead: tools/check_dpp_hazard.py walks the disassembly and build() fails on a violation; lanes::settle() is the
    // cure at a site it flags.
    asm(USV_DPP_HEAD "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(b_remote), "v"(a_own), "n"(K));
}

// The same for up to four terms into one accumulator, in the order given: c += bcast<K0>(b0) * a0; c += bcast<K1>(b1) * a1; ...
// One asm statement per group: between two asm statements that touch the same register the compiler pads a wait state
// (s_nop 0) it cannot prove unnecessary - a chain of single-term statements was one fifth s_nop (measured on the part,
// tools/micro/dpp_hazard.hip: dependent v_fmac_f64_dpp issue back to back correctly).  The accumulator is early-clobber:
// it is written while later sources are still to be read.
template <int K0, int K1>
USV_DEV void fma_bc2(double &c, double b0, double a0, double b1, double a1)
{
    asm(USV_DPP_HEAD "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %3, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
        : "+&v"(c) : "v"(b0), "v"(a0), "v"(b1), "v"(a1), "n"(K0), "n"(K1));
}
template <int K0, int K1, int K2>
USV_DEV void fma_bc3(double &c, double b0, double a0, double b1, double a1, double b2, double a2)
{
    asm(USV_DPP_HEAD "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %3, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %5, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+&v"(c) : "v"(b0), "v"(a0), "v"(b1), "v"(a1), "v"(b2), "v"(a2), "n"(K0), "n"(K1), "n"(K2));
}
template <int K0, int K1, int K2, int K3>
USV_DEV void fma_bc4(double &c, double b0, double a0, double b1, double a1, double b2, double a2, double b3, double a3)
{
    asm(USV_DPP_HEAD "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %3, %4 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %5, %6 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %7, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
        : "+&v"(c) : "v"(b0), "v"(a0), "v"(b1), "v"(a1), "v"(b2), "v"(a2), "v"(b3), "v"(a3), "n"(K0), "n"(K1), "n"(K2), "n"(K3));
}

// two wait states on a value that is about to be a DPP source (see fma_bc)
USV_DEV void settle(double &v)
{
    asm volatile("s_nop 1" : "+v"(v));
}

// value held by lane `src` (0..15, run-time, may differ per lane) of this group: ds_bpermute_b32 goes
// through the LDS crossbar but touches no LDS memory
USV_DEV double gather(double v, int src)
{
    const int addr = (int)((((unsigned)threadIdx.x & 48u) | ((unsigned)src & 15u)) << 2);
    const long l = __builtin_bit_cast(long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)l);
    const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(l >> 32));
    return __builtin_bit_cast(double, ((long)(unsigned)lo) | ((long)hi << 32));
}

// rotate right by N lanes within the group (DPP row_ror:N)
template <int N>
USV_DEV double ror(double v)
{
    static_assert(N >= 1 && N < 16, "rotation");
    return dpp_mov<0x120 + N>(v);
}

// all-reduce over the 16 lanes of the group; every lane receives bitwise the same result
USV_DEV double gsum(double v)
{
    v += ror<8>(v);
    v += ror<4>(v);
    v += ror<2>(v);
    v += ror<1>(v);
    return v;
}
// max(a, b), max(a, |b|), max(|a|, |b|) as ONE v_max_f64 each.  fmax() costs two where the compiler cannot prove an operand
// canonical (anything that went through a select or a load): it puts a v_max_f64 x, x, x in front - half of the kernel's
// v_max_f64 were those.  The instruction itself returns the other operand for a NaN one, as fmax does.
USV_DEV double vmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
USV_DEV double vmax_abs(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
USV_DEV double vmax_abs2(double a, double b)
{
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
USV_DEV double gmax(double v)
{
    v = vmax(v, ror<8>(v));
    v = vmax(v, ror<4>(v));
    v = vmax(v, ror<2>(v));
    v = vmax(v, ror<1>(v));
    return v;
}
USV_DEV double gmin(double v)
{
    v = fmin(v, ror<8>(v));
    v = fmin(v, ror<4>(v));
    v = fmin(v, ror<2>(v));
    v = fmin(v, ror<1>(v));
    return v;
}

// A value that is the same in every lane of the wave, moved to a scalar register.  Values read from global
// memory arrive in vector registers and the compiler then treats every test on them as divergent (EXEC masking,
// vector compares, conservative waits) although the control flow is wave-uniform by construction.
USV_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// no instruction may be scheduled across this point
USV_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// one more in a global counter
USV_DEV void count_one(int *p) { atomicAdd(p, 1); }
// ... returning the value before (work-queue head)
USV_DEV int fetch_add(int *p) { return atomicAdd(p, 1); }

// true if the predicate holds in any lane of the wave (four instances)
USV_DEV bool wave_any(bool p) { return __any((int)p) != 0; }

// 1/x and 1/sqrt(x) from the hardware estimate + two Newton steps (full FP64 accuracy for the normal-range operands of the IPM;
// ~10 instructions instead of the ~25 of an IEEE division).  The arithmetic of the shipped kernels is THIS, with no build-time
// variant: the IEEE-division build of the parity experiment (profiles/r03_parity_tail.txt) is a patch, tools/experiments/exact_div.patch.
USV_DEV double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
// 1/a and 1/b from ONE reciprocal: r = 1/(a b), 1/a = r b, 1/b = r a.  v_rcp_f64 is quarter rate and the Newton steps are four
// FMAs, so a pair costs 8 instructions instead of 10 and one slow one instead of two.  For operands whose product stays inside
// the double range (slacks and multipliers of the IPM: the C ABI refuses bounds beyond 1e100).
USV_DEV void frcp2(double a, double b, double &ia, double &ib)
{
    const double r = frcp(a * b);
    ia = r * b;
    ib = r * a;
}
USV_DEV double frsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
    y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
    return y;
}

// Hand-over of an instance's results to a kernel that runs CONCURRENTLY on another stream (the next tick's lineariser inside this
// launch's tail, usvmpc.hip): the XCDs' L2s are not coherent with each other, so the payload goes out and comes in with accesses
// that bypass them (agent-scope relaxed atomics = sc1), and the flag is stored after the wave's stores have drained
// (MI355X_MICROARCH.md: "sc1 payload -> vmcnt(0) -> sc1 flag").  The consumer's payload loads are control-dependent on the flag.
USV_DEV void st_shared(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
USV_DEV double ld_shared(const double *p)
{
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
USV_DEV void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
USV_DEV void publish(int *flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
USV_DEV int observe(const int *flag) { return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
USV_DEV void set_bits(int *word, int bits) { atomicOr(word, bits); }
// Plain stores of this wave handed to a wave on another CU / XCD while both kernels run (hand-over to the co-resident follow-up kernel): the
// producer writes its XCD's L2 back before the flag goes out (the asm wait is the one the compiler must not drop: MI355X_MICROARCH.md,
// "Compiler hazard"), the consumer invalidates its L1 / non-local L2 lines after it has seen the flag.
USV_DEV void release_agent() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
USV_DEV void acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
USV_DEV bool claim(int *entry, int seen) { return atomicCAS(entry, seen, -2 - seen) == seen; }

// lane index inside the wave (a wave carries four 16-lane groups)
USV_DEV unsigned wave_lane() { return threadIdx.x & 63u; }
// rows (16-lane groups) of a wave and this lane's row
constexpr int WAVE_ROWS = 4;
USV_DEV unsigned wave_row() { return (threadIdx.x >> 4) & 3u; }
// ---- the WIDE mapping of qp_ipm.hpp (ONE instance per wave: its four rows hold the same values and share out the stage-local row
// work): what crosses rows.  xrow_*: all-reduce over the four lanes that sit at the same position of the four rows (ds_bpermute
// through the LDS crossbar, no LDS memory); wave_first_i: lane 0's value in every lane; lds_fence: rows hand values to each other
// through LDS - the operations of one wave execute in order, the fence only keeps the compiler from moving reads above writes.
USV_DEV double xrow_shfl(double v, unsigned mask)
{
    const int addr = (int)((((unsigned)threadIdx.x & 63u) ^ mask) << 2);
    const long l = __builtin_bit_cast(long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)l);
    const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(l >> 32));
    return __builtin_bit_cast(double, ((long)(unsigned)lo) | ((long)hi << 32));
}
USV_DEV double xrow_max(double v) { v = vmax(v, xrow_shfl(v, 16u)); v = vmax(v, xrow_shfl(v, 32u)); return v; }
USV_DEV double xrow_sum(double v) { v += xrow_shfl(v, 16u); v += xrow_shfl(v, 32u); return v; }
USV_DEV int wave_first_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// (several waves per instance: qp_ipm.hpp WW) this lane's row among the rows of the workgroup; workgroup barrier
USV_DEV unsigned block_row() { return threadIdx.x >> 4; }
USV_DEV void block_sync() { __syncthreads(); }
USV_DEV void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
// the workgroup's dynamic LDS (one wave per workgroup in the QP kernel): the planes of PlanesLds, or the aux area of qp_ipm.hpp
USV_DEV double *dyn_lds()
{
    extern __shared__ double usv_lds[];
    return usv_lds;
}

// The lane-major planes of ONE stage in HBM: [group][plane][16 lanes] - a group's (= OCP instance's) planes of a stage
// are 128-byte rows back to back, so every group streams its own contiguous block whatever groups share its wave.
// Addressed through a buffer resource: the descriptor (the stage's window over all groups) lives in SGPRs, the
// group + lane offset in ONE VGPR that only changes when a row takes up another group, and the plane offset
// (plane * 128) is a compile-time constant of the instruction, so a plane access costs neither VALU address
// arithmetic nor a live scalar register (buffer_load_dwordx2 ... offen).  base must be wave-uniform.
// Cache policy of the plane accesses (aux operand of the buffer instructions: 0 default, 2 = nt, non-temporal).
// A stored plane is not read again before tens of gigabytes have passed: non-temporal STORES measure -3 % (M2) /
// -5 % (M1) on the QP kernel; non-temporal loads +-0, both together +6 % (tools/micro/store_policy.hip has the
// isolated streams).
constexpr int PLANE_LOAD_AUX = 0, PLANE_STORE_AUX = 2;
struct Planes {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff; // byte offset of this lane's entry of plane 0 inside the stage window

    // byte offset of (group, lane) in a stage window whose groups hold nplanes planes each
    USV_DEV static unsigned lane_offset(long group, int nplanes, int lane) { return (unsigned)(group * nplanes * 128 + lane * 8); }
    USV_DEV Planes(const double *base, unsigned nbytes, unsigned voff_) : voff(voff_)
    {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, (int)nbytes, 0x00020000);
    }
    USV_DEV double ld(int plane) const
    {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        // (plane offset added to the VGPR offset: the compiler folds the constant into the instruction's 12-bit offset field;
        // passed as the scalar offset operand it costs an s_movk per access)
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(voff + (unsigned)(plane * 128)), 0, PLANE_LOAD_AUX);
        return __builtin_bit_cast(double, v);
    }
    USV_DEV void st(int plane, double x) const
    {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, x), rsrc, (int)(voff + (unsigned)(plane * 128)), 0, PLANE_STORE_AUX);
    }
};

// Wave-private exchange area in LDS: every row (16 lanes = one OCP instance) has NENT double slots, laid out
// [slot][4 rows] so that the 64 lanes of a "lane L writes slot 16 q + L" store cover 512 contiguous bytes.  A lane puts
// values into slots and any lane of the same row reads any slot back: the run-time lane permutation that would otherwise
// cost two ds_bpermute plus address arithmetic and selects PER VALUE becomes one ds_read_b64 at a per-lane slot number -
// and slots holding constants (0.0, 1.0) replace the selects for entries that are structurally zero / one.  This is how
// the packed stage matrix is turned into the row (backward sweeps) or column (forward sweeps) form the products need.
// LDS operations of one wave execute in order; sync() only keeps the compiler from moving reads above the writes.
// PERWAVE (the WIDE mapping: the four rows of a wave hold the same values): one copy per wave, ROWS = waves of the workgroup - the
// waves of a workgroup drift apart between its barriers, so they must not share one.
template <int NENT, int ROWS = 4, bool PERWAVE = false>
struct Xpose {
    USV_DEV static double *area()
    {
        __shared__ double s[NENT * ROWS];
        return s;
    }
    USV_DEV static unsigned row() { return PERWAVE ? (ROWS == 1 ? 0u : (threadIdx.x >> 6)) : ((threadIdx.x >> 4) & 3u); }
    USV_DEV static void put(int slot, double v) { area()[slot * ROWS + row()] = v; }
    USV_DEV static double get(int slot) { return area()[slot * ROWS + row()]; }
    USV_DEV static void sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
};

// Per-lane constants of a wave parked in LDS ([slot][64 lanes], wave-private): values that live for the whole kernel and are
// read once or twice per stage.  In VGPRs they are what the register allocator spills first - and a spill reload is a
// scratch (vector-memory) load, whose s_waitcnt vmcnt(0) waits for every plane load issued before it: the reload in the
// middle of a stage drains the prefetch queue (this was the largest single stall of the QP kernel).
// An LDS read waits on lgkmcnt only.
// WLANES = 16 (the WIDE mapping: the four rows hold the same values): one row's worth for the wave.
template <int NSLOT, int WLANES = 64>
struct Stash {
    USV_DEV static double *area()
    {
        __shared__ double s[NSLOT * WLANES];
        return s;
    }
    USV_DEV static void put(int slot, double v) { area()[slot * WLANES + (threadIdx.x & (unsigned)(WLANES - 1))] = v; }
    USV_DEV static double get(int slot) { return area()[slot * WLANES + (threadIdx.x & (unsigned)(WLANES - 1))]; }
    // two small values sharing one slot (halves 0 / 1, kept as floats: exact for flags and small integers)
    USV_DEV static void puth(int slot, int half, double v) { ((float *)area())[2 * (slot * WLANES + (threadIdx.x & (unsigned)(WLANES - 1))) + half] = (float)v; }
    USV_DEV static double geth(int slot, int half) { return (double)((const float *)area())[2 * (slot * WLANES + (threadIdx.x & (unsigned)(WLANES - 1))) + half]; }
};

// The same planes held in the CU's LDS instead of HBM (small batches: the whole horizon of an instance's planes fits in
// the 160 KB of a CU, so no sweep waits for HBM).  Layout inside a workgroup's LDS: [row][stage][plane][16 lanes]; `off` is
// this lane's entry of plane 0 of the stage, in doubles; rows without an instance of their own never store (live).
struct PlanesLds {
    unsigned off;
    bool live;
    USV_DEV PlanesLds(unsigned off_, bool live_) : off(off_), live(live_) {}
    USV_DEV double ld(int plane) const
    {
        extern __shared__ double usv_lds[];
        return usv_lds[off + plane * 16];
    }
    USV_DEV void st(int plane, double x) const
    {
        extern __shared__ double usv_lds[];
        if (live) usv_lds[off + plane * 16] = x;
    }
};

// The planes of the WIDE mapping in LDS: only those the solve writes live there (MAP::at(plane): position among them; the unused box
// planes of the packed layouts are squeezed out, the lineariser's planes - read once per sweep - stay in HBM / L2).
template <class MAP>
struct PlanesLdsMapped {
    unsigned off;
    bool live;
    USV_DEV PlanesLdsMapped(unsigned off_, bool live_) : off(off_), live(live_) {}
    USV_DEV double ld(int plane) const
    {
        extern __shared__ double usv_lds[];
        return usv_lds[off + MAP::at(plane) * 16];
    }
    USV_DEV void st(int plane, double x) const
    {
        extern __shared__ double usv_lds[];
        if (live) usv_lds[off + MAP::at(plane) * 16] = x;
    }
};

} // namespace lanes
