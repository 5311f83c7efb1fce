// lanes.hpp (CPU emulation, TEST-ONLY) — same interface as
// mpc_collisionavoidance_amd/csrc/gfx950/lanes.hpp, implemented with 16 cooperative fibers so
// that the *unmodified* kernel bodies (linearize.hpp, qp_ipm.hpp) can be executed and debugged
// on the CPU.  Selected by include path (-Itests/emu before csrc/gfx950); never part of the
// shipped library.  Every cross-lane primitive is a rendezvous of all 16 lanes: a lane that
// skips one (divergent control flow around a DPP op — undefined on the GPU) is detected.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define USV_DEV inline __attribute__((always_inline))

namespace lanes {

constexpr int GROUP = 16;
constexpr int MAXF = 256;  // fibers of a whole emulated workgroup (the WIDE mapping of qp_ipm.hpp: 4 .. 16 rows in lock step)

struct Emu {
    int cur = 0;           // fiber currently running: row * 16 + lane
    int rows = 1;          // rows emulated side by side (1: one 16-lane group; 4: a whole wave)
    long group = 0;        // group index handed to the body
    double slot[2][MAXF];  // exchange buffers, double-buffered by parity
    int par[MAXF];         // per-lane parity
    long nops[MAXF];       // per-lane exchange counter (divergence check)
    bool finished[MAXF];
    void *sp[MAXF];        // saved stack pointers of the fibers
    void *main_sp = nullptr;
};
extern Emu g_emu;
extern "C" void usv_emu_switch(void **save_sp, void *load_sp);

inline int lane() { return g_emu.cur & 15; }
inline long group_linear() { return g_emu.group; }

// publish v, wait until all lanes have published, return the value published by fiber src (of the whole emulated wave)
inline double exchange_wave(double v, int src)
{
    Emu &e = g_emu;
    const int me = e.cur;
    const int p = e.par[me];
    e.slot[p][me] = v;
    e.nops[me]++;
    usv_emu_switch(&e.sp[me], e.main_sp); // yield to the scheduler
    e.par[me] = p ^ 1;
    return e.slot[p][src];
}
// ... by lane src of the caller's own row
inline double exchange(double v, int src) { return exchange_wave(v, (g_emu.cur & ~15) | (src & 15)); }

template <int K>
inline double bcast(double v) { return exchange(v, K); }

template <int K>
inline int bcast_i(int v) { return (int)exchange((double)v, K); }

template <int K>
inline void fma_bc(double &c, double b_remote, double a_own) { c = std::fma(bcast<K>(b_remote), a_own, c); }
template <int K0, int K1>
inline void fma_bc2(double &c, double b0, double a0, double b1, double a1) { fma_bc<K0>(c, b0, a0); fma_bc<K1>(c, b1, a1); }
template <int K0, int K1, int K2>
inline void fma_bc3(double &c, double b0, double a0, double b1, double a1, double b2, double a2) { fma_bc2<K0, K1>(c, b0, a0, b1, a1); fma_bc<K2>(c, b2, a2); }
template <int K0, int K1, int K2, int K3>
inline void fma_bc4(double &c, double b0, double a0, double b1, double a1, double b2, double a2, double b3, double a3) { fma_bc2<K0, K1>(c, b0, a0, b1, a1); fma_bc2<K2, K3>(c, b2, a2, b3, a3); }
inline void settle(double &) {} // hazard padding of the device build: nothing to emulate

inline double gather(double v, int src) { return exchange(v, src); }

template <int N>
inline double ror(double v) { return exchange(v, (lane() - N) & 15); }

inline double gsum(double v)
{
    v += ror<8>(v);
    v += ror<4>(v);
    v += ror<2>(v);
    v += ror<1>(v);
    return v;
}
inline double vmax(double a, double b) { return std::fmax(a, b); }
inline double vmax_abs(double a, double b) { return std::fmax(a, std::fabs(b)); }
inline double vmax_abs2(double a, double b) { return std::fmax(std::fabs(a), std::fabs(b)); }
inline double gmax(double v)
{
    v = std::fmax(v, ror<8>(v));
    v = std::fmax(v, ror<4>(v));
    v = std::fmax(v, ror<2>(v));
    v = std::fmax(v, ror<1>(v));
    return v;
}
inline double gmin(double v)
{
    v = std::fmin(v, ror<8>(v));
    v = std::fmin(v, ror<4>(v));
    v = std::fmin(v, ror<2>(v));
    v = std::fmin(v, ror<1>(v));
    return v;
}

// the emulator runs one group at a time, so "any lane of the wave" is "any lane of the group";
// finished instances are frozen by the kernels, so results do not depend on the grouping
inline bool wave_any(bool p) { return gmax(p ? 1.0 : 0.0) > 0.5; }

inline int uniform(int v) { return v; }
inline void sched_fence() {}
inline void count_one(int *p) { ++*p; }
inline int fetch_add(int *p) { return (*p)++; }

inline double frcp(double x) { return 1.0 / x; }
inline void frcp2(double a, double b, double &ia, double &ib) { const double r = 1.0 / (a * b); ia = r * b; ib = r * a; }
inline double frsqrt(double x) { return 1.0 / std::sqrt(x); }

// hand-over primitives (gfx950/lanes.hpp): plain memory on the CPU
inline void st_shared(double *p, double v) { *p = v; }
inline double ld_shared(const double *p) { return *p; }
inline void drain_stores() {}
inline void publish(int *flag, int v) { *flag = v; }
inline int observe(const int *flag) { return *flag; }
inline void set_bits(int *word, int bits) { *word |= bits; }
inline void release_agent() {}
inline void acquire_agent() {}
inline bool claim(int *entry, int seen) { if (*entry != seen) return false; *entry = -2 - seen; return true; }

// lane index inside the (emulated) wave: the group's quarter of its 4-group tile (a whole emulated wave: the fiber number)
inline unsigned wave_lane() { return g_emu.rows > 1 ? (unsigned)(g_emu.cur & 63) : (unsigned)((g_emu.group & 3) * 16 + g_emu.cur); }
// across the rows of a whole emulated wave (gfx950/lanes.hpp): rendezvous of all its fibers
inline double xrow_shfl(double v, unsigned mask) { return exchange_wave(v, (int)((unsigned)g_emu.cur ^ mask)); }
inline double xrow_max(double v)
{
    if (g_emu.rows < 4) return v;
    v = std::fmax(v, xrow_shfl(v, 16u)); v = std::fmax(v, xrow_shfl(v, 32u));
    return v;
}
inline double xrow_sum(double v)
{
    if (g_emu.rows < 4) return v;
    v += xrow_shfl(v, 16u); v += xrow_shfl(v, 32u);
    return v;
}
inline int wave_first_i(int v) { return (int)exchange_wave((double)v, g_emu.cur & ~63); }
inline void lds_fence() { (void)exchange(0.0, 0); }
inline unsigned block_row() { return (unsigned)(g_emu.cur >> 4); }
inline void block_sync() { (void)exchange(0.0, 0); }

// the planes of one stage [group][plane][16] (plain pointers here; a buffer resource on the GPU)
struct Planes {
    double *base;
    unsigned nbytes, voff;
    static unsigned lane_offset(long group, int nplanes, int lane) { return (unsigned)(group * nplanes * 128 + lane * 8); }
    Planes(const double *b, unsigned n, unsigned v) : base(const_cast<double *>(b)), nbytes(n), voff(v) {}
    // (a row that has nothing more to do parks its offset at the end of the window: the GPU's buffer range check returns 0 for its
    // loads and drops its stores - qp_ipm.hpp QpIpm::solve; any OTHER access outside the window is a bug)
    double ld(int plane) const
    {
        const unsigned o = voff + (unsigned)plane * 128u;
        if (voff >= nbytes) return 0.0;
        if (plane < 0 || o + 8 > nbytes) { std::fprintf(stderr, "Planes::ld out of window (plane %d)\n", plane); std::abort(); }
        return base[o / 8];
    }
    template <int AUX>
    double ld_policy(int plane) const { return ld(plane); }
    void st(int plane, double x) const
    {
        const unsigned o = voff + (unsigned)plane * 128u;
        if (voff >= nbytes) return;
        if (plane < 0 || o + 8 > nbytes) { std::fprintf(stderr, "Planes::st out of window (plane %d)\n", plane); std::abort(); }
        base[o / 8] = x;
    }
};

// wave-private exchange area (gfx950/lanes.hpp): one emulated row, sync() is a rendezvous of its 16 lanes
template <int NENT, int ROWS = 4, bool PERWAVE = false>
struct Xpose {
    static double *area() { static double s[4][NENT]; return s[PERWAVE ? (g_emu.cur >> 6) : ((g_emu.cur >> 4) & 3)]; }
    static void put(int slot, double v) { area()[slot] = v; }
    static double get(int slot) { return area()[slot]; }
    static void sync() { (void)exchange(0.0, 0); }
};

// per-lane constants parked in LDS (gfx950/lanes.hpp): one emulated row
template <int NSLOT, int WLANES = 64>
struct Stash {
    static double *area() { static double s[4][NSLOT * 16]; return s[WLANES == 16 ? 0 : ((g_emu.cur >> 4) & 3)]; }
    static void put(int slot, double v) { area()[slot * 16 + lane()] = v; }
    static double get(int slot) { return area()[slot * 16 + lane()]; }
    static void puth(int slot, int half, double v) { ((float *)area())[2 * (slot * 16 + lane()) + half] = (float)v; }
    static double geth(int slot, int half) { return (double)((const float *)area())[2 * (slot * 16 + lane()) + half]; }
};

// the workgroup's LDS (one emulated row per "wave")
extern double *g_emu_lds;
constexpr int WAVE_ROWS = 1;
inline unsigned wave_row() { return (unsigned)((g_emu.cur >> 4) & 3); }
inline double *dyn_lds() { return g_emu_lds; }
struct PlanesLds {
    unsigned off;
    bool live;
    PlanesLds(unsigned o, bool l) : off(o), live(l) {}
    double ld(int plane) const { return g_emu_lds[off + plane * 16]; }
    void st(int plane, double x) const { if (live) g_emu_lds[off + plane * 16] = x; }
};

template <class MAP>
struct PlanesLdsMapped {
    unsigned off;
    bool live;
    PlanesLdsMapped(unsigned o, bool l) : off(o), live(l) {}
    double ld(int plane) const
    {
        if (MAP::at(plane) < 0) { std::fprintf(stderr, "PlanesLdsMapped::ld of a plane that is not kept in LDS (%d)\n", plane); std::abort(); }
        return g_emu_lds[off + MAP::at(plane) * 16];
    }
    void st(int plane, double x) const
    {
        if (MAP::at(plane) < 0) { std::fprintf(stderr, "PlanesLdsMapped::st of a plane that is not kept in LDS (%d)\n", plane); std::abort(); }
        if (live) g_emu_lds[off + MAP::at(plane) * 16] = x;
    }
};

// run body(lane) on 16 fibers in lock step (rows = 4: on the 64 fibers of a whole wave, every row handed the same group)
void run_group(long group, void (*body)(void *), void *arg, int rows = 1);

} // namespace lanes

// device math names used by the kernel bodies
using std::atan2;
using std::fabs;
using std::fma;
using std::fmax;
using std::fmin;
using std::sqrt;
