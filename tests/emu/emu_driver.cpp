// emu_driver.cpp (TEST-ONLY) — runs the unmodified kernel bodies of
// mpc_collisionavoidance_amd/csrc/{linearize,qp_ipm}.hpp on the CPU through the fiber-based lane
// emulator (tests/emu/lanes.hpp).  Lets the CPU test suite check the cross-lane algorithm of the
// HIP kernels against the oracle without a GPU.  Not part of the product library.
#include "lanes.hpp"

#include "host_spec.hpp"
#include "linearize.hpp"
#include "models.hpp"
#include "qp_ipm.hpp"
#include "cond_ipm.hpp" // (partial condensing: compiled by the host compiler, one CPU thread plays the team of an instance)
#ifdef USV_GEN_MODEL_HEADER
#include USV_GEN_MODEL_HEADER
#endif

#include <cstring>
#include <vector>

// Build parts (tests/emu/build_emu.py): the kernel bodies of one (model, obstacle chunks) pair take minutes to compile on the host compiler
// too, so the stock emulator library compiles THIS file six times in parallel - -DEMU_PART=0: the fibers, the switches, the C entry points;
// 1 .. 5: run_all of one pair each (explicit instantiation there, extern template elsewhere).  Without EMU_PART: one translation unit (the
// generated-model libraries of genbuild.py).
#ifndef EMU_PART
#define EMU_PART -1
#endif
#define EMU_MAIN (EMU_PART <= 0)

#if EMU_MAIN
namespace lanes {

Emu g_emu;
double *g_emu_lds = nullptr;

// minimal x86-64 SysV context switch: callee-saved registers + stack pointer
asm(R"(
.text
.globl usv_emu_switch
.type usv_emu_switch,@function
usv_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size usv_emu_switch,.-usv_emu_switch
)");

static void (*g_body)(void *) = nullptr;
static void *g_arg = nullptr;

static void fiber_entry()
{
    Emu &e = g_emu;
    g_body(g_arg);
    e.finished[e.cur] = true;
    for (;;) usv_emu_switch(&e.sp[e.cur], e.main_sp);
}

void run_group(long group, void (*body)(void *), void *arg, int rows)
{
    Emu &e = g_emu;
    constexpr size_t STACK = 1 << 20;
    static std::vector<char> stacks(MAXF * STACK);
    const int NF = GROUP * rows;
    g_body = body;
    g_arg = arg;
    e.group = group;
    e.rows = rows;
    for (int l = 0; l < NF; l++) {
        char *top = stacks.data() + (size_t)(l + 1) * STACK;
        uintptr_t sp = ((uintptr_t)top) & ~(uintptr_t)15;
        void **s = (void **)sp;
        // layout popped by usv_emu_switch: r15 r14 r13 r12 rbx rbp, then ret -> fiber_entry with
        // rsp = 8 mod 16 as after a call
        *--s = nullptr;              // alignment slot (acts as the fake return address)
        *--s = (void *)&fiber_entry; // ret target
        for (int i = 0; i < 6; i++) *--s = nullptr;
        e.sp[l] = (void *)s;
        e.par[l] = 0;
        e.nops[l] = 0;
        e.finished[l] = false;
    }
    for (;;) {
        int nfin = 0;
        for (int l = 0; l < NF; l++) {
            if (e.finished[l]) { nfin++; continue; }
            e.cur = l;
            usv_emu_switch(&e.main_sp, e.sp[l]);
            if (e.finished[l]) nfin++;
        }
        if (nfin == NF) break;
        for (int l = 0; l < NF; l++) {
            if (e.finished[l] != e.finished[0] || e.nops[l] != e.nops[0]) {
                std::fprintf(stderr, "lane emulator: divergent cross-lane op (lane %d: %ld ops, fin %d; lane 0: %ld ops, fin %d)\n",
                             l, e.nops[l], (int)e.finished[l], e.nops[0], (int)e.finished[0]);
                std::abort();
            }
        }
    }
    e.rows = 1;
    e.cur = 0;
}

} // namespace lanes
#endif // EMU_MAIN

using namespace usv;

// the test switches (defined in the main part)
#if EMU_MAIN
#define EMU_VAR(decl, init) decl = init
#else
#define EMU_VAR(decl, init) extern decl
#endif
EMU_VAR(int g_emu_cpc, -1);         // option "cond_pred_corr" of the next solves: -1 as the descriptor says (its QP solver profile), 0 / 1 forced
EMU_VAR(double g_emu_cpc_factor, 2.0);
EMU_VAR(int g_emu_lds_mode, 0); // 1: run the RTI solves with the workspace in (emulated) LDS
EMU_VAR(int g_emu_merge, 1);    // 1: box rows processed in their slot lanes when all of them ride there (as the device library does)
EMU_VAR(int g_emu_aux, 0);      // 1: RTI solves of the packed one-chunk layouts keep the aux plane in (emulated) LDS (AUXLDS instantiations)
EMU_VAR(int g_emu_wide, 0);     // 1 / 2 / 4: RTI solves of the packed one-chunk layouts on the WIDE mapping, that many emulated waves per instance
EMU_VAR(long g_emu_wide_runs, 0); // rows started on the WIDE mapping since the switch was set
EMU_VAR(int g_emu_handover_lds, 0); // the follow-up launch copies the planes into LDS (QpIpm::copy_in)
EMU_VAR(int g_emu_handover, 0); // > 0: RTI solves on the 16-lane mapping hand instances past this many iterations over once the queue is empty (QpIpm::suspend);
                                // the follow-up pass resumes them on the WIDE mapping over the planes in "HBM" (usv_qp_resume on the device)
EMU_VAR(long g_emu_handed, 0);  // instances handed over since the switch was set
EMU_VAR(double *g_emu_lam, nullptr); // [B][N+1][nlam] each: filled after the solve when set (usv_emu_set_export)
EMU_VAR(double *g_emu_t, nullptr);
// inspection copies of the lineariser's output: BAt [N][nx][Bp*16] (the packed planes expanded back to one plane
// per row), rb0 [N][Bp*16], gq [N+1][Bp*16]
EMU_VAR(double *g_dbg_BAt, nullptr);
EMU_VAR(double *g_dbg_rb0, nullptr);
EMU_VAR(double *g_dbg_gq, nullptr);
EMU_VAR(long g_emu_rows, 2); // persistent rows of an emulated RTI solve (0: one row per group, no queue)
EMU_VAR(int g_emu_cond_N2, 0); // > 0: RTI solves condense the QP to this many stages first (cond_ipm.hpp)

namespace {

struct Job { const DevPtrs *P; long gid; int qp_phase; int queue0; };

template <class M, int KCH, bool SOFT>
void lin_body(void *a)
{
    Job *j = (Job *)a;
    if (j->P->spec->sim_steps > 1) Linearize<M, KCH, SOFT, true>::run(*j->P, j->gid);
    else Linearize<M, KCH, SOFT, false>::run(*j->P, j->gid);
}
template <class M, int KCH, bool SOFT, bool HDIAG, bool PACK, bool SOFTBOX = false, bool MERGE = false>
void qp_body(void *a)
{
    Job *j = (Job *)a;
    if constexpr (HDIAG && !SOFTBOX) if (g_emu_lds_mode && j->qp_phase == 0) { // the workspace of the (single) emulated row in "LDS"
        QpIpm<M, KCH, SOFT, HDIAG, PACK, SOFTBOX, true, MERGE> q(*j->P, j->gid, 0);
        q.solve(j->qp_phase, j->queue0);
        return;
    }
    if constexpr (HDIAG && PACK && !SOFTBOX) if (g_emu_aux && j->qp_phase == 0) {
        QpIpm<M, KCH, SOFT, HDIAG, PACK, SOFTBOX, false, MERGE, true> q(*j->P, j->gid);
        q.solve(j->qp_phase, j->queue0);
        return;
    }
    QpIpm<M, KCH, SOFT, HDIAG, PACK, SOFTBOX, false, MERGE> q(*j->P, j->gid);
    q.solve(j->qp_phase, j->queue0);
}

// the WIDE mapping: the body runs on the 64 fibers of a whole wave; row 0 owns the LDS region, rows 1 - 3 share it
template <class M, int KCH, bool SOFT, bool MERGE, bool LDSWS, int WW = 1>
void qp_wide_body(void *a)
{
    Job *j = (Job *)a;
    if constexpr (KCH >= 1) {
        QpIpm<M, KCH, SOFT, true, true, false, LDSWS, MERGE, false, true, WW> q(*j->P, j->gid, lanes::block_row() == 0 ? 0 : -1);
        q.solve(j->qp_phase, j->queue0);
    } else if constexpr (KCH == 0 && !MERGE) {
        QpIpm<M, KCH, SOFT, true, false, false, LDSWS, false, false, true, WW> q(*j->P, j->gid, lanes::block_row() == 0 ? 0 : -1);
        q.solve(j->qp_phase, j->queue0);
    }
}
// ... with soft state bounds (box rows in planes of their own, with or without obstacle rows)
template <class M, int KCH, bool SOFT, bool LDSWS, bool SOFTBOX = true, int WW = 1>
void qp_wide_softbox_body(void *a)
{
    Job *j = (Job *)a;
    QpIpm<M, KCH, SOFT, true, false, SOFTBOX, LDSWS, false, false, true, WW> q(*j->P, j->gid, lanes::block_row() == 0 ? 0 : -1);
    q.solve(j->qp_phase, j->queue0);
}
template <class M, int KCH, bool SOFT, bool MERGE, bool LDSWS>
void run_wide(long g, Job &j, int ww)
{
    if (ww == 4) lanes::run_group(g, &qp_wide_body<M, KCH, SOFT, MERGE, LDSWS, 4>, &j, 16);
    else if (ww == 2) lanes::run_group(g, &qp_wide_body<M, KCH, SOFT, MERGE, LDSWS, 2>, &j, 8);
    else lanes::run_group(g, &qp_wide_body<M, KCH, SOFT, MERGE, LDSWS, 1>, &j, 4);
}
template <class M, int KCH, bool SOFT>
size_t wide_lds(int N)
{
    if constexpr (KCH >= 1) return (size_t)QpIpm<M, KCH, SOFT, true, true, false, true, true, false, true, 4>::wide_lds_doubles(N) + 16 * 2 * LANES; // (any variant)
    else if constexpr (KCH == 0) return (size_t)QpIpm<M, KCH, SOFT, true, false, false, true, false, false, true, 4>::wide_lds_doubles(N) + 16 * 2 * LANES;
    else return 0;
}

// multiplier read-back (QpIpm::export_rows) of one group, as the device's usv_qp_export kernel runs it
template <class M, int KCH, bool SOFT, bool PACK, bool SOFTBOX>
void export_body(void *a)
{
    Job *j = (Job *)a;
    QpIpm<M, KCH, SOFT, true, PACK, SOFTBOX> q(*j->P, j->gid);
    q.export_rows();
}

// the condensed solve of every instance, serially (the device runs one team of threads per instance)
template <class M, int KCH, bool SOFT>
int cond_all(const DevPtrs &P, const DevSpec &S, int N2)
{
    CondDims D;
    if (!cond_dims(S, M::NX, M::NU, M::IPX, M::IPY, N2, 1, SOFT, D)) return -4;
    std::vector<double> scratch((size_t)D.total, 0.0), lds((size_t)D.lds_doubles, 0.0);
    for (long g = 0; g < S.B; g++) {
        CondIpm<M, KCH, SOFT, CondTeam> q(P, D, scratch.data(), lds.data());
        q.solve(g);
    }
    return 0;
}

template <class M, int KCH, bool SOFT>
void expand_packed(const DevPtrs &P, const DevSpec &S)
{
    using MP = MatPack<M>;
    using WL = WsLayout<M, KCH, SOFT>;
    const long stride = (long)S.Bp * LANES;
    // workspace element (stage k, plane e, group g, lane r): [stage][group][plane][16 lanes]
    auto at = [&](int k, int e, long g, int r) -> double {
        return P.ws[(((long)k * S.Bp + g) * S.npt + e) * LANES + r];
    };
    for (int k = 0; k <= S.N; k++)
        for (long g = 0; g < S.Bp; g++)
            for (int r = 0; r < LANES; r++) {
                if (g_dbg_gq) g_dbg_gq[(long)k * stride + g * LANES + r] = at(k, WL::P_GQ, g, r);
                if (g_dbg_rb0 && k < S.N) g_dbg_rb0[(long)k * stride + g * LANES + r] = at(k, WL::P_RB0, g, r);
            }
    if (!g_dbg_BAt) return;
    for (int k = 0; k < S.N; k++)
        for (int j = 0; j < M::NX; j++) {
            if ((M::OUT_UNIT >> j) & 1u) continue; // unit rows are not stored: left at zero
            for (long g = 0; g < S.Bp; g++)
                for (int c = 0; c < MP::NZ; c++) {
                    double v;
                    if ((MP::row_mask(j) >> c) & 1u) {
                        const int pos = MP::start(j) + MP::rank(MP::row_mask(j), c);
                        v = at(k, WL::P_MAT + pos / 16, g, pos % 16);
                    } else {
                        v = (c == M::NU + j && MP::diag_one(j)) ? 1.0 : 0.0;
                    }
                    g_dbg_BAt[((long)k * M::NX + j) * stride + g * LANES + c] = v;
                }
        }
}

} // namespace

// (external linkage: a split build defines each instantiation in a translation unit of its own - "Build parts" above)
template <class M, int KCH, bool SOFT>
void run_all(const DevPtrs &P_, const DevSpec &S, int phase, int qp_phase)
{
    const_cast<DevSpec &>(S).npt = S.any_bsoft ? WsLayout<M, KCH, SOFT, true>::NPT : WsLayout<M, KCH, SOFT, false>::NPT;
    if (phase & 1)
        for (long gid = 0; gid < (long)(S.N + 1) * S.Bp; gid++) {
            Job j{&P_, gid, 0, -1};
            lanes::run_group(gid, &lin_body<M, KCH, SOFT>, &j);
        }
    if ((phase & 1) && (g_dbg_BAt || g_dbg_rb0 || g_dbg_gq)) expand_packed<M, KCH, SOFT>(P_, S);
    // RTI: a few persistent rows that pull the remaining groups from the queue (as the device launch does); full SQP: one
    // group per row
    if ((phase & 2) && qp_phase == 0 && g_emu_cond_N2 > 0 && !S.any_bsoft) {
        if (cond_all<M, KCH, SOFT>(P_, S, g_emu_cond_N2)) std::abort();
        return;
    }
    const bool queue = qp_phase == 0 && g_emu_rows > 0 && g_emu_rows < S.Bp;
    const long nrows = queue ? g_emu_rows : S.Bp;
    if (queue) *P_.queue = 0;
    // hand-over (RTI launches that refill from the queue, layouts with a WIDE mapping): the lists of this "launch"
    std::vector<int> susp_list((size_t)S.B + 1, 0);
    std::vector<double> susp_rec((size_t)S.B * 4 + 4, 0.0);
    int susp_count = 0;
    DevPtrs Ph = P_;
    const bool hand = g_emu_handover > 0 && !g_emu_wide && !g_emu_lds_mode && S.hdiag && !S.any_bsoft && ((KCH == 1 && S.boxpack != 0) || KCH == 0);
    if (hand) { Ph.susp_count = &susp_count; Ph.susp_list = susp_list.data(); Ph.susp_rec = susp_rec.data(); Ph.handover_iter = g_emu_handover; }
    const DevPtrs &P = Ph;
    if (phase & 2)
        for (long g = 0; g < nrows; g++) {
            static std::vector<double> lds; // the emulated row's LDS region (allocated here: the body runs once per lane)
            lds.assign((size_t)(S.N + 1) * S.npt * LANES, 0.0);
            lanes::g_emu_lds = lds.data();
            Job j{&P, g, qp_phase, queue ? (int)nrows : -1};
            constexpr bool CANPACK = KCH > 0;
            const bool pack = CANPACK && S.boxpack != 0;
            if (S.any_bsoft && S.hdiag && g_emu_wide && (qp_phase == 0 || !g_emu_lds_mode)) { // the WIDE mapping, one emulated wave per instance
                lds.assign((size_t)QpIpm<M, KCH, SOFT, true, false, true, true, false, false, true, 1>::wide_lds_doubles(S.N) + 16 * 2 * LANES, 0.0);
                lanes::g_emu_lds = lds.data();
                if (g_emu_lds_mode) lanes::run_group(g, &qp_wide_softbox_body<M, KCH, SOFT, true>, &j, 4);
                else lanes::run_group(g, &qp_wide_softbox_body<M, KCH, SOFT, false>, &j, 4);
                g_emu_wide_runs++;
                continue;
            }
            if (S.any_bsoft) {
                if (S.hdiag) lanes::run_group(g, &qp_body<M, KCH, SOFT, true, false, true>, &j);
                else lanes::run_group(g, &qp_body<M, KCH, SOFT, false, false, true>, &j);
                continue;
            }
            // (the launches of a full SQP - qp_phase 1 / 2 - on the WIDE mapping: over the planes in HBM, where the multipliers persist)
            if constexpr (KCH >= 1) {
                if (!pack && g_emu_wide && (qp_phase == 0 || !g_emu_lds_mode) && S.hdiag) { // box rows in planes of their own beside obstacle rows
                    lds.assign((size_t)QpIpm<M, KCH, SOFT, true, false, false, true, false, false, true, 1>::wide_lds_doubles(S.N) + 16 * 2 * LANES, 0.0);
                    lanes::g_emu_lds = lds.data();
                    if (g_emu_lds_mode) lanes::run_group(g, &qp_wide_softbox_body<M, KCH, SOFT, true, false>, &j, 4);
                    else lanes::run_group(g, &qp_wide_softbox_body<M, KCH, SOFT, false, false>, &j, 4);
                    g_emu_wide_runs++;
                    continue;
                }
            }
            if (((KCH >= 1 && pack) || KCH == 0) && g_emu_wide && (qp_phase == 0 || !g_emu_lds_mode) && S.hdiag) {
                lds.assign(wide_lds<M, KCH, SOFT>(S.N), 0.0);
                lanes::g_emu_lds = lds.data();
                // (lds mode off: the wide sweeps over the planes in HBM - horizons that do not fit a CU's LDS)
                const bool mg = KCH >= 1 && g_emu_merge && !S.box_dense;
                if (g_emu_lds_mode) { if (mg) run_wide<M, KCH, SOFT, true, true>(g, j, g_emu_wide); else run_wide<M, KCH, SOFT, false, true>(g, j, g_emu_wide); }
                else { if (mg) run_wide<M, KCH, SOFT, true, false>(g, j, g_emu_wide); else run_wide<M, KCH, SOFT, false, false>(g, j, g_emu_wide); }
                g_emu_wide_runs++;
                continue;
            }
            if (S.hdiag && pack && g_emu_merge && !S.box_dense) lanes::run_group(g, &qp_body<M, KCH, SOFT, true, CANPACK, false, CANPACK>, &j);
            else if (S.hdiag) lanes::run_group(g, pack ? &qp_body<M, KCH, SOFT, true, CANPACK> : &qp_body<M, KCH, SOFT, true, false>, &j);
            else lanes::run_group(g, pack ? &qp_body<M, KCH, SOFT, false, CANPACK> : &qp_body<M, KCH, SOFT, false, false>, &j);
        }
    if constexpr (KCH <= 1) {
        if ((phase & 2) && hand) { // the follow-up launch: one emulated wave per suspended instance, planes in "HBM"
            static std::vector<double> lds;
            for (int i = 0; i < susp_count; i++) {
                lds.assign(wide_lds<M, KCH, SOFT>(S.N), 0.0);
                lanes::g_emu_lds = lds.data();
                Job j{&P, (long)susp_list[i], 3, -1};
                const bool mg = KCH == 1 && g_emu_merge && !S.box_dense;
                if (g_emu_handover_lds) {
                    if (mg) run_wide<M, KCH, SOFT, true, true>((long)susp_list[i], j, 1);
                    else run_wide<M, KCH, SOFT, false, true>((long)susp_list[i], j, 1);
                } else if (mg) run_wide<M, KCH, SOFT, true, false>((long)susp_list[i], j, 1);
                else run_wide<M, KCH, SOFT, false, false>((long)susp_list[i], j, 1);
            }
            g_emu_handed += susp_count;
        }
    }
    if ((phase & 2) && P.lam_out)
        for (long g = 0; g < S.Bp; g++) {
            Job j{&P, g, 0, -1};
            constexpr bool CANPACK = KCH > 0;
            const bool pack = CANPACK && S.boxpack != 0;
            if (S.any_bsoft) lanes::run_group(g, &export_body<M, KCH, SOFT, false, true>, &j);
            else if (pack) lanes::run_group(g, &export_body<M, KCH, SOFT, CANPACK, false>, &j);
            else lanes::run_group(g, &export_body<M, KCH, SOFT, false, false>, &j);
        }
}

#if EMU_PART >= 0 && !defined(USV_GEN_ONLY) && !defined(USV_GEN_MODEL_HEADER)
#define EMU_PAIR(PART, M, KCH, SOFT) EMU_PAIR_##PART(M, KCH, SOFT)
#define EMU_DEF(M, KCH, SOFT) template void run_all<M, KCH, SOFT>(const DevPtrs &, const DevSpec &, int, int);
#define EMU_EXT(M, KCH, SOFT) extern template void run_all<M, KCH, SOFT>(const DevPtrs &, const DevSpec &, int, int);
#if EMU_PART == 1
EMU_DEF(ModelM0, 0, false)
#else
EMU_EXT(ModelM0, 0, false)
#endif
#if EMU_PART == 2
EMU_DEF(ModelM1, 1, true)
#else
EMU_EXT(ModelM1, 1, true)
#endif
#if EMU_PART == 3
EMU_DEF(ModelM1, 2, true)
#else
EMU_EXT(ModelM1, 2, true)
#endif
#if EMU_PART == 4
EMU_DEF(ModelM2, 1, false)
#else
EMU_EXT(ModelM2, 1, false)
#endif
#if EMU_PART == 5
EMU_DEF(ModelM2, 2, false)
#else
EMU_EXT(ModelM2, 2, false)
#endif
#endif

#if EMU_MAIN

// One RTI iteration (sqp = 0) or a full SQP run (sqp = 1: the host loop of usvmpc_solve_sqp) of every instance
// on the emulator. Arrays as in include/usvmpc.h (host).
// dbg (optional, may be NULL) receives the linearisation planes for inspection:
// BAt [N][nx][Bp*16], rb0 [N][Bp*16], gq [N+1][Bp*16].
static int emu_run(const usvmpc_desc *d, int sqp, double *x, double *u, const double *x0,
                   const double *yref, const double *yref_e, const double *p,
                   const double *lh, double *sl, double *su, double *pi, int *status,
                   int *qp_status, int *qp_iter, double *res, double *dbg_BAt,
                   double *dbg_rb0, double *dbg_gq, int *sqp_iter, double *nlp_res)
{
    DevSpec S;
    const std::string err = build_spec(*d, S);
    if (!err.empty()) {
        std::fprintf(stderr, "usv_emu_solve: %s\n", err.c_str());
        return -1;
    }
    if (g_emu_cpc >= 0) { S.cpc = g_emu_cpc; S.cpc_factor = g_emu_cpc_factor; }
    int nx, nu;
    model_dims(d->model, nx, nu);
    const int N = S.N;
    int kch = (S.K + LANES - 1) / LANES;
    bool soft = d->soft != 0;
#ifdef USV_GEN_MODEL_HEADER
    if (d->model == USVMPC_MODEL_GENERATED) { kch = USV_GEN_KCH; soft = USV_GEN_SOFT != 0; }
#endif
    const long stride = (long)S.Bp * LANES;
    std::vector<double> ws((size_t)(N + 1) * ws_planes(nx, nu, kch, soft, 16, true) * stride), // upper bounds: 16 >= MatPack::NPK, soft-box planes
        nres((size_t)S.B * 4);
    std::vector<int> sit(S.B, 0), sstate(S.B, -1);
    int running = 0;
    DevPtrs P;
    std::memset(&P, 0, sizeof(P));
    P.spec = &S;
    P.x = x; P.u = u; P.x0 = x0; P.yref = yref; P.yref_e = yref_e; P.p = p; P.lh = lh;
    P.sl = sl; P.su = su; P.pi = pi; P.status = status; P.qp_iter = qp_iter; P.qp_status = qp_status; P.res = res;
    P.ws = ws.data();
    P.nlp_res = nres.data(); P.sqp_iter = sit.data(); P.sqp_state = sstate.data(); P.sqp_running = &running;
    int queue = 0;
    P.queue = &queue;
    if (g_emu_lam && g_emu_t) {
        P.lam_out = g_emu_lam; P.t_out = g_emu_t; P.nlam = lam_len(S, soft);
        std::memset(g_emu_lam, 0, sizeof(double) * (size_t)S.B * (N + 1) * P.nlam);
        std::memset(g_emu_t, 0, sizeof(double) * (size_t)S.B * (N + 1) * P.nlam);
    }
    g_dbg_BAt = dbg_BAt; g_dbg_rb0 = dbg_rb0; g_dbg_gq = dbg_gq;
    auto one = [&](int qp_phase) -> int {
        const int phase = 3;
#ifdef USV_GEN_MODEL_HEADER
        if (d->model == USVMPC_MODEL_GENERATED) { run_all<ModelGen, USV_GEN_KCH, (USV_GEN_SOFT != 0)>(P, S, phase, qp_phase); return 0; }
#endif
#ifndef USV_GEN_ONLY
        if (d->model == USVMPC_MODEL_USV) run_all<ModelM0, 0, false>(P, S, phase, qp_phase);
        else if (d->model == USVMPC_MODEL_GUIDANCE_CA1) {
            if (!soft) return -2;
            if (kch <= 1) run_all<ModelM1, 1, true>(P, S, phase, qp_phase);
            else run_all<ModelM1, 2, true>(P, S, phase, qp_phase);
        } else if (d->model == USVMPC_MODEL_PF_CA) {
            if (soft) return -2;
            if (kch <= 1) run_all<ModelM2, 1, false>(P, S, phase, qp_phase);
            else run_all<ModelM2, 2, false>(P, S, phase, qp_phase);
        } else return -3;
        return 0;
#else
        return -3;
#endif
    };
    if (!sqp) {
        const int rc = one(0);
        if (rc) return rc;
    } else {
        const int max_iter = d->nlp_max_iter > 0 ? d->nlp_max_iter : 100;
        for (int it = 0; it < max_iter; it++) {
            running = 0;
            const int rc = one(it == 0 ? 1 : 2);
            if (rc) return rc;
            if (running == 0) break;
        }
        for (int b = 0; b < S.B; b++) {
            status[b] = sstate[b] < 0 ? 2 : sstate[b];
            if (sqp_iter) sqp_iter[b] = sit[b];
            if (nlp_res) std::memcpy(nlp_res + (size_t)b * 4, nres.data() + (size_t)b * 4, 4 * sizeof(double));
        }
    }
    return 0;
}

// test switches: workspace of the RTI solves in emulated LDS (lds != 0); persistent rows pulling from the queue (rows, 0 = none)
extern "C" void usv_emu_set_mode(int lds, long rows) { g_emu_lds_mode = lds; g_emu_rows = rows; }
extern "C" void usv_emu_set_merge(int merge) { g_emu_merge = merge; }
extern "C" void usv_emu_set_cond(int N2) { g_emu_cond_N2 = N2; }
extern "C" void usv_emu_set_aux(int aux) { g_emu_aux = aux; }
extern "C" void usv_emu_set_wide(int wide) { g_emu_wide = wide; g_emu_wide_runs = 0; }
extern "C" long usv_emu_wide_runs() { return g_emu_wide_runs; }
extern "C" void usv_emu_set_cpc(int on, double factor) { g_emu_cpc = on; g_emu_cpc_factor = factor; }
extern "C" void usv_emu_set_handover(int iters) { g_emu_handover = iters; g_emu_handed = 0; }
extern "C" void usv_emu_set_handover_lds(int on) { g_emu_handover_lds = on; }
extern "C" long usv_emu_handed() { return g_emu_handed; }
// the next solves also deliver the multipliers / slacks of their QPs (the device's usvmpc_get "lam" / "t"); NULL switches it off
extern "C" void usv_emu_set_export(double *lam, double *t) { g_emu_lam = lam; g_emu_t = t; }

extern "C" int usv_emu_solve(const usvmpc_desc *d, double *x, double *u, const double *x0,
                             const double *yref, const double *yref_e, const double *p,
                             const double *lh, double *sl, double *su, double *pi, int *status,
                             int *qp_status, int *qp_iter, double *res, double *dbg_BAt,
                             double *dbg_rb0, double *dbg_gq)
{
    return emu_run(d, 0, x, u, x0, yref, yref_e, p, lh, sl, su, pi, status, qp_status, qp_iter, res, dbg_BAt, dbg_rb0,
                   dbg_gq, nullptr, nullptr);
}

extern "C" int usv_emu_sqp(const usvmpc_desc *d, double *x, double *u, const double *x0,
                           const double *yref, const double *yref_e, const double *p,
                           const double *lh, double *sl, double *su, double *pi, int *status,
                           int *qp_status, int *qp_iter, double *res, int *sqp_iter, double *nlp_res)
{
    return emu_run(d, 1, x, u, x0, yref, yref_e, p, lh, sl, su, pi, status, qp_status, qp_iter, res, nullptr, nullptr,
                   nullptr, sqp_iter, nlp_res);
}

// ---- the pipelined lineariser's three modes (linearize.hpp) on the emulator: MODE 0 under `perm_next` into ws_plain; MODE 1 under
// (perm_next, perm_cur) with epoch[b] = ready[b] ? tick : tick - 1, then MODE 2, into ws_piped; redo_out receives the mask MODE 1 left
// ([B][(N + 32) / 32]).  Workspaces: [(N + 1)][Bp][npt][16] doubles each (npt returned).
namespace {
template <class M, int KCH, bool SOFT, int MODE>
void lin_mode_body(void *a)
{
    Job *j = (Job *)a;
    if (j->P->spec->sim_steps > 1) Linearize<M, KCH, SOFT, true, MODE>::run(*j->P, j->gid);
    else Linearize<M, KCH, SOFT, false, MODE>::run(*j->P, j->gid);
}
template <class M, int KCH, bool SOFT>
int lin_modes(DevPtrs P, DevSpec &S, const int *ready, const int *perm_next, const int *perm_cur, double *ws_plain, double *ws_piped, int *redo_out)
{
    S.npt = S.any_bsoft ? WsLayout<M, KCH, SOFT, true>::NPT : WsLayout<M, KCH, SOFT, false>::NPT;
    const long ngroups = (long)(S.N + 1) * S.Bp;
    const int tick = 7, words = (S.N + 32) / 32;
    std::vector<int> epoch(S.B), redo((size_t)S.B * words, 0);
    for (int b = 0; b < S.B; b++) epoch[b] = ready[b] ? tick : tick - 1;
    P.spec = &S; P.perm = perm_next; P.tick = tick;
    P.ws = ws_plain;
    for (long gid = 0; gid < ngroups; gid++) { Job j{&P, gid, 0, -1}; lanes::run_group(gid, &lin_mode_body<M, KCH, SOFT, 0>, &j); }
    P.ws = ws_piped; P.epoch = epoch.data(); P.redo = redo.data(); P.redo_words = words; P.perm_cur = perm_cur;
    for (long gid = 0; gid < ngroups; gid++) { Job j{&P, gid, 0, -1}; lanes::run_group(gid, &lin_mode_body<M, KCH, SOFT, 1>, &j); }
    std::memcpy(redo_out, redo.data(), redo.size() * sizeof(int));
    for (long gid = 0; gid < ngroups; gid++) { Job j{&P, gid, 0, -1}; lanes::run_group(gid, &lin_mode_body<M, KCH, SOFT, 2>, &j); }
    return S.npt;
}
} // namespace

extern "C" int usv_emu_lin_modes(const usvmpc_desc *d, const double *x, const double *u, const double *yref, const double *yref_e,
                                 const int *ready, const int *perm_next, const int *perm_cur, double *ws_plain, double *ws_piped, int *redo_out)
{
    DevSpec S;
    if (!build_spec(*d, S).empty()) return -1;
    DevPtrs P;
    std::memset(&P, 0, sizeof(P));
    P.x = const_cast<double *>(x); P.u = const_cast<double *>(u); P.yref = yref; P.yref_e = yref_e;
#ifndef USV_GEN_ONLY
    if (d->model == USVMPC_MODEL_USV) return lin_modes<ModelM0, 0, false>(P, S, ready, perm_next, perm_cur, ws_plain, ws_piped, redo_out);
    if (d->model == USVMPC_MODEL_GUIDANCE_CA1) return lin_modes<ModelM1, 1, true>(P, S, ready, perm_next, perm_cur, ws_plain, ws_piped, redo_out);
    if (d->model == USVMPC_MODEL_PF_CA) return lin_modes<ModelM2, 1, false>(P, S, ready, perm_next, perm_cur, ws_plain, ws_piped, redo_out);
#endif
    return -3;
}
#endif // EMU_MAIN
