// linearize.hpp — preparation phase of one SQP-RTI iteration, one 16-lane group per
// (instance, stage) pair.  Replaces acados' sim_erk (+ generated *_expl_vde_forw),
// ocp_nlp_cost_ls evaluation (the bgh obstacle rows are evaluated in qp_ipm.hpp) for the reference's OCPs
// (/root/reference/catkin_ws/src/nmpc_ca/scripts/usv_guidance_ca1/acados_settings.py:83-194).
//
// Lane r (variable r of [u;x]) integrates ITS OWN column of the forward sensitivities
// S = [dx+/du | dx+/dx] through the 4 RK stages (the VDE is linear in S column by column), while
// every lane carries the nominal RK stage points.  The lane therefore ends up holding row r of
// [B A]' — exactly the operand layout of the Riccati recursion (qp_ipm.hpp) — with no transpose; the
// informative entries are then packed densely for HBM (MatPack, params.hpp).
#pragma once
#include "lanes.hpp"
#include "params.hpp"
#include "sfor.hpp"

namespace usv {

// A model may prepare quantities that are constant over a shooting interval (they depend only on states whose
// right-hand side is zero): `struct Pre`, `prepare(x)`, `fjvp_pre(pre, ...)`.  Models without them are called through
// their plain fjvp.
template <class M, class = void>
struct ModelCall {
    struct Pre {};
    USV_DEV static Pre prepare(const double *) { return {}; }
    USV_DEV static void fjvp(const Pre &, const double *x, const double *U, const double *s, const double *su, double *f, double *js)
    {
        M::fjvp(x, U, s, su, f, js);
    }
};
template <class M>
struct ModelCall<M, std::void_t<typename M::Pre>> {
    using Pre = typename M::Pre;
    USV_DEV static Pre prepare(const double *x) { return M::prepare(x); }
    USV_DEV static void fjvp(const Pre &p, const double *x, const double *U, const double *s, const double *su, double *f, double *js)
    {
        M::fjvp_pre(p, x, U, s, su, f, js);
    }
};

// MULTI: more than one RK4 step per interval (the initial sensitivity column is then a carried variable
// instead of a lane pattern the compiler rematerialises for free: 28 more VGPRs for M2, hence a separate build)
// MODE 0: every (instance, stage) of the batch, between two launches of the QP kernel (the plain path).
// MODE 1: speculative, for the NEXT tick, while the QP launch of tick P.tick is still running (second stream, its workgroups fill the
//         compute units that launch vacates in its tail): a group goes ahead only if its instance's results of that tick are final
//         AND so are those of the instance that still owns the group's planes under the running launch's map; otherwise it marks the
//         instance for MODE 2.  The iterate is read with loads that bypass the non-coherent L2s (lanes::ld_shared).
// MODE 2: fix-up after that launch: only the instances MODE 1 marked.
template <class M, int KCH, bool SOFT, bool MULTI = false, int MODE = 0>
struct Linearize {
    static constexpr int NX = M::NX, NU = M::NU, NZ = NX + NU;
    using WL = WsLayout<M, KCH, SOFT>;

    // gid = k * Bp + g  (groups of a wave share the stage k)
    USV_DEV static void run(const DevPtrs &P, long gid)
    {
        const DevSpec &S = *P.spec;
        const int lane = lanes::lane();
        const int N = lanes::uniform(S.N), K = lanes::uniform(S.K);
        const long Bp = lanes::uniform(S.Bp);
        const int k = (int)(gid / Bp);
        const long g = gid - (long)k * Bp;
        const int nB = lanes::uniform(S.B);
        const long gi = g < nB ? g : (long)nB - 1; // padded groups replay the last instance
        const long b = P.perm ? (long)P.perm[gi] : gi;
        if constexpr (MODE == 1) {
            const long owner = P.perm_cur ? (long)P.perm_cur[gi] : gi;
            const bool ready = lanes::observe(P.epoch + b) == P.tick && lanes::observe(P.epoch + owner) == P.tick;
            if (!ready) { // (the whole 16-lane group leaves: nothing below crosses groups)
                if (lane == 0) lanes::set_bits(P.redo + b * P.redo_words + (k >> 5), 1 << (k & 31)); // this stage of this instance: later
                return;
            }
        }
        if constexpr (MODE == 2) {
            if (((P.redo[b * P.redo_words + (k >> 5)] >> (k & 31)) & 1) == 0) return;
        }
        auto ld = [](const double *q) { // the iterate: handed over by a kernel that may still be running (MODE 1)
            if constexpr (MODE == 1) return lanes::ld_shared(q);
            else return *q;
        };
        // workspace: [stage][group][plane][16 lanes] (lanes::Planes)
        double *tile = P.ws + (((long)k * Bp + g) * lanes::uniform(S.npt)) * LANES + lane;
        const bool xlane = lane >= NU && lane < NZ;

        double x[NX], U[NU > 0 ? NU : 1];
        const double *xk = P.x + ((long)b * (N + 1) + k) * NX;
        sfor<0, NX>([&](auto i) { x[i] = ld(xk + i); });
        if (k < N) {
            const double *uk = P.u + ((long)b * N + k) * NU;
            sfor<0, NU>([&](auto i) { U[i] = ld(uk + i); });
        } else {
            sfor<0, NU>([&](auto i) { U[i] = 0.0; });
        }

        // ---- cost gradient, reference part: -Mc yref (stage) | -Me yref_e (terminal).  The QP kernel keeps its iterate in
        // absolute form (zbar + z), so the gradient at it is this plus H (zbar + z): the H zbar term is not formed here ----
        {
            const double *Mrow = (k < N ? S.Mc : S.Me) + lane * LANES;
            const double *yr = (k < N) ? P.yref + ((long)b * N + k) * S.ny : P.yref_e + (long)b * S.ny_e;
            const int ny = (k < N) ? S.ny : S.ny_e;
            double acc = 0.0;
            for (int y = 0; y < ny; y++) acc = fma(-Mrow[y], yr[y], acc);
            tile[WL::P_GQ * LANES] = acc;
        }
        if (k == N) return; // wave-uniform

        // ---- ERK4 + forward VDE for this lane's sensitivity column; sim_steps steps of size dt / sim_steps
        // (acados sim_method_num_steps; the reference leaves it at 1): the column is simply carried on ----
        const int nsteps = MULTI ? S.sim_steps : 1;
        const double dt = S.dt / (double)nsteps;
        double s0[NX], f[NX], js[NX], xs[NX], ss[NX], xa[NX], sa[NX], su[NU > 0 ? NU : 1];
        sfor<0, NU>([&](auto l) { su[l] = (lane == l) ? 1.0 : 0.0; });
        sfor<0, NX>([&](auto i) { s0[i] = (lane == NU + i) ? 1.0 : 0.0; });
        using MC = ModelCall<M>;
        const typename MC::Pre pre = MC::prepare(x);
        for (int step = 0; step < nsteps; step++) { // wave-uniform
            MC::fjvp(pre, x, U, s0, su, f, js);
            sfor<0, NX>([&](auto i) {
                xa[i] = f[i];
                sa[i] = js[i];
                xs[i] = fma(0.5 * dt, f[i], x[i]);
                ss[i] = fma(0.5 * dt, js[i], s0[i]);
            });
            MC::fjvp(pre, xs, U, ss, su, f, js);
            sfor<0, NX>([&](auto i) {
                xa[i] = fma(2.0, f[i], xa[i]);
                sa[i] = fma(2.0, js[i], sa[i]);
                xs[i] = fma(0.5 * dt, f[i], x[i]);
                ss[i] = fma(0.5 * dt, js[i], s0[i]);
            });
            MC::fjvp(pre, xs, U, ss, su, f, js);
            sfor<0, NX>([&](auto i) {
                xa[i] = fma(2.0, f[i], xa[i]);
                sa[i] = fma(2.0, js[i], sa[i]);
                xs[i] = fma(dt, f[i], x[i]);
                ss[i] = fma(dt, js[i], s0[i]);
            });
            MC::fjvp(pre, xs, U, ss, su, f, js);
            sfor<0, NX>([&](auto i) {
                x[i] = fma(dt / 6.0, xa[i] + f[i], x[i]);
                sa[i] = fma(dt / 6.0, sa[i] + js[i], s0[i]);
                if constexpr (MULTI) s0[i] = sa[i];
            });
        }
        const double *xn = P.x + ((long)b * (N + 1) + k + 1) * NX;
        double bres = 0.0;
        sfor<0, NX>([&](auto i) { bres = (lane == NU + i) ? x[i] - ld(xn + i) : bres; });
        // lane r now holds row r of [B A]' (sa[i] = d x+_i / d z_r).  The structurally informative entries (MatPack:
        // M::SENS) are packed into MatPack<M>::NPK planes: lane L of plane q stores entry number 16 q + L of the
        // stream, i.e. for the row j whose range contains it the value sa[j] held by the lane of its column - a lane
        // gather per row that the plane touches.
        using MP = MatPack<M>;
        sfor<0, MP::NPK>([&](auto q) {
            const int sidx = 16 * q + lane;
            double val = 0.0;
            sfor<0, NX>([&](auto j) {
                constexpr int s0 = MP::start(j), cnt = MP::count(j);
                if constexpr (cnt > 0 && s0 < 16 * q + 16 && s0 + cnt > 16 * q) {
                    const int within = sidx - s0;
                    int c_l = 0; // lane that owns this entry's column
                    sfor<0, cnt>([&](auto ci) { c_l = (within == ci) ? MP::nth(MP::row_mask(j), ci) : c_l; });
                    const double gth = lanes::gather(sa[j], c_l);
                    val = (within >= 0 && within < cnt) ? gth : val;
                }
            });
            tile[(WL::P_MAT + q) * LANES] = val;
        });
        tile[WL::P_RB0 * LANES] = xlane ? bres : 0.0;
        // (obstacle rows are linearised inside the QP kernel from the iterate and (p, lh): QpIpm::obs_geom)
    }
};

} // namespace usv
