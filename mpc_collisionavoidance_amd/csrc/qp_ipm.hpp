// qp_ipm.hpp — feedback phase of one SQP-RTI iteration: the OCP-structured QP solved by a
// Mehrotra predictor-corrector interior-point method on a Riccati recursion, followed by the
// full RTI step.  Replaces acados' ocp_qp_hpipm (HPIPM d_ocp_qp_ipm_solve + Riccati KKT
// factor/solve) and ocp_nlp_update_variables for the reference's OCPs (qp_solver =
// PARTIAL_CONDENSING_HPIPM with the default block size, i.e. no condensing:
// /root/reference/catkin_ws/src/nmpc_ca/scripts/usv_guidance_ca1/acados_settings.py:190-194).
//
// Scheduling: a wave is four independent rows.  Each row works through OCP instances ("groups", in the order of the
// difficulty binning): when its QP has converged it writes its results, takes the next unclaimed group from a device
// counter and cold-starts it, while the other three rows carry on with their own iterations - no row waits for the
// slowest of its wave, and no plane of a finished instance is streamed again.  Only the sweeps run in lock step;
// which iteration a row is in is per-row state.  (Without the queue - full SQP, option dynamic_rows = 0 - a row
// keeps its first group and idles once it is done.)
//
// Mapping: one OCP instance = one 16-lane DPP row (lanes.hpp).  Lane r owns variable r of the
// stage vector z = [u;x], row r of every stage matrix ([B A]', P, G), the box constraint on
// variable r, and obstacle row c*16+r of chunk c.  All matrix products are "own row x broadcast
// row" FMAs; the reductions are the obstacle-row sums, the nu gain dot products and the rows of the
// forward product [B A] dz (the stage matrix is stored once, packed: MatPack in params.hpp).
//
// Riccati form: classical (explicit, symmetric P_k; Cholesky of the nu x nu block only).  In the
// row-per-lane layout every product it needs is a natural one, whereas the square-root form
// would need transposed products (L'b) and a 16-step sequential potrf per stage.  P_k never
// leaves the registers: it is consumed by the next stage of the same backward sweep.
//
// Dynamics multipliers by the adjoint recursion: the Newton step (dz, dlam, dt) does not depend
// on the current pi, so instead of carrying pi += alpha*dpi (which needs P_k dx in the forward
// sweep, i.e. P_k stored and re-read), pi_k := (H z + g - C'(ll-lu))_x + A_k' pi_{k+1} is
// recomputed inside the backward sweep from the current primal/inequality iterate.  This zeroes
// the x rows of the stationarity residual by construction; the u rows remain as the residual.
//
// Four sweeps over the horizon per IPM iteration, all state streamed through lane-major planes
// in HBM: backward A (apply the previous step, residuals, factorise, predictor rhs), forward A
// (affine step), backward B (corrector rhs), forward B (step + step length).
//
// The primal iterate is kept in absolute form, zfull = zbar + z (plane P_Z): the box rows and the Hessian product
// only ever need the sum, so the linearisation point is not streamed beside it; the two values of it that the
// obstacle rows do need (its position) ride in the aux plane with the other per-stage odds and ends (WsLayout).
#pragma once
#include "lanes.hpp"
#include "params.hpp"
#include "sfor.hpp"

// Floating-point contraction is taken out of the optimiser's hands for everything in this file: a multiply-add is fused where ONE source
// expression says a * b + c (the language rule, formed by the front end before any inlining) or where the code says fma / lanes::fma_bc,
// and nowhere else.  Under the default (fuse whatever ends up adjacent after inlining) the same source contracted differently from
// instantiation to instantiation, and the mappings of one library - 16 lanes per instance, one instance per wave, planes in LDS or in
// HBM - returned iterates that differed by rounding, which the hard-row model amplifies (an instance's result then depended on the size
// of the batch it sat in).  With it they return the same bits (tests/test_gpu_wide.py).
#pragma clang fp contract(on)

namespace usv {

// One two-sided inequality row  dl <= v (+ sl),  v (- su) <= du  with its multipliers/slacks,
// and the closed-form elimination of (lambda, t, sl, su) used by HPIPM-style IPMs.
// MIXED (only with SOFTROW): the rows handled by this code are partly soft - `soft` says which; a hard row keeps
// sl = su = 0 and its slack multipliers out of everything.
template <bool SOFTROW, bool MIXED = false>
struct RowCalc {
    double ll, lu, tl, tu, sl, su, lsl, lsu, tsl, tsu;          // state
    double dl, du, zl, zu, Zl, Zu, bsl, bsu;                    // data
    bool act, soft;
    double rdl, rdu, rsl, rsu, rdsl, rdsu;                      // residuals
    USV_DEV bool is_soft() const
    {
        if constexpr (!SOFTROW) return false;
        else if constexpr (MIXED) return soft;
        else return true;
    }
    double itl, itu, itsl, itsu;                                // reciprocals of the slacks
    double Gl, Gu, iDl, iDu, rhol, rhou;                        // elimination
    double ml, mu, msl, msu;                                    // complementarity targets
    double dll, dlu, dtl, dtu, dsl, dsu, dlsl, dlsu, dtsl, dtsu; // step

    USV_DEV void neutral()
    {
        ll = lu = 0.0; tl = tu = 1.0; sl = su = 0.0; lsl = lsu = 0.0; tsl = tsu = 1.0;
        dl = -1.0; du = 1.0; zl = zu = 0.0; Zl = Zu = 1.0; bsl = bsu = -1.0;
        soft = SOFTROW && !MIXED;
    }
    USV_DEV void resid(double v)
    {
        rdl = v + sl - dl - tl;
        rdu = du - v + su - tu;
        lanes::frcp2(tl, tu, itl, itu); // (both reciprocals of a pair from one v_rcp_f64)
        if constexpr (SOFTROW) {
            rsl = Zl * sl + zl - ll - lsl;
            rsu = Zu * su + zu - lu - lsu;
            rdsl = sl - bsl - tsl;
            rdsu = su - bsu - tsu;
            lanes::frcp2(tsl, tsu, itsl, itsu);
            if constexpr (MIXED) {
                if (!soft) { rsl = 0.0; rsu = 0.0; rdsl = 0.0; rdsu = 0.0; }
            }
        }
    }
    USV_DEV void targets_pred()
    {
        ml = ll * tl; mu = lu * tu;
        if constexpr (SOFTROW) { msl = lsl * tsl; msu = lsu * tsu; }
    }
    // CPC (option "cond_pred_corr"): so = 1 - the Mehrotra corrector as above -, or 0: the centring-only target of a row whose corrected step
    // was refused (QpIpm::solve); with so = 1 the values are the plain form's bit for bit (1 * x is x)
    template <bool CPC = false>
    USV_DEV void targets_corr(double sigmu, double so = 1.0) // uses the affine step currently held in d*
    {
        if constexpr (!CPC) {
            ml = ll * tl + dll * dtl - sigmu; mu = lu * tu + dlu * dtu - sigmu;
            if constexpr (SOFTROW) { msl = lsl * tsl + dlsl * dtsl - sigmu; msu = lsu * tsu + dlsu * dtsu - sigmu; }
        } else {
            ml = ll * tl + so * (dll * dtl) - sigmu; mu = lu * tu + so * (dlu * dtu) - sigmu;
            if constexpr (SOFTROW) { msl = lsl * tsl + so * (dlsl * dtsl) - sigmu; msu = lsu * tsu + so * (dlsu * dtsu) - sigmu; }
        }
        if constexpr (SOFTROW && MIXED) {
            if (!soft) { msl = 0.0; msu = 0.0; }
        }
    }
    // Gh: coefficient of c c' added to the stage Hessian; gam: coefficient of c added to the gradient
    USV_DEV void reduce(double &Gh, double &gam)
    {
        Gl = ll * itl; Gu = lu * itu;
        double Ghl, Ghu, gl, gu;
        if constexpr (SOFTROW) {
            const double Gsl = lsl * itsl, Gsu = lsu * itsu;
            iDl = lanes::frcp(Zl + Gl + Gsl);
            iDu = lanes::frcp(Zu + Gu + Gsu);
            rhol = -rsl - ml * itl - Gl * rdl - msl * itsl - Gsl * rdsl;
            rhou = -rsu - mu * itu - Gu * rdu - msu * itsu - Gsu * rdsu;
            Ghl = Gl * (1.0 - Gl * iDl);
            Ghu = Gu * (1.0 - Gu * iDu);
            gl = ml * itl + Gl * rdl + Gl * rhol * iDl;
            gu = mu * itu + Gu * rdu + Gu * rhou * iDu;
            if constexpr (MIXED) {
                if (!soft) {
                    Ghl = Gl; Ghu = Gu;
                    gl = ml * itl + Gl * rdl;
                    gu = mu * itu + Gu * rdu;
                }
            }
        } else {
            Ghl = Gl; Ghu = Gu;
            gl = ml * itl + Gl * rdl;
            gu = mu * itu + Gu * rdu;
        }
        Gh = act ? Ghl + Ghu : 0.0;
        gam = act ? gl - gu : 0.0;
    }
    USV_DEV void expand(double w) // w = c' dz
    {
        if constexpr (SOFTROW) {
            dsl = (rhol - Gl * w) * iDl;
            dsu = (rhou + Gu * w) * iDu;
            if constexpr (MIXED) {
                if (!soft) { dsl = 0.0; dsu = 0.0; }
            }
            dtsl = dsl + rdsl;
            dtsu = dsu + rdsu;
            dlsl = -(msl + lsl * dtsl) * itsl;
            dlsu = -(msu + lsu * dtsu) * itsu;
        } else {
            dsl = 0.0; dsu = 0.0;
        }
        dtl = w + dsl + rdl;
        dtu = -w + dsu + rdu;
        dll = -(ml + ll * dtl) * itl;
        dlu = -(mu + lu * dtu) * itu;
    }
    // largest -dv/v over the pairs of this row (the step length is 1/max(1, that))
    USV_DEV double blocking(double q) const
    {
        if (!act) return q;
        q = lanes::vmax(q, -dtl * itl); q = lanes::vmax(q, -dtu * itu);
        double ill, ilu;
        lanes::frcp2(ll, lu, ill, ilu);
        q = lanes::vmax(q, -dll * ill); q = lanes::vmax(q, -dlu * ilu);
        if constexpr (SOFTROW) {
            if (is_soft()) {
                q = lanes::vmax(q, -dtsl * itsl); q = lanes::vmax(q, -dtsu * itsu);
                double ilsl, ilsu;
                lanes::frcp2(lsl, lsu, ilsl, ilsu);
                q = lanes::vmax(q, -dlsl * ilsl); q = lanes::vmax(q, -dlsu * ilsu);
            }
        }
        return q;
    }
    USV_DEV void apply(double a)
    {
        ll += a * dll; lu += a * dlu; tl += a * dtl; tu += a * dtu;
        if constexpr (SOFTROW) {
            sl += a * dsl; su += a * dsu;
            lsl += a * dlsl; lsu += a * dlsu; tsl += a * dtsl; tsu += a * dtsu;
        }
    }
};

// SOFTBOX: some state bounds are soft (acados idxsbx); their rows carry slacks like the soft obstacle rows and
// keep ten planes of their own (no packing).
// The reference's obstacle row h = sqrt((px - ox)^2 + (py - oy)^2) (scripts/usv_guidance_ca1/usv_model.py:133-140,
// scripts/usv_pf_ca/usv_model.py:165-168) and its gradient with respect to the position, from (dx, dy) = pos - centre.
USV_DEV void obs_dist(double dx, double dy, double &d, double &ux, double &uy)
{
    const double d2 = dx * dx + dy * dy;
    const double id = lanes::frsqrt(d2);
    d = d2 * id;
    ux = dx * id; uy = dy * id;
}

// LDSWS: the workspace planes of a row's instance live in LDS for the whole solve (lanes::PlanesLds) - for batches small
// enough that every instance in flight fits (host: usvmpc.hip); the lineariser's planes are copied in at the cold start.
// MERGE (with PACK, no dense rows: host_spec.hpp / usvmpc.hip): the box rows are processed where they are stored - as rows of the
// last obstacle chunk, in its idle lanes - instead of being gathered to their variables' lanes for a row pass of their own:
// one RowCalc pass per sweep instead of two.  The workspace layout is the one of PACK; only the four sweeps differ.
// AUXLDS: the aux plane (WsLayout::P_AUX - dense box rows, position of the linearisation point, r_g, l_u: at most ten values
// per stage) lives in the wave's LDS for the whole launch instead of being streamed with the planes: 4 reads + 2 writes of the
// 59 + 13 plane accesses per stage and IPM iteration go (the kernel streams at the HBM ceiling: profiles/r03_bound_experiment.txt).
// RTI launches whose horizon fits (host: usvmpc.hip); finish() leaves a copy in the HBM plane for the read-back paths.
// WIDE (every row layout: packed box rows beside one or two obstacle chunks - either row-pass form -, box rows in planes of their own - no
// obstacle rows, obstacle rows that leave no idle lanes, soft state bounds -; the solver's planes in LDS, LDSWS, or - horizons that do not
// fit there, the launches of a full SQP - in HBM): the latency mapping - ONE instance per wave.  The four rows of the wave are given the
// same instance and hold the same values; what a lone row spends most of a sweep on, the chains of a stage's box / obstacle rows
// (stage-local: they depend on nothing outside their stage), the rows do for FOUR CONSECUTIVE STAGES at once - row r takes stage
// kb -+ r of a block - and leave each stage's terms (Gamma, gamma, S_xx ...) in an exchange area of the workgroup's LDS; the
// Riccati / forward recursion then runs over the block's stages in all four rows alike (a wave64 instruction costs the same for
// one active row as for four: profiles/r03_exec16_microbench.txt).  Every sum is taken in the order of the 16-lane sweeps, so
// the results equal theirs bit for bit - on the lane emulator and, with the contraction rule below, on the device; only row 0 writes
// results.
// WW (with WIDE): waves per instance.  The row phase scales on: a workgroup of WW waves shares out the row work of 4 WW consecutive stages
// (wave w, row r: stage kb -+ (4 w + r)), every wave runs the recursion over the block; the exchange area, the planes in LDS and the
// parked constants are the workgroup's, phases are separated by workgroup barriers, wave 0 / row 0 writes.  For the single instance and
// the few dozen: a CU (WW = 4) or half a CU (WW = 2) per instance.
// CPC: HPIPM's conditional predictor-corrector built in (option "cond_pred_corr", on in every HPIPM mode acados can select: DESIGN.md
// section 2; QpIpm::solve) - since round 6 in EVERY kernel (the parameter is kept so that a build without it can be measured against: it
// costs the sums of mu(alpha) in forward B and one LDS slot); the option switches the test, not the kernel.
template <class M, int KCH, bool SOFT, bool HDIAG, bool PACK, bool SOFTBOX = false, bool LDSWS = false, bool MERGE = false, bool AUXLDS = false,
          bool WIDE = false, int WW = 1, bool CPC = true>
struct QpIpm {
    static_assert(WW == 1 || (WIDE && (WW == 2 || WW == 4)), "several waves per instance: the wide mapping only");
    static_assert(!WIDE || (HDIAG && !AUXLDS), "the wide mapping works on every row layout of an OCP with a diagonal Hessian");
    static_assert(!MERGE || PACK, "merged row pass works on the packed layout");
    static_assert(!(AUXLDS && LDSWS), "with the whole workspace in LDS the aux plane is there already");
    static_assert(!PACK || KCH > 0, "box rows are packed into obstacle planes");
    static_assert(!(PACK && SOFTBOX), "soft state bounds are not packed");
    static constexpr int NX = M::NX, NU = M::NU, NZ = NX + NU;
    static constexpr int PXL = NU + M::IPX, PYL = NU + M::IPY;
    using MP = MatPack<M>;
    static_assert(NZ <= LANES, "one lane per variable of [u;x]");
    // aux-plane lane map (WsLayout): dense box rows in lanes 0..7, l_u from lane 11 down, r_g from 13 down, position in 14 / 15
    static_assert(NU <= 2, "the aux plane has room for two controls' r_g / l_u entries");
    static_assert(MP::NPK * 16 + 2 <= 256, "slot numbers of the exchange area are kept in one byte");
    // plane map of the per-stage workspace window
    // (P_Z: zbar + z;  P_AUX: dense box rows | position of the linearisation point | r_g | l_u;  P_PB: P_{k+1} b_k;
    // P_PI: pi_k - see WsLayout)
    // (plane numbers: WsLayout, params.hpp - shared with the lineariser, which fills P_RB0, P_GQ and P_MAT..)
    using WL = WsLayout<M, KCH, SOFT, SOFTBOX>;
    enum : int { P_Z = WL::P_Z, P_AUX = WL::P_AUX, P_DZA = WL::P_DZA, P_DZ = WL::P_DZ, P_DX0 = WL::P_DX0, P_PB = WL::P_PB,
                 P_PI = WL::P_PI, P_BLL = WL::P_BLL, P_BLU = WL::P_BLU, P_BTL = WL::P_BTL, P_BTU = WL::P_BTU,
                 P_OBS = WL::P_OBS, P_LZU = WL::P_LZU, P_RB0 = WL::P_RB0, P_GQ = WL::P_GQ, P_MAT = WL::P_MAT,
                 P_BS = WL::P_BS };
    enum : int { AXL_ZX = WL::AXL_ZX, AXL_ZY = WL::AXL_ZY, AXL_RG = WL::AXL_RG, AXL_LU = WL::AXL_LU };
    static constexpr int OBSN = WL::OBSN;
    static constexpr int NPL = WL::NPT;

    static constexpr bool out_unit(int j) { return ((M::OUT_UNIT >> j) & 1u) != 0u; }
    static constexpr unsigned XMASK = (1u << NX) - 1u, ZMASK = (1u << NZ) - 1u;
    static constexpr unsigned NONUNIT = XMASK & ~(unsigned)M::OUT_UNIT; // rows of [B A] that are stored

    // Row-times-broadcast sums, terms in ascending order of the set bits j of MASK, four per instruction group (lanes::fma_bc4):
    //   dot_lanes: acc += sum_j bcast<KOFF + j>(b) * a_of(j)     one source vector, its lanes KOFF + j
    //   dot_col  : acc += sum_j bcast<K>(b_of(j)) * a_of(j)      lane K of a different vector per term
    // a_of / b_of take std::integral_constant<int, j>.
    template <unsigned MASK, class F>
    USV_DEV static void groups_of_four(F f)
    {
        constexpr int n = __builtin_popcount(MASK);
        sfor<0, (n + 3) / 4>([&](auto g) {
            constexpr int i0 = 4 * g, m = (n - i0 < 4) ? n - i0 : 4;
            constexpr int j0 = MP::nth(MASK, i0), j1 = m > 1 ? MP::nth(MASK, i0 + 1) : 0, j2 = m > 2 ? MP::nth(MASK, i0 + 2) : 0,
                          j3 = m > 3 ? MP::nth(MASK, i0 + 3) : 0;
            f(std::integral_constant<int, m>{}, std::integral_constant<int, j0>{}, std::integral_constant<int, j1>{},
              std::integral_constant<int, j2>{}, std::integral_constant<int, j3>{});
        });
    }
    template <unsigned MASK, int KOFF, class FA>
    USV_DEV static void dot_lanes(double &acc, double b, FA a_of)
    {
        groups_of_four<MASK>([&](auto m, auto j0, auto j1, auto j2, auto j3) {
            if constexpr (m == 4) lanes::fma_bc4<KOFF + j0, KOFF + j1, KOFF + j2, KOFF + j3>(acc, b, a_of(j0), b, a_of(j1), b, a_of(j2), b, a_of(j3));
            else if constexpr (m == 3) lanes::fma_bc3<KOFF + j0, KOFF + j1, KOFF + j2>(acc, b, a_of(j0), b, a_of(j1), b, a_of(j2));
            else if constexpr (m == 2) lanes::fma_bc2<KOFF + j0, KOFF + j1>(acc, b, a_of(j0), b, a_of(j1));
            else lanes::fma_bc<KOFF + j0>(acc, b, a_of(j0));
        });
    }
    template <unsigned MASK, int K, class FB, class FA>
    USV_DEV static void dot_col(double &acc, FB b_of, FA a_of)
    {
        groups_of_four<MASK>([&](auto m, auto j0, auto j1, auto j2, auto j3) {
            if constexpr (m == 4) lanes::fma_bc4<K, K, K, K>(acc, b_of(j0), a_of(j0), b_of(j1), a_of(j1), b_of(j2), a_of(j2), b_of(j3), a_of(j3));
            else if constexpr (m == 3) lanes::fma_bc3<K, K, K>(acc, b_of(j0), a_of(j0), b_of(j1), a_of(j1), b_of(j2), a_of(j2));
            else if constexpr (m == 2) lanes::fma_bc2<K, K>(acc, b_of(j0), a_of(j0), b_of(j1), a_of(j1));
            else lanes::fma_bc<K>(acc, b_of(j0), a_of(j0));
        });
    }

    using BoxRow = RowCalc<SOFTBOX, SOFTBOX>;
    using ObsRow = RowCalc<SOFT, SOFT && MERGE>; // (merged: the chunk's slot lanes carry hard box rows next to soft obstacle rows)
    // WIDE: the LDS holds what the solve writes - P_Z .. P_PI, the row planes, L_zu; the four box planes the packed layouts never
    // touch are squeezed out and the lineariser's planes (P_RB0, P_GQ, P_MAT..: read once per sweep) stay in HBM, so that the
    // horizon of an instance takes half the LDS and twice as many waves are resident.
    struct WideMap {
        static constexpr int at(int plane)
        {
            // (box rows in planes of their own - no obstacle rows, or soft state bounds, whose six slack planes sit behind the lineariser's)
            if (!PACK) return plane < WL::P_RB0 ? plane : ((SOFTBOX && plane >= WL::P_BS && plane < WL::P_BS + 6) ? WL::P_RB0 + (plane - WL::P_BS) : -1);
            return plane < WL::P_BLL ? plane : (plane >= WL::P_OBS && plane < WL::P_RB0 ? plane - 4 : -1);
        }
    };
    static constexpr int NPLW = (WIDE && LDSWS) ? WL::P_RB0 - (PACK ? 4 : 0) + (SOFTBOX ? 6 : 0) : WL::NPT; // planes per stage of an LDS region
    using Planes = std::conditional_t<WIDE && LDSWS, lanes::PlanesLdsMapped<WideMap>, std::conditional_t<LDSWS, lanes::PlanesLds, lanes::Planes>>;

    const DevPtrs &P;
    const DevSpec &S;
    double rbscale; // product of (1 - alpha) over the steps taken: scales the dynamics residual
    // Copies of the DevSpec scalars the sweeps test per stage.  DevSpec lives in global memory and the compiler
    // must assume the plane stores alias it, so every `S.field` inside a sweep is re-loaded with a VECTOR load
    // followed by s_waitcnt vmcnt(0) - which drains the whole prefetch queue of the stage.
    int Kn, nB, itmax; // S.K, S.B, S.iter_max
    bool pstat;     // S.p_static
    bool keep;      // full SQP: this instance is finished, its workspace (multipliers of the last QP) must survive
    int lane, N;
    // per row (the same in its 16 lanes): the group it works on, that group's instance, the lane's offset in a stage window
    long g, b;
    unsigned voff;
    unsigned loff;   // LDSWS: this lane's entry of (stage 0, plane 0) of its row's LDS region, in doubles
    bool live;       // the row owns a workspace (LDSWS: surplus rows of a wave share row 0's region read-only)
    // AUXLDS: this lane's entry of stage 0 in the wave's aux area [stage][slot][row] (doubles), the stage stride, and whether the
    // lane holds anything (slots: dense box values 0 .. 4 nd - 1 | zx | zy | r_g (nu) | l_u (nu))
    unsigned auxoff;
    int auxstride;
    bool auxlive;
    long stage_stride;     // doubles between consecutive stages of the workspace: Bp * NPL * 16
    unsigned stage_bytes;  // bytes of one stage's window
    bool xlane, ulane, valid, isPX, isPY;
    // per-lane constants, read once: box bounds of this lane's variable, Hessian diagonal
    // PACK: the box rows' (lambda_l, lambda_u, t_l, t_u) do not get four planes of their own.  A *slot* row
    // lives in an idle lane (>= K_last) of the last obstacle chunk's four (lambda, t) planes; rows that do not
    // fit there are *dense*: their four values sit in four consecutive lanes of the aux plane (P_AUX).  Slot lanes
    // and dense lanes are disjoint (host_spec.hpp), so one run-time gather per value serves both kinds.
    //   bsrc   : lane this variable's value 0 comes from (slot lane, or first of the four dense lanes)
    //   bstep  : 0 for a slot row, 1 for a dense row (value e comes from lane bsrc + e*bstep)
    //   ssrc   : variable whose row this lane stores (as slot lane or as dense lane)
    //   isslot / isdense : what this lane stores
    bool ounit; // this lane is the state of a structurally unit row of [A B] (M::OUT_UNIT)
    // The packed stage matrix goes through a wave-private LDS exchange area (lanes::Xpose): the lanes put the planes as they
    // were loaded, and every lane reads back the slots it needs - row form (lane r: entry (j, r) of every stored row j) for
    // the backward sweeps, column form (lane nu+j: entry (j, c) of every stored column c) for the forward product.  Slot
    // numbers per lane, one byte each; structural zeros / unit diagonals read the constant slots ZSLOT / OSLOT.
    static constexpr unsigned CMASK = MP::col_mask();
    static constexpr int NCOL = __builtin_popcount(CMASK);
    static constexpr int ZSLOT = MP::NPK * 16, OSLOT = ZSLOT + 1;
    using XP = lanes::Xpose<MP::NPK * 16 + 2, (WIDE ? WW : 4), WIDE>; // (WIDE: the rows of a wave hold the same matrix - one copy per wave)
    unsigned rowtab[(NX + 3) / 4], coltab[(NCOL + 3) / 4];
    bool selfone; // this lane's state has the exact unit diagonal and its own column is not among the stored ones
    bool isslot, isdense, anydense;
    bool slot_u; // MERGE: the box row this slot lane carries belongs to a control (active at stages 0..N-1; a state's: 1..N-1)
    int bsrc, bstep, ssrc;
    bool hasb;
    // These per-lane constants live in LDS (lanes::Stash), not in registers: see there.
    static constexpr int KC = KCH > 0 ? KCH : 1;
    enum : int {
        ST_LB = 0, ST_UB, ST_HDS, ST_HDT, ST_UH, ST_OX = ST_UH + KC, ST_OY = ST_OX + KC, ST_LH = ST_OY + KC, ST_SOFT = ST_LH + KC,
        ST_ZL = ST_SOFT, ST_ZU = ST_ZL + (SOFT ? KC : 0), ST_QL = ST_ZU + (SOFT ? KC : 0), ST_QU = ST_QL + (SOFT ? KC : 0),
        ST_BSL = ST_QU + (SOFT ? KC : 0), ST_BSU = ST_BSL + (SOFT ? KC : 0), ST_BOX = ST_BSU + (SOFT ? KC : 0),
        ST_SLOT = ST_BOX + (SOFTBOX ? 6 : 0), // MERGE: bounds of the box row a slot lane carries
        ST_CPC = ST_SLOT + (MERGE ? 2 : 0),   // option "cond_pred_corr": second-order factor of this pass's / of the pending step's corrector targets
        ST_N = ST_CPC + (CPC ? 1 : 0)         // (0 or 1 each: the two share ONE slot - with the aux plane in LDS a wave at N = 40 has 600 bytes left of its 20 KB)
    };
    using ST = lanes::Stash<ST_N, (WIDE ? 16 : 64)>; // (WIDE: the rows hold the same constants - one row's worth)
    struct CRef {
        int slot;
        USV_DEV operator double() const { return ST::get(slot); }
        USV_DEV void operator=(double v) const { ST::put(slot, v); }
    };
    template <int SLOT0>
    struct CArr {
        USV_DEV CRef operator[](int c) const { return CRef{SLOT0 + c}; }
    };
    template <int SLOT>
    struct CVal {
        USV_DEV operator double() const { return ST::get(SLOT); }
        USV_DEV void operator=(double v) const { ST::put(SLOT, v); }
    };
    template <int SLOT, int HALF>
    struct CHalf {
        USV_DEV operator double() const { return ST::geth(SLOT, HALF); }
        USV_DEV void operator=(double v) const { ST::puth(SLOT, HALF, v); }
    };
    CVal<ST_LB> lbv; CVal<ST_UB> ubv; CVal<ST_HDS> hd_stage; CVal<ST_HDT> hd_term;
    CVal<ST_SLOT> slot_lb; CVal<ST_SLOT + 1> slot_ub;
    CHalf<ST_CPC, 0> so_cur; CHalf<ST_CPC, 1> so_prv; // (parked in LDS like the other per-lane constants: they must not cost the sweeps a register)
    bool bsoft;                               // SOFTBOX: this lane's state bound is soft
    // its slack penalties (scaled by dt) and slack lower bounds
    CVal<ST_BOX + 0> bzl; CVal<ST_BOX + 1> bzu; CVal<ST_BOX + 2> bZl; CVal<ST_BOX + 3> bZu; CVal<ST_BOX + 4> bbsl; CVal<ST_BOX + 5> bbsu;
    CArr<ST_ZL> c_zl; CArr<ST_ZU> c_zu; CArr<ST_QL> c_Zl; CArr<ST_QU> c_Zu; CArr<ST_BSL> c_bsl; CArr<ST_BSU> c_bsu;
    // obstacle data of this lane's row(s): upper bound, and - when the obstacle set is the same on every
    // stage (DevSpec::p_static, what the reference's callers do) - centre and lower bound
    CArr<ST_UH> c_uh; CArr<ST_OX> c_ox; CArr<ST_OY> c_oy; CArr<ST_LH> c_lh;

    // lds_row: which of the workgroup's LDS regions the row owns (LDSWS), < 0: none
    USV_DEV QpIpm(const DevPtrs &P_, long g_, int lds_row = 0) : P(P_), S(*P_.spec)
    {
        lane = lanes::lane();
        N = lanes::uniform(S.N);
        Kn = lanes::uniform(S.K);
        pstat = lanes::uniform(S.p_static) != 0;
        nB = lanes::uniform(S.B);
        itmax = lanes::uniform(S.iter_max);
        {   // workspace layout: [stage][group][plane][16 lanes]
            const long nbp = (long)lanes::uniform(S.Bp);
            stage_stride = nbp * NPL * LANES;
            stage_bytes = (unsigned)(nbp * NPL * 128);
        }
        ulane = lane < NU;
        xlane = lane >= NU && lane < NZ;
        valid = lane < NZ;
        isPX = KCH > 0 && lane == PXL;
        isPY = KCH > 0 && lane == PYL;
        hasb = S.has_b[lane] != 0;
        ounit = xlane && ((M::OUT_UNIT >> (xlane ? lane - NU : 0)) & 1u) != 0u;
        {
            sfor<0, (NX + 3) / 4>([&](auto w) { rowtab[w] = 0u; });
            sfor<0, (NCOL + 3) / 4>([&](auto w) { coltab[w] = 0u; });
            sfor<0, NX>([&](auto j) {
                constexpr unsigned m = MP::row_mask(j);
                const unsigned slot = ((m >> lane) & 1u) ? (unsigned)(MP::start(j) + __builtin_popcount(m & ((1u << lane) - 1u)))
                                                         : ((MP::diag_one(j) && lane == NU + j) ? (unsigned)OSLOT : (unsigned)ZSLOT);
                rowtab[j / 4] |= slot << (8 * (j % 4));
            });
            sfor<0, NCOL>([&](auto ci) {
                constexpr int c = MP::nth(CMASK, ci);
                unsigned slot = (unsigned)ZSLOT;
                sfor<0, NX>([&](auto j) { // the row this lane owns (if it is a state lane)
                    constexpr unsigned m = MP::row_mask(j);
                    if (lane == NU + j) {
                        if constexpr (((m >> c) & 1u) != 0u) slot = (unsigned)(MP::start(j) + MP::rank(m, c));
                        else if constexpr (MP::diag_one(j) && c == NU + j) slot = (unsigned)OSLOT;
                    }
                });
                coltab[ci / 4] |= slot << (8 * (ci % 4));
            });
            bool so = false;
            sfor<0, NX>([&](auto j) {
                if constexpr (MP::diag_one(j) && ((CMASK >> (NU + j)) & 1u) == 0u) so = (lane == NU + j) ? true : so;
            });
            selfone = so;
            XP::put(ZSLOT, 0.0);
            XP::put(OSLOT, 1.0);
        }
        isslot = PACK && S.slot_is[lane] == 1;
        isdense = PACK && S.slot_is[lane] == 2;
        anydense = PACK && lanes::uniform(S.box_dense) != 0; // wave-uniform
        bsrc = S.box_slot[lane];
        bstep = S.box_step[lane];
        ssrc = S.slot_var[lane];
        slot_u = isslot && ssrc < NU;
        if constexpr (MERGE) {
            slot_lb = S.lb[isslot ? ssrc : 0];
            slot_ub = S.ub[isslot ? ssrc : 0];
        }
        lbv = S.lb[lane];
        ubv = S.ub[lane];
        hd_stage = S.Hc[lane * LANES + lane];
        hd_term = S.He[lane * LANES + lane];
        bsoft = SOFTBOX && S.bsoft[lane] != 0;
        if constexpr (SOFTBOX) { // (their stash slots exist only then)
            bzl = S.b_zl[lane]; bzu = S.b_zu[lane]; bZl = bsoft ? S.b_Zl[lane] : 1.0; bZu = bsoft ? S.b_Zu[lane] : 1.0;
            bbsl = S.b_lsl[lane]; bbsu = S.b_lsu[lane];
        }
        if constexpr (KCH > 0) {
            sfor<0, KCH>([&](auto c) {
                const int i = c * LANES + lane;
                c_uh[c] = S.uh[i < S.K ? i : 0];
            });
        }
        auxoff = 0; auxstride = 0; auxlive = false;
        if constexpr (AUXLDS) {
            const int nd4 = lanes::uniform(S.aux_dense4);
            constexpr int ZB = KCH > 0 ? 2 : 0;
            int slot = lane < nd4 ? lane : -1;
            if constexpr (KCH > 0) slot = lane == AXL_ZX ? nd4 : (lane == AXL_ZY ? nd4 + 1 : slot);
            sfor<0, NU>([&](auto l) {
                slot = lane == AXL_RG - l ? nd4 + ZB + l : slot;
                slot = lane == AXL_LU - l ? nd4 + ZB + NU + l : slot;
            });
            auxlive = slot >= 0;
            auxstride = (nd4 + ZB + 2 * NU) * lanes::WAVE_ROWS;
            auxoff = (unsigned)((slot > 0 ? slot : 0) * lanes::WAVE_ROWS) + lanes::wave_row();
        }
        g = 0; b = 0;
        live = lds_row >= 0;
        loff = (unsigned)((lds_row > 0 ? lds_row : 0) * (N + 1) * NPLW * LANES + lane);
        if constexpr (KCH > 0) sfor<0, KCH>([&](auto c) { c_ox[c] = 0.0; c_oy[c] = 0.0; c_lh[c] = 0.0; });
        bind(g_, true);
        if constexpr (KCH > 0 && SOFT) {
            sfor<0, KCH>([&](auto c) {
                const int i = c * LANES + lane;
                const int ii = i < S.K ? i : 0;
                c_zl[c] = S.zl[ii]; c_zu[c] = S.zu[ii];
                c_Zl[c] = i < S.K ? S.Zl[ii] : 1.0; c_Zu[c] = i < S.K ? S.Zu[ii] : 1.0;
                c_bsl[c] = S.lsl[ii]; c_bsu[c] = S.lsu[ii];
            });
        }
    }

    // Point the rows selected by `sel` at group gn: instance, workspace offset, per-instance constants.  Groups beyond the
    // batch (padding of the last wave) replay the last instance and never write results.
    USV_DEV void bind(long gn, bool sel)
    {
        g = sel ? gn : g;
        const long gi = g < nB ? g : (long)nB - 1;
        const long bn = P.perm ? (long)P.perm[gi] : gi;
        b = sel ? bn : b;
        voff = sel ? lanes::Planes::lane_offset(g, NPL, lane) : voff; // (a parked row - solve() - stays parked)
        if constexpr (KCH > 0) {
            sfor<0, KCH>([&](auto c) {
                const int i = c * LANES + lane;
                const int ii = i < Kn ? i : 0;
                const double ox = P.p[(long)b * (N + 1) * 2 * Kn + 2 * ii], oy = P.p[(long)b * (N + 1) * 2 * Kn + 2 * ii + 1];
                const double lh = P.lh[(long)b * N * Kn + ii];
                c_ox[c] = sel ? ox : c_ox[c]; c_oy[c] = sel ? oy : c_oy[c]; c_lh[c] = sel ? lh : c_lh[c];
            });
        }
    }

    // the row's planes of stage k in HBM (where the lineariser writes; the whole workspace unless LDSWS)
    USV_DEV lanes::Planes wsg(int k) const { return lanes::Planes(P.ws + (long)k * stage_stride, stage_bytes, voff); }
    USV_DEV Planes ws(int k) const
    {
        if constexpr (LDSWS) return Planes(loff + (unsigned)(k * NPLW * LANES), live);
        else return wsg(k);
    }
    // the aux plane of stage k: from / to the wave's LDS area (AUXLDS) or the workspace plane
    USV_DEV double aux_ld(int k, const Planes &W) const
    {
        if constexpr (AUXLDS) return lanes::dyn_lds()[auxoff + (unsigned)(k * auxstride)];
        else return W.ld(P_AUX);
    }
    USV_DEV void aux_st(int k, const Planes &W, double v) const
    {
        if constexpr (AUXLDS) { if (auxlive) lanes::dyn_lds()[auxoff + (unsigned)(k * auxstride)] = v; }
        else W.st(P_AUX, v);
    }
    // iterate value of this lane's variable at stage k (caller-visible arrays)
    USV_DEV double zbar(int k) const
    {
        if (ulane) return (k < N) ? P.u[((long)b * N + k) * NU + lane] : 0.0;
        if (xlane) return P.x[((long)b * (N + 1) + k) * NX + (lane - NU)];
        return 0.0;
    }

    // (the row value of a box row is the absolute iterate zbar + z, its bounds are the caller's lb / ub)
    USV_DEV void box_data(int k, BoxRow &r) const
    {
        const bool stage_ok = ulane ? (k < N) : (k >= 1 && k < N);
        r.act = valid && hasb && stage_ok;
        r.dl = r.act ? lbv : -1.0;
        r.du = r.act ? ubv : 1.0;
        if constexpr (SOFTBOX) {
            r.soft = r.act && bsoft;
            r.zl = bzl; r.zu = bzu; r.Zl = bZl; r.Zu = bZu; r.bsl = bbsl; r.bsu = bbsu;
        }
    }
    template <class PL>
    USV_DEV static void box_store(const PL &W, const BoxRow &r)
    {
        W.st(P_BLL, r.ll); W.st(P_BLU, r.lu); W.st(P_BTL, r.tl); W.st(P_BTU, r.tu);
        if constexpr (SOFTBOX) {
            W.st(P_BS, r.sl); W.st(P_BS + 1, r.su); W.st(P_BS + 2, r.lsl); W.st(P_BS + 3, r.lsu);
            W.st(P_BS + 4, r.tsl); W.st(P_BS + 5, r.tsu);
        }
    }
    // Obstacle chunk c at stage k.  The row h_i = |pos - o_i| >= lh_i is linearised here, from the iterate's
    // position (the px / py lanes of zb) and the obstacle data, instead of being streamed from planes written
    // by the lineariser: a square root and a few multiplies replace four plane reads per sweep.
    // raw = (ox, oy, lh) of this lane's obstacle at this stage; (zbx, zby) = position of the linearisation point.
    template <int C>
    USV_DEV void obs_geom(int k, double zbx, double zby, const double *raw, ObsRow &r, double &cx, double &cy) const
    {
        const int i = C * LANES + lane;
        const bool stage_ok = (k >= 1 && k < N); // wave-uniform
        r.act = stage_ok && i < Kn;
        const double dx = zbx - raw[0], dy = zby - raw[1];
        double d, ux, uy;
        obs_dist(dx, dy, d, ux, uy);
        cx = r.act ? ux : 0.0; cy = r.act ? uy : 0.0;
        r.dl = r.act ? raw[2] - d : -1.0; r.du = r.act ? c_uh[C] - d : 1.0;
        if constexpr (SOFT) {
            r.zl = c_zl[C]; r.zu = c_zu[C]; r.Zl = c_Zl[C]; r.Zu = c_Zu[C]; r.bsl = c_bsl[C]; r.bsu = c_bsu[C];
        }
    }
    // (ox, oy, lh) of this lane's obstacle in chunk c at stage k, from the caller-visible arrays
    template <int C>
    USV_DEV void obs_raw(int k, double *raw) const
    {
        if (pstat) { // wave-uniform
            raw[0] = c_ox[C]; raw[1] = c_oy[C]; raw[2] = c_lh[C];
        } else {
            const int i = C * LANES + lane;
            const int ii = i < Kn ? i : 0;
            const int kl = k < N ? k : N - 1;
            const double *pk = P.p + ((long)b * (N + 1) + k) * 2 * Kn;
            raw[0] = pk[2 * ii]; raw[1] = pk[2 * ii + 1];
            raw[2] = P.lh[((long)b * N + kl) * Kn + ii];
        }
    }
    // pk: the box rows gathered to their storage lanes (PACK, last chunk), else nullptr
    template <class PL>
    USV_DEV void obs_store(const PL &W, int c, const ObsRow &r, const double *pk = nullptr) const
    {
        const int p0 = P_OBS + c * OBSN;
        const bool sel = PACK && pk != nullptr && isslot;
        W.st(p0, sel ? pk[0] : r.ll); W.st(p0 + 1, sel ? pk[1] : r.lu);
        W.st(p0 + 2, sel ? pk[2] : r.tl); W.st(p0 + 3, sel ? pk[3] : r.tu);
        if constexpr (SOFT) {
            W.st(p0 + 4, r.sl); W.st(p0 + 5, r.su); W.st(p0 + 6, r.lsl); W.st(p0 + 7, r.lsu);
            W.st(p0 + 8, r.tsl); W.st(p0 + 9, r.tsu);
        }
    }
    // box row values delivered to the lanes that store them (call under wave-uniform control flow); returns what this
    // lane contributes to the dense part of the aux plane: value (lane & 3) of the row the lane belongs to
    USV_DEV double box_pack(const BoxRow &r, double *pk) const
    {
        pk[0] = lanes::gather(r.ll, ssrc); pk[1] = lanes::gather(r.lu, ssrc);
        pk[2] = lanes::gather(r.tl, ssrc); pk[3] = lanes::gather(r.tu, ssrc);
        const int e = lane & 3;
        return e == 0 ? pk[0] : (e == 1 ? pk[1] : (e == 2 ? pk[2] : pk[3]));
    }
    // ---- the aux plane (WsLayout): composition and access
    USV_DEV static double aux_zx(double aux) { return KCH > 0 ? lanes::bcast<AXL_ZX>(aux) : 0.0; }
    USV_DEV static double aux_zy(double aux) { return KCH > 0 ? lanes::bcast<AXL_ZY>(aux) : 0.0; }
    // this lane's entry of a u-lane vector kept in the aux plane from lane TOP downwards (r_g, l_u)
    template <int TOP>
    USV_DEV double aux_ulane(double aux) const
    {
        double r = 0.0;
        sfor<0, NU>([&](auto l) {
            const double v = lanes::bcast<TOP - l>(aux);
            r = (lane == l) ? v : r;
        });
        return r;
    }
    // dense: this lane's dense box value (used on the dense lanes only); rg, luv: u-lane vectors
    USV_DEV double aux_compose(double dense, double zbx, double zby, double rg, double luv) const
    {
        double a = isdense ? dense : 0.0;
        sfor<0, NU>([&](auto l) {
            const double r_l = lanes::bcast<l>(rg), u_l = lanes::bcast<l>(luv);
            a = (lane == AXL_RG - l) ? r_l : a;
            a = (lane == AXL_LU - l) ? u_l : a;
        });
        if constexpr (KCH > 0) a = (lane == AXL_ZX) ? zbx : ((lane == AXL_ZY) ? zby : a);
        return a;
    }
    // the position lanes' share of the linearisation point: zfull - this = the step z there (obstacle rows)
    USV_DEV double pos_sel(double zbx, double zby) const { return isPX ? zbx : (isPY ? zby : 0.0); }
    // Row j of [B A]' for every stored row (bat[j]: lane r = d x+_j / d z_r) from the packed planes of stage k.
    // mat_issue puts the plane loads in flight, mat_unpack (call under wave-uniform control flow) distributes.
    USV_DEV void mat_issue(int k, double *pk) const
    {
        const Planes W = ws(k);
        sfor<0, MP::NPK>([&](auto q) { pk[q] = W.ld(P_MAT + q); });
    }
    // (call under wave-uniform control flow)
    USV_DEV void mat_put(const double *pk) const
    {
        sfor<0, MP::NPK>([&](auto q) { XP::put(16 * q + lane, pk[q]); });
        XP::sync();
    }
    USV_DEV void mat_unpack(const double *pk, double *bat) const
    {
        mat_put(pk);
        sfor<0, NX>([&](auto j) {
            if constexpr (out_unit(j)) bat[j] = 0.0; // never read: unit rows are handled structurally
            else bat[j] = XP::get((int)((rowtab[j / 4] >> (8 * (j % 4))) & 0xffu));
        });
    }
    // x+ = [B A] dz (+ acc) on the state lanes: lane nu+j reads row j of the matrix column by column
    USV_DEV double mat_apply(const double *pk, double dz, double acc) const
    {
        mat_put(pk);
        double v[NCOL];
        sfor<0, NCOL>([&](auto ci) { v[ci] = XP::get((int)((coltab[ci / 4] >> (8 * (ci % 4))) & 0xffu)); });
        dot_lanes<CMASK, 0>(acc, dz, [&](auto c) { return v[__builtin_popcount(CMASK & ((1u << c) - 1u))]; });
        return acc + (selfone ? dz : 0.0);
    }
    USV_DEV static double obs_dot(double cx, double cy, double vec)
    {
        return cx * lanes::bcast<PXL>(vec) + cy * lanes::bcast<PYL>(vec);
    }

    // ------------------------------------------------------------------ cold start
    // (an instance that a full SQP has frozen keeps the multipliers of its last QP: keep)
    // sel: rows to cold-start.  Returns whether x0 violates a HARD obstacle row of stage 0: acados applies the nh rows at
    // stages 0..N-1; at stage 0 they depend on no free variable (D = 0, x_0 is pinned to x0) and are not rows of the QP
    // solved here, but a violated hard one makes acados' QP infeasible - status 4, iterate untouched.
    USV_DEV bool init(bool sel)
    {
        const bool wr = sel && !keep && (!(WIDE && !LDSWS) || live); // (WIDE over HBM planes: the four rows would store the same values)
        double bad0 = 0.0;
        for (int k = 0; k <= N; k++) {
            const Planes W = ws(k);
            const double zb = zbar(k);
            if constexpr (LDSWS && !WIDE) { // the linearisation of this stage comes in from HBM
                const lanes::Planes G = wsg(k);
                const double gq = G.ld(P_GQ), rb = (k < N) ? G.ld(P_RB0) : 0.0;
                double mpk[MP::NPK];
                if (k < N) sfor<0, MP::NPK>([&](auto q) { mpk[q] = G.ld(P_MAT + q); });
                if (wr) {
                    W.st(P_GQ, gq);
                    if (k < N) {
                        W.st(P_RB0, rb);
                        sfor<0, MP::NPK>([&](auto q) { W.st(P_MAT + q, mpk[q]); });
                    }
                }
            }
            if (wr) {
                W.st(P_Z, zb); // z = 0
                if (k == 0) W.st(P_DX0, xlane ? P.x0[(long)b * NX + (lane - NU)] : 0.0);
            }
            BoxRow r;
            r.neutral();
            box_data(k, r);
            const double v0 = r.act ? zb : 0.0;
            r.tl = fmax(v0 - r.dl, S.thr0); r.tu = fmax(r.du - v0, S.thr0);
            r.ll = S.mu0 / r.tl; r.lu = S.mu0 / r.tu;
            if constexpr (SOFTBOX) {
                if (r.soft) {
                    r.tsl = fmax(0.0 - r.bsl, S.thr0); r.tsu = fmax(0.0 - r.bsu, S.thr0);
                    r.lsl = S.mu0 / r.tsl; r.lsu = S.mu0 / r.tsu;
                }
            }
            double pk[4], dv = 0.0;
            if constexpr (PACK) dv = box_pack(r, pk);
            else if (wr) box_store(W, r);
            const double zbx = KCH > 0 ? lanes::bcast<PXL>(zb) : 0.0, zby = KCH > 0 ? lanes::bcast<PYL>(zb) : 0.0;
            const double aux = aux_compose(dv, zbx, zby, 0.0, 0.0);
            if (wr) aux_st(k, W, aux);
            if constexpr (KCH > 0) {
                sfor<0, KCH>([&](auto c) {
                    ObsRow o;
                    double cx, cy, raw[3];
                    o.neutral();
                    obs_raw<c>(k, raw);
                    obs_geom<c>(k, zbx, zby, raw, o, cx, cy);
                    o.tl = fmax(0.0 - o.dl, S.thr0); o.tu = fmax(o.du - 0.0, S.thr0);
                    o.ll = S.mu0 / o.tl; o.lu = S.mu0 / o.tu;
                    if constexpr (SOFT) {
                        o.tsl = fmax(0.0 - o.bsl, S.thr0); o.tsu = fmax(0.0 - o.bsu, S.thr0);
                        o.lsl = S.mu0 / o.tsl; o.lsu = S.mu0 / o.tsu;
                    }
                    if (wr) obs_store(W, c, o, (PACK && c == KCH - 1) ? pk : nullptr);
                    if constexpr (!SOFT) {
                        if (k == 0) { // wave-uniform
                            const double e0 = xlane ? P.x0[(long)b * NX + (lane - NU)] - zb : 0.0; // x0 - xbar_0
                            double d, ux, uy;
                            obs_dist(zbx - raw[0], zby - raw[1], d, ux, uy);
                            const double v0 = ux * lanes::bcast<PXL>(e0) + uy * lanes::bcast<PYL>(e0);
                            const bool row = c * LANES + lane < Kn;
                            const double tol = S.tol_ineq;
                            bad0 = (row && (raw[2] - d - v0 > tol || d + v0 - c_uh[c] > tol)) ? 1.0 : bad0;
                        }
                    }
                });
            }
        }
        return lanes::gmax(bad0) > 0.5;
    }

    struct Norms { double rg, rb, rd, rm, musum, nan; };

    // row chain up to the elimination; corr: predictor step (from w_aff) then corrector targets
    template <bool CP = false, class R>
    USV_DEV static void chain(R &r, double v, bool corr, double w_aff, double sigmu, double &Gh, double &gam, double so = 1.0)
    {
        r.resid(v);
        r.targets_pred();
        r.reduce(Gh, gam);
        if (corr) {
            r.expand(w_aff);
            r.template targets_corr<CP>(sigmu, so);
            r.reduce(Gh, gam);
        }
    }

    // ------------------------------------------------------------------ staged plane reads
    // Raw plane values of one stage.  The sweeps read them one stage AHEAD of their use (software
    // prefetch): with 2-3 waves per SIMD there is no other wave to hide an HBM round trip, so each
    // stage's loads are put in flight while the previous stage is still being computed.
    struct StageIn {
        double z, aux, rb, dz, dza, gq, pb; // z: the absolute iterate zbar + z
        double lzu[NU];
        double box[4];
        double bxs[SOFTBOX ? 6 : 1]; // soft state bounds: sl, su, lsl, lsu, tsl, tsu of this lane's box row
        double obs[KCH > 0 ? KCH : 1][OBSN];
        double raw[KCH > 0 ? KCH : 1][3];
    };
    enum : int { SW_BACK_A = 0, SW_FWD_A = 1, SW_BACK_B = 2, SW_FWD_B = 3 };

    template <int SW>
    USV_DEV void load_in(int k, StageIn &in) const
    {
        const Planes W = ws(k);
        in.z = W.ld(P_Z);
        in.aux = aux_ld(k, W);
        // b_k of the current iterate = rbscale * (residual of the linearisation point): the forward sweeps
        // enforce the linearised dynamics, so every step scales it by (1 - alpha) and it is never rewritten
        // (raw value here: scaling it in place would make the prefetch wait for its own load)
        if constexpr (SW != SW_BACK_B) in.rb = (k < N) ? W.ld(P_RB0) : 0.0;
        if constexpr (SW == SW_BACK_A) {
            in.dz = W.ld(P_DZ);
            in.dza = W.ld(P_DZA);
            in.gq = W.ld(P_GQ);
        }
        if constexpr (SW == SW_BACK_B || SW == SW_FWD_B) in.dza = W.ld(P_DZA);
        if constexpr (SW == SW_BACK_B) in.pb = (k < N) ? W.ld(P_PB) : 0.0;
        if constexpr (SW != SW_BACK_A) {
            if (k < N) sfor<0, NU>([&](auto l) { in.lzu[l] = W.ld(P_LZU + l); });
            else sfor<0, NU>([&](auto l) { in.lzu[l] = 1.0; });
        }
        if constexpr (!PACK) {
            in.box[0] = W.ld(P_BLL); in.box[1] = W.ld(P_BLU); in.box[2] = W.ld(P_BTL); in.box[3] = W.ld(P_BTU);
            if constexpr (SOFTBOX) sfor<0, 6>([&](auto e) { in.bxs[e] = W.ld(P_BS + e); });
        }
        if constexpr (KCH > 0) {
            if (k >= 1 && k < N) { // wave-uniform
                sfor<0, KCH>([&](auto c) {
                    sfor<0, OBSN>([&](auto e) { in.obs[c][e] = W.ld(P_OBS + c * OBSN + e); });
                    if (!pstat) obs_raw<c>(k, in.raw[c]);
                });
            } else if (PACK && k == 0) { // the input bounds of stage 0 live there too
                sfor<0, 4>([&](auto e) { in.obs[KCH - 1][e] = W.ld(P_OBS + (KCH - 1) * OBSN + e); });
            }
        }
    }
    USV_DEV void box_from(const StageIn &in, int k, BoxRow &r) const
    {
        r.neutral();
        box_data(k, r);
        double b0, b1, b2, b3;
        if constexpr (PACK) {
            constexpr int CL = KCH > 0 ? KCH - 1 : 0;
            // every lane offers what a reader would want from it: a slot lane its obstacle-plane values, any
            // other lane the aux plane's value (its low lanes hold the dense rows)
            b0 = lanes::gather(isslot ? in.obs[CL][0] : in.aux, bsrc);
            b1 = lanes::gather(isslot ? in.obs[CL][1] : in.aux, bsrc + bstep);
            b2 = lanes::gather(isslot ? in.obs[CL][2] : in.aux, bsrc + 2 * bstep);
            b3 = lanes::gather(isslot ? in.obs[CL][3] : in.aux, bsrc + 3 * bstep);
        } else {
            b0 = in.box[0]; b1 = in.box[1]; b2 = in.box[2]; b3 = in.box[3];
        }
        r.ll = r.act ? b0 : 0.0; r.lu = r.act ? b1 : 0.0;
        r.tl = r.act ? b2 : 1.0; r.tu = r.act ? b3 : 1.0;
        if constexpr (SOFTBOX) {
            r.sl = r.soft ? in.bxs[0] : 0.0; r.su = r.soft ? in.bxs[1] : 0.0;
            r.lsl = r.soft ? in.bxs[2] : 0.0; r.lsu = r.soft ? in.bxs[3] : 0.0;
            r.tsl = r.soft ? in.bxs[4] : 1.0; r.tsu = r.soft ? in.bxs[5] : 1.0;
        }
    }
    // SLOTROWS = false: obstacle rows only, whatever MERGE says (nlp_residual keeps the two-pass form)
    template <int C, bool SLOTROWS = true>
    USV_DEV void obs_from(const StageIn &in, int k, double zbx, double zby, ObsRow &r, double &cx, double &cy) const
    {
        r.neutral();
        if (WIDE || (k >= 1 && k < N)) { // wave-uniform (WIDE: k differs from row to row; obs_geom yields the inactive row of the other branch)
            // with a stage-independent obstacle set the data sits in per-lane constants: no copy in the prefetch
            const double cst[3] = {c_ox[C], c_oy[C], c_lh[C]};
            obs_geom<C>(k, zbx, zby, pstat ? cst : in.raw[C], r, cx, cy);
        } else {
            r.act = false; cx = 0.0; cy = 0.0;
            if constexpr (SOFT) {
                r.zl = c_zl[C]; r.zu = c_zu[C]; r.Zl = c_Zl[C]; r.Zu = c_Zu[C]; r.bsl = c_bsl[C]; r.bsu = c_bsu[C];
            }
        }
        bool softrow = r.act; // lanes whose slack values are live
        if constexpr (MERGE) {
            if constexpr (SOFT) r.soft = r.act; // obstacle rows soft, box rows hard
            if constexpr (C == KCH - 1 && SLOTROWS) {
                // the box row riding in this slot lane: a row like the others, with the variable's bounds (its value and its
                // step come by gather: rowdot)
                const bool sact = isslot && (slot_u ? k < N : (k >= 1 && k < N));
                r.dl = sact ? (double)slot_lb : r.dl;
                r.du = sact ? (double)slot_ub : r.du;
                r.act = r.act || sact;
            }
        }
        r.ll = r.act ? in.obs[C][0] : 0.0; r.lu = r.act ? in.obs[C][1] : 0.0;
        r.tl = r.act ? in.obs[C][2] : 1.0; r.tu = r.act ? in.obs[C][3] : 1.0;
        if constexpr (SOFT) {
            r.sl = softrow ? in.obs[C][4] : 0.0; r.su = softrow ? in.obs[C][5] : 0.0;
            r.lsl = softrow ? in.obs[C][6] : 0.0; r.lsu = softrow ? in.obs[C][7] : 0.0;
            r.tsl = softrow ? in.obs[C][8] : 1.0; r.tsu = softrow ? in.obs[C][9] : 1.0;
        }
    }
    // c' vec of the row in this lane of chunk C: an obstacle row's gradient sits on the two position lanes (vobs: the vector it
    // multiplies), a box row's (MERGE, slot lanes of the last chunk) is the unit vector of its variable (vbox).  Wave-uniform
    // control flow.
    template <int C>
    USV_DEV double rowdot(double cx, double cy, double vobs, double vbox) const
    {
        double d = obs_dot(cx, cy, vobs);
        if constexpr (MERGE && C == KCH - 1) {
            const double g = lanes::gather(vbox, ssrc);
            d = isslot ? g : d;
        }
        return d;
    }

    // ------------------------------------------------------------------ backward sweeps
    // FACT = true : apply the pending step (pend), residuals + norms, adjoint multipliers, Hessian
    //               reduction, Riccati factorisation, predictor rhs
    // FACT = false: corrector rhs only, reusing the stored factors
    // CPC (option "cond_pred_corr", throughput kernels over planes in HBM): the rows' corrector targets carry the per-row factor so_cur (this
    // pass) / so_prv (the pending step, replayed in backward A) - 0 where the corrected step was refused and the centring-only one taken
    template <bool FACT>
    USV_DEV void backward(Norms &nm, double sigmu, bool pend, double a_prev, double sigmu_prev)
    {
        if constexpr (WIDE) { backward_wide<FACT>(nm, sigmu, pend, a_prev, sigmu_prev); return; }
        constexpr int SW = FACT ? SW_BACK_A : SW_BACK_B;
        double so_c = 1.0, so_p = 1.0;
        if constexpr (CPC) { so_c = so_cur; so_p = so_prv; }
        double Pn[NX], pn = 0.0, pin = 0.0;
        sfor<0, NX>([&](auto c) { Pn[c] = 0.0; });
        if (FACT) {
            nm.rg = nm.rb = nm.rd = nm.rm = nm.musum = nm.nan = 0.0;
            rbscale = pend ? rbscale * (1.0 - a_prev) : rbscale; // the step applied in this sweep
        }
        // Results of a stage are stored one stage LATE (dfr_*), right after the next stage has consumed its
        // prefetched planes: the wait at the top of a stage then covers loads that have been in flight for a
        // whole stage and nothing else - with the stores issued at the end of their own stage it would also cover
        // those, i.e. a store round trip per stage.
        double dfr_pi = 0.0, dfr_pb = 0.0, dfr_aux = 0.0, dfr_lz[NU > 0 ? NU : 1];
        int dfr_k = -1;
        sfor<0, NU>([&](auto l) { dfr_lz[l] = 0.0; });
        auto flush = [&]() {
            if (dfr_k >= 0) { // wave-uniform
                const Planes Wp = ws(dfr_k);
                if (!keep) {
                    if (FACT) Wp.st(P_PI, dfr_pi);
                    aux_st(dfr_k, Wp, dfr_aux);
                }
                if (FACT && dfr_k < N) {
                    Wp.st(P_PB, dfr_pb);
                    sfor<0, NU>([&](auto l) { Wp.st(P_LZU + l, dfr_lz[l]); });
                }
            }
        };
        StageIn nxt;
        load_in<SW>(N, nxt);
        for (int k = N; k >= 0; k--) {
            const StageIn in = nxt;
            lanes::sched_fence();
            flush();
            const Planes W = ws(k);
            // this stage's packed [B A] planes: in flight while the rows below are processed
            double mpk[MP::NPK];
            if (k < N) mat_issue(k, mpk);
            double z = in.z; // the absolute iterate zbar + z
            const double zbx = aux_zx(in.aux), zby = aux_zy(in.aux);
            const double psel = pos_sel(zbx, zby);
            // 1.0 on the state lanes of structurally unit rows of [A B]: "x += unit ? y : 0" as one FMA instead of two selects and an add
            const double ou1 = ounit ? 1.0 : 0.0;
            const double *Hrow = (k < N ? S.Hc : S.He) + lane * LANES; // only read when !HDIAG
            const double hd = (k < N) ? hd_stage : hd_term;
            const double dza = FACT ? 0.0 : in.dza;
            double rb = FACT ? in.rb * rbscale : 0.0;
            // ---- rows (with the pending update of the previous iteration applied first)
            BoxRow br;
            double Ghb = 0.0, gamb = 0.0, dlb = 0.0; // the box row of this lane's variable: Hessian / gradient terms, ll - lu
            const double dzp = FACT ? in.dz : 0.0, dzap = FACT ? in.dza : 0.0;
            double pk[4], dv = in.aux; // dense part of the aux plane: rebuilt where the rows change, else as loaded
            const double znew = (FACT && pend) ? z + a_prev * dzp : z;
            if constexpr (!MERGE) {
                box_from(in, k, br);
                if (FACT) {
                    if (pend && br.act) {
                        chain<CPC>(br, z, true, dzap, sigmu_prev, Ghb, gamb, so_p);
                        br.expand(dzp);
                        br.apply(a_prev);
                        if constexpr (!PACK) box_store(W, br);
                    }
                }
                if constexpr (FACT && PACK) dv = box_pack(br, pk);
                chain<CPC>(br, znew, !FACT, dza, sigmu, Ghb, gamb, so_c);
                dlb = br.act ? br.ll - br.lu : 0.0;
            }
            double Gh_m = 0.0, gam_m = 0.0, dl_m = 0.0; // MERGE: the last chunk's per-row terms, for the box rows among them
            double Sxx = 0.0, Sxy = 0.0, Syy = 0.0, gx = 0.0, gy = 0.0, lx = 0.0, ly = 0.0;
            if constexpr (KCH > 0) {
                sfor<0, KCH>([&](auto c) {
                    ObsRow o;
                    double cx, cy, Gh, gam;
                    obs_from<c>(in, k, zbx, zby, o, cx, cy);
                    if (FACT) {
                        const double vo = rowdot<c>(cx, cy, z - psel, z), wp = rowdot<c>(cx, cy, dzp, dzp), wap = rowdot<c>(cx, cy, dzap, dzap);
                        if (pend && o.act) {
                            chain<CPC>(o, vo, true, wap, sigmu_prev, Gh, gam, so_p);
                            o.expand(wp);
                            o.apply(a_prev);
                        }
                        const bool slot_here = c == KCH - 1 && isslot;
                        if (pend && (o.act || slot_here)) obs_store(W, c, o, (PACK && !MERGE && c == KCH - 1) ? pk : nullptr);
                    }
                    const double v = rowdot<c>(cx, cy, znew - psel, znew);
                    const double wa = FACT ? 0.0 : rowdot<c>(cx, cy, dza, dza);
                    chain<CPC>(o, v, !FACT, wa, sigmu, Gh, gam, so_c);
                    gx += gam * cx; gy += gam * cy;
                    if constexpr (MERGE && c == KCH - 1) { Gh_m = Gh; gam_m = gam; }
                    if (FACT) {
                        Sxx += Gh * cx * cx; Sxy += Gh * cx * cy; Syy += Gh * cy * cy;
                        const double dl_ = o.act ? o.ll - o.lu : 0.0;
                        if constexpr (MERGE && c == KCH - 1) dl_m = dl_;
                        lx += dl_ * cx; ly += dl_ * cy;
                        if (o.act) {
                            nm.rd = lanes::vmax(nm.rd, lanes::vmax_abs2(o.rdl, o.rdu));
                            nm.rm = lanes::vmax(nm.rm, lanes::vmax(o.ll * o.tl, o.lu * o.tu));
                            nm.musum += o.ll * o.tl + o.lu * o.tu;
                            nm.nan = fma(0.0, o.rdl + o.rdu, nm.nan);
                            if constexpr (SOFT) {
                                nm.rg = lanes::vmax(nm.rg, lanes::vmax_abs2(o.rsl, o.rsu));
                                nm.rd = lanes::vmax(nm.rd, lanes::vmax_abs2(o.rdsl, o.rdsu));
                                nm.rm = lanes::vmax(nm.rm, lanes::vmax(o.lsl * o.tsl, o.lsu * o.tsu));
                                nm.musum += o.lsl * o.tsl + o.lsu * o.tsu;
                                nm.nan = fma(0.0, o.rsl + o.rsu + o.rdsl + o.rdsu, nm.nan);
                            }
                        }
                    }
                });
                gx = lanes::gsum(gx); gy = lanes::gsum(gy);
                if (FACT) {
                    Sxx = lanes::gsum(Sxx); Sxy = lanes::gsum(Sxy); Syy = lanes::gsum(Syy);
                    lx = lanes::gsum(lx); ly = lanes::gsum(ly);
                }
                if constexpr (MERGE) { // the box rows' terms come home from their slot lanes (an inactive row has delivered zeros)
                    const double g1 = lanes::gather(gam_m, bsrc);
                    gamb = hasb ? g1 : 0.0;
                    if (FACT) {
                        const double g0 = lanes::gather(Gh_m, bsrc), g2 = lanes::gather(dl_m, bsrc);
                        Ghb = hasb ? g0 : 0.0;
                        dlb = hasb ? g2 : 0.0;
                    }
                }
            }
            z = znew;
            if (FACT && pend) W.st(P_Z, z);
            // ---- next stage's planes go in flight before the matrix work of this one
            if (k > 0) load_in<SW>(k - 1, nxt);
            double bat[NX];
            if (k < N) mat_unpack(mpk, bat); // wave-uniform
            else sfor<0, NX>([&](auto j) { bat[j] = 0.0; });

            double rg, pik = 0.0;
            if (FACT) {
                // t = H (zbar + z) - M yref + [B A]' pi_{k+1} - sum c (ll - lu);  x lanes: pi_k := t (adjoint
                // recursion, stationarity in x holds by construction);  u lanes: residual r_g
                double t = in.gq;
                if constexpr (HDIAG) t = fma(hd, z, t);
                else dot_lanes<ZMASK, 0>(t, z, [&](auto c) { return Hrow[c]; });
                lanes::settle(pin); // (a constant 0.0 in the peeled terminal stage: materialised right in front of its use)
                dot_lanes<NONUNIT, NU>(t, pin, [&](auto j) { return bat[j]; });
                if constexpr (M::OUT_UNIT != 0u) t = fma(ou1, pin, t);
                t -= dlb;
                t -= isPX ? lx : (isPY ? ly : 0.0);
                pik = xlane ? t : 0.0;
                rg = (ulane && k < N) ? t : 0.0;
                dfr_pi = rg + pik; // rg lives on the u lanes, pik on the x lanes
                nm.rg = lanes::vmax_abs(nm.rg, rg);
                nm.nan = fma(0.0, t, nm.nan);
                if constexpr (!MERGE) if (br.act) { // (MERGE: the box rows have been counted with the rows of their chunk)
                    nm.rd = lanes::vmax(nm.rd, lanes::vmax_abs2(br.rdl, br.rdu));
                    nm.rm = lanes::vmax(nm.rm, lanes::vmax(br.ll * br.tl, br.lu * br.tu));
                    nm.musum += br.ll * br.tl + br.lu * br.tu;
                    nm.nan = fma(0.0, br.rdl + br.rdu, nm.nan);
                    if constexpr (SOFTBOX) {
                        if (br.soft) {
                            nm.rg = lanes::vmax(nm.rg, lanes::vmax_abs2(br.rsl, br.rsu));
                            nm.rd = lanes::vmax(nm.rd, lanes::vmax_abs2(br.rdsl, br.rdsu));
                            nm.rm = lanes::vmax(nm.rm, lanes::vmax(br.lsl * br.tsl, br.lsu * br.tsu));
                            nm.musum += br.lsl * br.tsl + br.lsu * br.tsu;
                            nm.nan = fma(0.0, br.rsl + br.rsu + br.rdsl + br.rdsu, nm.nan);
                        }
                    }
                }
                nm.rb = lanes::vmax_abs(nm.rb, rb);
            } else {
                rg = (k < N) ? aux_ulane<AXL_RG>(in.aux) : 0.0;
            }
            const double gt = rg + gamb + (isPX ? gx : (isPY ? gy : 0.0));

            double pv, luv_new = 0.0;
            if (k == N) {
                if (FACT) sfor<0, NX>([&](auto c) {
                    if constexpr (HDIAG) Pn[c] = (lane == NU + c) ? hd : 0.0;
                    else Pn[c] = xlane ? Hrow[NU + c] : 0.0;
                });
                pv = xlane ? gt : 0.0;
            } else {
                double Lzu[NU], iLd[NU], Pb;
                if (FACT) {
                    // P_{k+1} b_k first: each column of P_{k+1} then dies as soon as its column of T is formed
                    Pb = 0.0;
                    lanes::settle(rb); // (scaled a few instructions ago)
                    dot_lanes<XMASK, NU>(Pb, rb, [&](auto c) { return Pn[c]; });
                    // T = [B A]' P_{k+1}   (row r: sum_j bat_j * P_{k+1}[j][:])
                    double T[NX];
                    sfor<0, NX>([&](auto c) {
                        double a = 0.0;
                        dot_lanes<NONUNIT, NU>(a, Pn[c], [&](auto j) { return bat[j]; });
                        // unit rows of [A B] (x+_j = x_j): row nu+j of T receives row j of P, which that lane owns
                        if constexpr (M::OUT_UNIT != 0u) a = fma(ou1, Pn[c], a);
                        T[c] = a;
                    });
                    // G = H~ + T [B A]     (row r, column c': sum_j T_j * BAt[c'][j]), one column at a time
                    auto gcol = [&](auto c) {
                        double a;
                        if constexpr (HDIAG) a = (lane == c) ? hd + Ghb : 0.0;
                        else a = Hrow[c] + ((lane == c) ? Ghb : 0.0);
                        if constexpr (KCH > 0) {
                            if constexpr (c == PXL) a += isPX ? Sxx : (isPY ? Sxy : 0.0);
                            if constexpr (c == PYL) a += isPX ? Sxy : (isPY ? Syy : 0.0);
                        }
                        if constexpr (((M::IN_UNIT >> c) & 1u) != 0u) {
                            // column c of [B A] is a unit vector (the variable feeds no right-hand side): the
                            // whole sum collapses to one term
                            if constexpr (c >= NU) a += T[c - NU];
                        } else {
                            dot_col<NONUNIT, c>(a, [&](auto j) { return bat[j]; }, [&](auto j) { return T[j]; });
                            // column nu+j of a unit row is e_j
                            if constexpr (c >= NU) { if constexpr (out_unit(c - NU)) a += T[c - NU]; }
                        }
                        return a;
                    };
                    // the nu control columns and their Cholesky factor, all rows at once
                    double Gu[NU > 0 ? NU : 1];
                    sfor<0, NU>([&](auto c) { Gu[c] = gcol(c); });
                    sfor<0, NU>([&](auto l) {
                        const double il = lanes::frsqrt(lanes::bcast<l>(Gu[l]));
                        Lzu[l] = Gu[l] * il;
                        lanes::settle(Lzu[l]); // DPP source of the updates below
                        iLd[l] = il;
                        sfor<l + 1, NU>([&](auto m) { lanes::fma_bc<m>(Gu[m], Lzu[l], -Lzu[l]); });
                    });
                    // P_k = G_xx - Lxu Lxu': every state column is consumed as soon as it is formed (16 columns
                    // of G are never alive together)
                    sfor<0, NX>([&](auto c) {
                        double a = gcol(std::integral_constant<int, NU + c>{});
                        dot_col<(1u << NU) - 1u, NU + c>(a, [&](auto l) { return Lzu[l]; }, [&](auto l) { return -Lzu[l]; });
                        // (all 16 lanes keep their value: nothing reads the control lanes of a column of P - T broadcasts
                        // state lanes only, P b and the vector recursion are masked where they are stored or summed)
                        Pn[c] = a;
                    });
                    // stored with the RECIPROCAL of the diagonal entry in its place: that is all the three later sweeps want of it
                    sfor<0, NU>([&](auto l) { dfr_lz[l] = (lane == l) ? iLd[l] : Lzu[l]; });
                } else {
                    Pb = xlane ? in.pb : 0.0;
                    sfor<0, NU>([&](auto l) {
                        Lzu[l] = in.lzu[l];
                        iLd[l] = lanes::bcast<l>(Lzu[l]); // (stored as the reciprocal)
                    });
                }
                // vector recursion
                double h = Pb + pn;
                lanes::settle(h);
                double rq = gt;
                dot_lanes<NONUNIT, NU>(rq, h, [&](auto j) { return bat[j]; });
                if constexpr (M::OUT_UNIT != 0u) rq = fma(ou1, h, rq);
                double lu[NU], luv = 0.0;
                sfor<0, NU>([&](auto l) {
                    double a = lanes::bcast<l>(rq);
                    sfor<0, l>([&](auto m) { a -= lanes::bcast<l>(Lzu[m]) * lu[m]; });
                    lu[l] = a * iLd[l];
                    luv = (lane == l) ? lu[l] : luv;
                });
                pv = rq;
                sfor<0, NU>([&](auto l) { pv -= Lzu[l] * lu[l]; });
                pv = xlane ? pv : 0.0;
                dfr_pb = Pb;
                luv_new = luv;
            }
            dfr_aux = aux_compose(dv, zbx, zby, rg, luv_new);
            pn = pv;
            pin = pik;
            dfr_k = k;
        }
        flush();
        if (FACT) {
            const Planes W0 = ws(0);
            const double e0 = xlane ? W0.ld(P_DX0) - W0.ld(P_Z) : 0.0; // x0 - (xbar_0 + dx_0)
            nm.rg = lanes::gmax(nm.rg);
            nm.rb = lanes::gmax(fmax(nm.rb, fabs(e0)));
            nm.rd = lanes::gmax(nm.rd);
            nm.rm = lanes::gmax(nm.rm);
            nm.musum = lanes::gsum(nm.musum);
            nm.nan = lanes::gsum(nm.nan);
        }
    }

    // ------------------------------------------------------------------ forward sweeps
    // FINAL = false: affine step -> alpha_aff and the sums for mu_aff, stores dza
    // FINAL = true : corrected step -> alpha, stores dz
    // (CPC: the corrected step also delivers the sums S1, S2 - the duality measure after a step of length a is (musum + a S1 + a^2 S2) / nc)
    template <bool FINAL>
    USV_DEV void forward(double sigmu, double &alpha, double &S1, double &S2)
    {
        if constexpr (WIDE) { forward_wide<FINAL>(sigmu, alpha, S1, S2); return; }
        constexpr bool SUMS = !FINAL || CPC;
        double so_c = 1.0;
        if constexpr (CPC) so_c = so_cur;
        constexpr int SW = FINAL ? SW_FWD_B : SW_FWD_A;
        double dzx;
        {
            const Planes W0 = ws(0);
            dzx = xlane ? W0.ld(P_DX0) - W0.ld(P_Z) : 0.0;
        }
        double q = 1.0, s1 = 0.0, s2 = 0.0, dfr_dz = 0.0;
        StageIn nxt;
        load_in<SW>(0, nxt);
        for (int k = 0; k <= N; k++) {
            const StageIn in = nxt;
            lanes::sched_fence();
            if (k > 0) ws(k - 1).st(FINAL ? P_DZ : P_DZA, dfr_dz); // stored one stage late (see backward())
            const Planes W = ws(k);
            // this stage's packed [B A] planes and the next stage's small planes: both in flight during the
            // gain / row computations below
            double mpk[MP::NPK];
            if (k < N) {
                mat_issue(k, mpk);
                load_in<SW>(k + 1, nxt);
            }
            double dz;
            const double zbx = aux_zx(in.aux), zby = aux_zy(in.aux);
            if (k < N) {
                double t[NU], du[NU];
                sfor<0, NU>([&](auto l) {
                    t[l] = lanes::bcast<AXL_LU - l>(in.aux) + lanes::gsum(xlane ? in.lzu[l] * dzx : 0.0);
                });
                sfor<0, NU>([&](auto qq) { // back substitution with Luu'
                    constexpr int l = NU - 1 - qq;
                    double acc = t[l];
                    sfor<l + 1, NU>([&](auto m) { acc -= lanes::bcast<m>(in.lzu[l]) * du[m]; });
                    du[l] = acc * lanes::bcast<l>(in.lzu[l]); // (the diagonal entry is stored as its reciprocal)
                });
                dz = xlane ? dzx : 0.0;
                sfor<0, NU>([&](auto l) { dz = (lane == l) ? -du[l] : dz; });
            } else {
                dz = xlane ? dzx : 0.0;
            }
            // ---- rows of stage k
            {
                const double z = in.z;
                const double dza = FINAL ? in.dza : dz;
                if constexpr (!MERGE) { // (MERGE: the box rows are rows of the last chunk below)
                    BoxRow br;
                    box_from(in, k, br);
                    double Gh, gam;
                    chain<CPC>(br, z, FINAL, dza, sigmu, Gh, gam, so_c);
                    br.expand(dz);
                    q = br.blocking(q);
                    if (SUMS && br.act) {
                        s1 += br.ll * br.dtl + br.tl * br.dll + br.lu * br.dtu + br.tu * br.dlu;
                        s2 += br.dll * br.dtl + br.dlu * br.dtu;
                        if constexpr (SOFTBOX) {
                            if (br.soft) {
                                s1 += br.lsl * br.dtsl + br.tsl * br.dlsl + br.lsu * br.dtsu + br.tsu * br.dlsu;
                                s2 += br.dlsl * br.dtsl + br.dlsu * br.dtsu;
                            }
                        }
                    }
                }
                if constexpr (KCH > 0) {
                    sfor<0, KCH>([&](auto c) {
                        ObsRow o;
                        double cx, cy, Gh2, gam2;
                        obs_from<c>(in, k, zbx, zby, o, cx, cy);
                        const double v = rowdot<c>(cx, cy, z - pos_sel(zbx, zby), z);
                        const double w = rowdot<c>(cx, cy, dz, dz);
                        const double wa = FINAL ? rowdot<c>(cx, cy, dza, dza) : w;
                        chain<CPC>(o, v, FINAL, wa, sigmu, Gh2, gam2, so_c);
                        o.expand(w);
                        q = o.blocking(q);
                        if (SUMS && o.act) {
                            s1 += o.ll * o.dtl + o.tl * o.dll + o.lu * o.dtu + o.tu * o.dlu;
                            s2 += o.dll * o.dtl + o.dlu * o.dtu;
                            if constexpr (SOFT) {
                                s1 += o.lsl * o.dtsl + o.tsl * o.dlsl + o.lsu * o.dtsu + o.tsu * o.dlsu;
                                s2 += o.dlsl * o.dtsl + o.dlsu * o.dtsu;
                            }
                        }
                    });
                }
            }
            dfr_dz = dz;
            if (k < N) {
                // dx+ = b + [B A] dz: the state lanes read their rows of the matrix column by column from the exchange area
                // (no row-wise reductions, no transposed copy of the matrix in HBM)
                const double dxn = mat_apply(mpk, dz, in.rb * rbscale);
                dzx = xlane ? dxn : 0.0;
            }
        }
        ws(N).st(FINAL ? P_DZ : P_DZA, dfr_dz);
        alpha = 1.0 / lanes::gmax(q); // q >= 1: alpha = min(1, min over blocking pairs of -v/dv)
        if (SUMS) { S1 = lanes::gsum(s1); S2 = lanes::gsum(s2); }
    }

    // ------------------------------------------------------------------ the sweeps of the WIDE mapping
    // Exchange area behind the instance's planes in the workgroup's LDS: [row][EX_N][16 lanes].  Row r leaves the terms of the stage it
    // has just processed in its own slice; the recursion reads slice j for the block's j-th stage in every row.
    // (EX_SC: the row-uniform sums S_xx, S_xy, S_yy, g_x, g_y, l_x, l_y in lanes 0 .. 6 of one plane)
    // EX_Z, EX_DV (planes in HBM only): the stage's iterate after the pending step and the dense box values, which with the planes in LDS
    // the recursion reads back from the planes the row phase has just written.
    // (EX_MU1 + c, EX_MU2 + c: the complementarity sums of obstacle chunk c - hard pairs, slack pairs -, EX_MU3: of the box rows of the two-pass
    // form; the forward sweeps reuse the area for the sums of mu_aff: chunk c in planes 4 c .. 4 c + 3, the box rows in 4 KC, 4 KC + 1)
    // (EX_MU4, soft state bounds only: the sums of the box rows' slack pairs)
    enum : int { EX_GHB = 0, EX_GAMB, EX_DLB, EX_SC, EX_MU1, EX_MU2 = EX_MU1 + KC, EX_MU3 = EX_MU2 + KC, EX_MU4 = EX_MU3 + (SOFTBOX ? 1 : 0), EX_Z, EX_DV, EX_ALL };
    static constexpr int EX_FWD = 4 * KC + 2 + (SOFTBOX ? 2 : 0);
    static constexpr int EX_N = LDSWS ? ((int)EX_Z > EX_FWD ? (int)EX_Z : EX_FWD) : ((int)EX_ALL > EX_FWD ? (int)EX_ALL : EX_FWD);
    static_assert(!WIDE || EX_N == (LDSWS ? wide_ex_planes(KCH, SOFTBOX) : wide_ex_planes_hbm(KCH, SOFTBOX)), "host-side size of the exchange area");
    static constexpr int BS = 4 * WW; // stages per block = rows of the workgroup
    static constexpr int wide_lds_doubles(int N_) { return (LDSWS ? (N_ + 1) * NPLW * LANES : 0) + BS * EX_N * LANES + (WW > 1 ? LANES : 0); }
    // phases of a sweep hand values from row to row through LDS (or, WW > 1, from wave to wave: a workgroup barrier)
    USV_DEV static void wide_sync()
    {
        if constexpr (WW > 1) lanes::block_sync();
        else lanes::lds_fence();
    }
    // ... at the start of a sweep: the cold start's / the last sweep's stores of other waves have landed
    USV_DEV static void sweep_sync()
    {
        if constexpr (WW > 1 && !LDSWS) lanes::drain_stores();
        wide_sync();
    }
    // a wave-uniform value reduced over the workgroup's waves (WW > 1): through a few LDS words behind the exchange area
    template <bool MAX>
    USV_DEV double xwave(double v) const
    {
        if constexpr (WW == 1) return v;
        else {
            double *sc = lanes::dyn_lds() + (LDSWS ? (N + 1) * NPLW * LANES : 0) + BS * EX_N * LANES;
            sc[lanes::block_row() >> 2] = v;
            lanes::block_sync();
            double r = sc[0];
            for (int w = 1; w < WW; w++) r = MAX ? lanes::vmax(r, sc[w]) : r + sc[w];
            lanes::block_sync();
            return r;
        }
    }
    USV_DEV int xwave_first_i(int v) const // wave 0's value
    {
        if constexpr (WW == 1) return v;
        else {
            double *sc = lanes::dyn_lds() + (LDSWS ? (N + 1) * NPLW * LANES : 0) + BS * EX_N * LANES;
            if (lanes::block_row() < 4u) sc[0] = (double)v;
            lanes::block_sync();
            const int r = (int)sc[0];
            lanes::block_sync();
            return r;
        }
    }
    USV_DEV unsigned ex_at(int row, int plane) const { return (unsigned)((LDSWS ? (N + 1) * NPLW * LANES : 0) + (row * EX_N + plane) * LANES + lane); }
    USV_DEV void ex_put(int row, int plane, double v) const { lanes::dyn_lds()[ex_at(row, plane)] = v; }
    USV_DEV double ex_get(int row, int plane) const { return lanes::dyn_lds()[ex_at(row, plane)]; }
    // The planes of one stage as the wide sweeps see them: every lane loads, the lanes `on` selects store.
    struct WPl {
        Planes W;
        bool on;
        USV_DEV double ld(int plane) const { return W.ld(plane); }
        USV_DEV void st(int plane, double v) const
        {
            if constexpr (LDSWS) W.st(plane, v); // (the LDS planes carry the predicate themselves)
            else { if (on) W.st(plane, v); }
        }
    };
    // ... for the row phase: row r addresses ITS stage (own = this row has a stage in the block) and stores that stage's rows.
    // Planes in HBM: the rows of a workgroup address the BS consecutive stages of a block through ONE buffer window that starts at the
    // block's lowest stage kbase (wave-uniform) - offsets stay below 2^32 whatever the batch (BS stage windows; the host checks), so
    // the mapping also serves the long runners a 65 536-instance launch hands over (suspend / resume below).
    USV_DEV WPl ws_row(int k, int kbase, bool own) const
    {
        if constexpr (LDSWS) return WPl{Planes(loff + (unsigned)(k * NPLW * LANES), own), own};
        else {
            const int nst = (N + 1 - kbase < BS) ? N + 1 - kbase : BS;
            return WPl{lanes::Planes(P.ws + (long)kbase * stage_stride, (unsigned)nst * stage_bytes, voff + (unsigned)(k - kbase) * stage_bytes), own};
        }
    }
    USV_DEV static int base_back(int kb) { return kb >= BS - 1 ? kb - (BS - 1) : 0; } // lowest stage of the backward sweeps' block at kb
    // ... for the recursion (the same stage in all rows): row 0 stores
    USV_DEV WPl ws_seq(int k) const { return WPl{ws(k), live}; }
    // What the recursion reads of a stage, asked for one stage ahead of its use: the lineariser's planes (always in HBM / L2) and - with
    // the solver's planes in HBM too - the aux plane, P b and L_zu.
    struct SeqIn { double mpk[MP::NPK], gq, rb, aux, pb, lzu[NU > 0 ? NU : 1]; };
    template <int SW>
    USV_DEV void seq_load(int k, SeqIn &si) const
    {
        const lanes::Planes G = wsg(k);
        if (k < N) { // wave-uniform
            sfor<0, MP::NPK>([&](auto q) { si.mpk[q] = G.ld(P_MAT + q); });
            si.rb = G.ld(P_RB0);
        } else {
            si.rb = 0.0;
        }
        if (SW == SW_BACK_A) si.gq = G.ld(P_GQ);
        if constexpr (!LDSWS) {
            si.aux = G.ld(P_AUX);
            if (SW == SW_BACK_B) si.pb = (k < N) ? G.ld(P_PB) : 0.0;
            if (SW != SW_BACK_A) {
                if (k < N) sfor<0, NU>([&](auto l) { si.lzu[l] = G.ld(P_LZU + l); });
                else sfor<0, NU>([&](auto l) { si.lzu[l] = 1.0; });
            }
        }
    }
    // What the row phase reads of its stage (all row planes whatever the stage: in bounds, unused values are masked).  With the planes in
    // HBM the next block's are asked for before the recursion of this one starts.
    template <int SW>
    USV_DEV void row_load(int k, const WPl &W, StageIn &in) const
    {
        in.z = W.ld(P_Z);
        in.aux = W.ld(P_AUX);
        if (SW == SW_BACK_A) in.dz = W.ld(P_DZ);
        if (SW != SW_FWD_A) in.dza = W.ld(P_DZA);
        if constexpr (KCH > 0) {
            sfor<0, KCH>([&](auto c) {
                sfor<0, OBSN>([&](auto e) { in.obs[c][e] = W.ld(P_OBS + c * OBSN + e); });
                if (!pstat) obs_raw<c>(k, in.raw[c]);
            });
        }
        if constexpr (!PACK) {
            in.box[0] = W.ld(P_BLL); in.box[1] = W.ld(P_BLU); in.box[2] = W.ld(P_BTL); in.box[3] = W.ld(P_BTU);
            if constexpr (SOFTBOX) sfor<0, 6>([&](auto e) { in.bxs[e] = W.ld(P_BS + e); });
        }
    }

    template <bool FACT>
    USV_DEV void backward_wide(Norms &nm, double sigmu, bool pend, double a_prev, double sigmu_prev)
    {
        const int row = (int)lanes::block_row();
        double so_c = 1.0, so_p = 1.0; // (CPC: the same in every row and wave of the workgroup - they hold one instance)
        if constexpr (CPC) { so_c = so_cur; so_p = so_prv; }
        double Pn[NX], pn = 0.0, pin = 0.0;
        sfor<0, NX>([&](auto c) { Pn[c] = 0.0; });
        double rg_r = 0.0, rd_r = 0.0, rm_r = 0.0, nan_r = 0.0; // what the norms get from the rows this row has processed
        if (FACT) {
            nm.rg = nm.rb = nm.rd = nm.rm = nm.musum = nm.nan = 0.0;
            rbscale = pend ? rbscale * (1.0 - a_prev) : rbscale; // the step applied in this sweep
        }
        constexpr int SW = FACT ? SW_BACK_A : SW_BACK_B;
        sweep_sync();
        SeqIn nxt;
        seq_load<SW>(N, nxt);
        StageIn rnx;
        if constexpr (!LDSWS) {
            const int kr = N - row;
            row_load<SW>(kr >= 0 ? kr : 0, ws_row(kr >= 0 ? kr : 0, base_back(N), false), rnx);
        }
        for (int kb = N; kb >= 0; kb -= BS) {
            // ---- row phase: row r on stage kb - r (with the pending update of the previous iteration applied first)
            {
                const int kr = kb - row;
                const bool own = kr >= 0;
                const int k = own ? kr : 0;
                const WPl W = ws_row(k, base_back(kb), own);
                StageIn in;
                if constexpr (LDSWS) row_load<SW>(k, W, in);
                else in = rnx;
                const double z = in.z, aux = in.aux;
                const double dzp = FACT ? in.dz : 0.0;
                const double dzs = in.dza;
                const double dzap = FACT ? dzs : 0.0, dza = FACT ? 0.0 : dzs;
                const double zbx = aux_zx(aux), zby = aux_zy(aux);
                const double psel = pos_sel(zbx, zby);
                const double znew = (FACT && pend) ? z + a_prev * dzp : z;
                double Sxx = 0.0, Sxy = 0.0, Syy = 0.0, gx = 0.0, gy = 0.0, lx = 0.0, ly = 0.0, dl_m = 0.0, mu3 = 0.0, mu4 = 0.0;
                // the two-pass form (a box row rides in the dense part of the aux plane): the box rows in their variables' lanes first
                double Ghb = 0.0, gamb = 0.0, dlb = 0.0, pk[4];
                if constexpr (!MERGE) {
                    BoxRow br;
                    in.aux = aux;
                    box_from(in, k, br);
                    br.act = br.act && own;
                    if (FACT) {
                        if (pend && br.act) {
                            chain<CPC>(br, z, true, dzap, sigmu_prev, Ghb, gamb, so_p);
                            br.expand(dzp);
                            br.apply(a_prev);
                        }
                        if constexpr (PACK) {
                            const double dv = box_pack(br, pk);
                            // (the recursion composes the aux plane around the dense lanes)
                            if constexpr (LDSWS) ws_row(k, 0, own && isdense).st(P_AUX, dv);
                            else ex_put(row, EX_DV, dv);
                        } else {
                            if (pend && br.act) box_store(W, br);
                            if constexpr (!LDSWS) ex_put(row, EX_DV, aux);
                        }
                    }
                    chain<CPC>(br, znew, !FACT, dza, sigmu, Ghb, gamb, so_c);
                    dlb = br.act ? br.ll - br.lu : 0.0;
                    if (FACT && br.act) {
                        rd_r = lanes::vmax(rd_r, lanes::vmax_abs2(br.rdl, br.rdu));
                        rm_r = lanes::vmax(rm_r, lanes::vmax(br.ll * br.tl, br.lu * br.tu));
                        mu3 = br.ll * br.tl + br.lu * br.tu;
                        nan_r = fma(0.0, br.rdl + br.rdu, nan_r);
                        if constexpr (SOFTBOX) {
                            if (br.soft) {
                                rg_r = lanes::vmax(rg_r, lanes::vmax_abs2(br.rsl, br.rsu));
                                rd_r = lanes::vmax(rd_r, lanes::vmax_abs2(br.rdsl, br.rdsu));
                                rm_r = lanes::vmax(rm_r, lanes::vmax(br.lsl * br.tsl, br.lsu * br.tsu));
                                mu4 = br.lsl * br.tsl + br.lsu * br.tsu;
                                nan_r = fma(0.0, br.rsl + br.rsu + br.rdsl + br.rdsu, nan_r);
                            }
                        }
                    }
                }
                double mu1[KC], mu2[KC];
                sfor<0, KC>([&](auto c) { mu1[c] = 0.0; mu2[c] = 0.0; });
                if constexpr (KCH > 0) {
                double Gh_m = 0.0, gam_m = 0.0; // the last chunk's per-row terms (MERGE: the box rows are among them)
                sfor<0, KCH>([&](auto c) {
                    ObsRow o;
                    double cx, cy, Gh, gam;
                    obs_from<c>(in, k, zbx, zby, o, cx, cy);
                    o.act = o.act && own;
                    if (FACT) {
                        const double vo = rowdot<c>(cx, cy, z - psel, z), wp = rowdot<c>(cx, cy, dzp, dzp), wap = rowdot<c>(cx, cy, dzap, dzap);
                        if (pend && o.act) {
                            chain<CPC>(o, vo, true, wap, sigmu_prev, Gh, gam, so_p);
                            o.expand(wp);
                            o.apply(a_prev);
                        }
                        const bool slot_here = c == KCH - 1 && isslot;
                        if (pend && (o.act || slot_here)) obs_store(W, c, o, (!MERGE && c == KCH - 1) ? pk : nullptr);
                    }
                    const double v = rowdot<c>(cx, cy, znew - psel, znew);
                    const double wa = FACT ? 0.0 : rowdot<c>(cx, cy, dza, dza);
                    chain<CPC>(o, v, !FACT, wa, sigmu, Gh, gam, so_c);
                    gx += gam * cx; gy += gam * cy;
                    if constexpr (c == KCH - 1) { Gh_m = Gh; gam_m = gam; }
                    if (FACT) {
                        Sxx += Gh * cx * cx; Sxy += Gh * cx * cy; Syy += Gh * cy * cy;
                        const double dl_ = o.act ? o.ll - o.lu : 0.0;
                        if constexpr (c == KCH - 1) dl_m = dl_;
                        lx += dl_ * cx; ly += dl_ * cy;
                        if (o.act) {
                            rd_r = lanes::vmax(rd_r, lanes::vmax_abs2(o.rdl, o.rdu));
                            rm_r = lanes::vmax(rm_r, lanes::vmax(o.ll * o.tl, o.lu * o.tu));
                            mu1[c] = o.ll * o.tl + o.lu * o.tu;
                            nan_r = fma(0.0, o.rdl + o.rdu, nan_r);
                            if constexpr (SOFT) {
                                rg_r = lanes::vmax(rg_r, lanes::vmax_abs2(o.rsl, o.rsu));
                                rd_r = lanes::vmax(rd_r, lanes::vmax_abs2(o.rdsl, o.rdsu));
                                rm_r = lanes::vmax(rm_r, lanes::vmax(o.lsl * o.tsl, o.lsu * o.tsu));
                                mu2[c] = o.lsl * o.tsl + o.lsu * o.tsu;
                                nan_r = fma(0.0, o.rsl + o.rsu + o.rdsl + o.rdsu, nan_r);
                            }
                        }
                    }
                });
                gx = lanes::gsum(gx); gy = lanes::gsum(gy);
                if (FACT) {
                    Sxx = lanes::gsum(Sxx); Sxy = lanes::gsum(Sxy); Syy = lanes::gsum(Syy);
                    lx = lanes::gsum(lx); ly = lanes::gsum(ly);
                }
                if constexpr (MERGE) { // the box rows' terms come home from their slot lanes (an inactive row has delivered zeros)
                    const double g1 = lanes::gather(gam_m, bsrc);
                    gamb = hasb ? g1 : 0.0;
                    if (FACT) {
                        const double g0 = lanes::gather(Gh_m, bsrc), g2 = lanes::gather(dl_m, bsrc);
                        Ghb = hasb ? g0 : 0.0;
                        dlb = hasb ? g2 : 0.0;
                    }
                }
                } // (KCH > 0)
                ex_put(row, EX_GAMB, gamb);
                {
                    double sc = lane == 3 ? gx : gy; // (lanes beyond 6 are never read)
                    if (FACT) {
                        sc = lane == 0 ? Sxx : sc; sc = lane == 1 ? Sxy : sc; sc = lane == 2 ? Syy : sc;
                        sc = lane == 5 ? lx : sc; sc = lane == 6 ? ly : sc;
                    }
                    ex_put(row, EX_SC, sc);
                }
                if (FACT) {
                    ex_put(row, EX_GHB, Ghb);
                    ex_put(row, EX_DLB, dlb);
                    if constexpr (!MERGE) ex_put(row, EX_MU3, mu3);
                    if constexpr (SOFTBOX) ex_put(row, EX_MU4, mu4);
                    sfor<0, KC>([&](auto c) {
                        ex_put(row, EX_MU1 + c, mu1[c]);
                        if constexpr (SOFT) ex_put(row, EX_MU2 + c, mu2[c]);
                    });
                    if (pend) W.st(P_Z, znew);
                }
                if constexpr (!LDSWS) ex_put(row, EX_Z, znew);
            }
            wide_sync();
            if constexpr (!LDSWS) { // the next block's row planes go in flight before the recursion of this one
                const int kr = kb - BS - row;
                if (kb >= BS) row_load<SW>(kr >= 0 ? kr : 0, ws_row(kr >= 0 ? kr : 0, base_back(kb - BS), false), rnx); // wave-uniform
            }
            // ---- the recursion over the block's stages, in all four rows alike
            for (int j = 0; j < BS; j++) {
                const int k = kb - j;
                if (k < 0) break; // wave-uniform
                const WPl W = ws_seq(k);
                const SeqIn cur = nxt;
                if (k > 0) seq_load<SW>(k - 1, nxt);
                double z, aux;
                if constexpr (LDSWS) { z = W.ld(P_Z); aux = W.ld(P_AUX); }
                else { z = ex_get(j, EX_Z); aux = cur.aux; }
                const double zbx = aux_zx(aux), zby = aux_zy(aux);
                const double ou1 = ounit ? 1.0 : 0.0;
                const double hd = (k < N) ? hd_stage : hd_term;
                double rb = FACT ? cur.rb * rbscale : 0.0;
                const double gamb = ex_get(j, EX_GAMB), sc = ex_get(j, EX_SC);
                const double gx = lanes::bcast<3>(sc), gy = lanes::bcast<4>(sc);
                double Ghb = 0.0, dlb = 0.0, Sxx = 0.0, Sxy = 0.0, Syy = 0.0, lx = 0.0, ly = 0.0;
                if (FACT) {
                    Ghb = ex_get(j, EX_GHB); dlb = ex_get(j, EX_DLB);
                    Sxx = lanes::bcast<0>(sc); Sxy = lanes::bcast<1>(sc); Syy = lanes::bcast<2>(sc);
                    lx = lanes::bcast<5>(sc); ly = lanes::bcast<6>(sc);
                    sfor<0, KC>([&](auto c) { // (chunk by chunk, hard pairs then slack pairs: the order of the 16-lane sweep)
                        nm.musum += ex_get(j, EX_MU1 + c);
                        if constexpr (SOFT) nm.musum += ex_get(j, EX_MU2 + c);
                    });
                    if constexpr (!MERGE) nm.musum += ex_get(j, EX_MU3);
                    if constexpr (SOFTBOX) nm.musum += ex_get(j, EX_MU4);
                }
                double bat[NX];
                if (k < N) mat_unpack(cur.mpk, bat); // wave-uniform
                else sfor<0, NX>([&](auto jj) { bat[jj] = 0.0; });

                double rg, pik = 0.0;
                if (FACT) {
                    double t = cur.gq;
                    t = fma(hd, z, t);
                    lanes::settle(pin);
                    dot_lanes<NONUNIT, NU>(t, pin, [&](auto jj) { return bat[jj]; });
                    if constexpr (M::OUT_UNIT != 0u) t = fma(ou1, pin, t);
                    t -= dlb;
                    t -= isPX ? lx : (isPY ? ly : 0.0);
                    pik = xlane ? t : 0.0;
                    rg = (ulane && k < N) ? t : 0.0;
                    if (!keep) W.st(P_PI, rg + pik); // rg lives on the u lanes, pik on the x lanes
                    nm.rg = lanes::vmax_abs(nm.rg, rg);
                    nm.nan = fma(0.0, t, nm.nan);
                    nm.rb = lanes::vmax_abs(nm.rb, rb);
                } else {
                    rg = (k < N) ? aux_ulane<AXL_RG>(aux) : 0.0;
                }
                const double gt = rg + gamb + (isPX ? gx : (isPY ? gy : 0.0));

                double pv, luv_new = 0.0;
                if (k == N) {
                    if (FACT) sfor<0, NX>([&](auto c) { Pn[c] = (lane == NU + c) ? hd : 0.0; });
                    pv = xlane ? gt : 0.0;
                } else {
                    double Lzu[NU], iLd[NU], Pb;
                    if (FACT) {
                        Pb = 0.0;
                        lanes::settle(rb);
                        dot_lanes<XMASK, NU>(Pb, rb, [&](auto c) { return Pn[c]; });
                        double T[NX];
                        sfor<0, NX>([&](auto c) {
                            double a = 0.0;
                            dot_lanes<NONUNIT, NU>(a, Pn[c], [&](auto jj) { return bat[jj]; });
                            if constexpr (M::OUT_UNIT != 0u) a = fma(ou1, Pn[c], a);
                            T[c] = a;
                        });
                        auto gcol = [&](auto c) {
                            double a = (lane == c) ? hd + Ghb : 0.0;
                            if constexpr (c == PXL) a += isPX ? Sxx : (isPY ? Sxy : 0.0);
                            if constexpr (c == PYL) a += isPX ? Sxy : (isPY ? Syy : 0.0);
                            if constexpr (((M::IN_UNIT >> c) & 1u) != 0u) {
                                if constexpr (c >= NU) a += T[c - NU];
                            } else {
                                dot_col<NONUNIT, c>(a, [&](auto jj) { return bat[jj]; }, [&](auto jj) { return T[jj]; });
                                if constexpr (c >= NU) { if constexpr (out_unit(c - NU)) a += T[c - NU]; }
                            }
                            return a;
                        };
                        double Gu[NU > 0 ? NU : 1];
                        sfor<0, NU>([&](auto c) { Gu[c] = gcol(c); });
                        sfor<0, NU>([&](auto l) {
                            const double il = lanes::frsqrt(lanes::bcast<l>(Gu[l]));
                            Lzu[l] = Gu[l] * il;
                            lanes::settle(Lzu[l]);
                            iLd[l] = il;
                            sfor<l + 1, NU>([&](auto m) { lanes::fma_bc<m>(Gu[m], Lzu[l], -Lzu[l]); });
                        });
                        sfor<0, NX>([&](auto c) {
                            double a = gcol(std::integral_constant<int, NU + c>{});
                            dot_col<(1u << NU) - 1u, NU + c>(a, [&](auto l) { return Lzu[l]; }, [&](auto l) { return -Lzu[l]; });
                            Pn[c] = a;
                        });
                        W.st(P_PB, Pb);
                        sfor<0, NU>([&](auto l) { W.st(P_LZU + l, (lane == l) ? iLd[l] : Lzu[l]); });
                    } else {
                        double pb;
                        if constexpr (LDSWS) pb = W.ld(P_PB);
                        else pb = cur.pb;
                        Pb = xlane ? pb : 0.0;
                        sfor<0, NU>([&](auto l) {
                            if constexpr (LDSWS) Lzu[l] = W.ld(P_LZU + l);
                            else Lzu[l] = cur.lzu[l];
                            iLd[l] = lanes::bcast<l>(Lzu[l]); // (stored as the reciprocal)
                        });
                    }
                    double h = Pb + pn;
                    lanes::settle(h);
                    double rq = gt;
                    dot_lanes<NONUNIT, NU>(rq, h, [&](auto jj) { return bat[jj]; });
                    if constexpr (M::OUT_UNIT != 0u) rq = fma(ou1, h, rq);
                    double lu[NU], luv = 0.0;
                    sfor<0, NU>([&](auto l) {
                        double a = lanes::bcast<l>(rq);
                        sfor<0, l>([&](auto m) { a -= lanes::bcast<l>(Lzu[m]) * lu[m]; });
                        lu[l] = a * iLd[l];
                        luv = (lane == l) ? lu[l] : luv;
                    });
                    pv = rq;
                    sfor<0, NU>([&](auto l) { pv -= Lzu[l] * lu[l]; });
                    pv = xlane ? pv : 0.0;
                    luv_new = luv;
                }
                {
                    double dense = aux;
                    if constexpr (!LDSWS && !MERGE) { if (FACT) dense = ex_get(j, EX_DV); }
                    if (!keep) W.st(P_AUX, aux_compose(dense, zbx, zby, rg, luv_new));
                }
                pn = pv;
                pin = pik;
            }
            wide_sync();
        }
        if constexpr (!LDSWS) lanes::drain_stores(); // (rows read each other's stores in the next sweep)
        if constexpr (!LDSWS && WW > 1) lanes::block_sync(); // (... and stage 0's iterate of another wave right below)
        if (FACT) {
            const Planes W0 = ws(0);
            const double e0 = xlane ? W0.ld(P_DX0) - W0.ld(P_Z) : 0.0; // x0 - (xbar_0 + dx_0)
            if constexpr (WW == 1) {
                nm.rg = lanes::gmax(lanes::vmax(nm.rg, lanes::xrow_max(rg_r)));
                nm.rd = lanes::gmax(lanes::xrow_max(rd_r));
                nm.rm = lanes::gmax(lanes::xrow_max(rm_r));
                nm.nan = lanes::gsum(nm.nan + lanes::xrow_sum(nan_r));
            } else { // (maxima, and a sum of zeros or NaNs: the same values whatever the order)
                nm.rg = lanes::vmax(lanes::gmax(nm.rg), xwave<true>(lanes::gmax(lanes::xrow_max(rg_r))));
                nm.rd = xwave<true>(lanes::gmax(lanes::xrow_max(rd_r)));
                nm.rm = xwave<true>(lanes::gmax(lanes::xrow_max(rm_r)));
                nm.nan = lanes::gsum(nm.nan) + xwave<false>(lanes::gsum(lanes::xrow_sum(nan_r)));
            }
            nm.rb = lanes::gmax(fmax(nm.rb, fabs(e0)));
            nm.musum = lanes::gsum(nm.musum);
        }
    }

    template <bool FINAL>
    USV_DEV void forward_wide(double sigmu, double &alpha, double &S1, double &S2)
    {
        const int row = (int)lanes::block_row();
        constexpr bool SUMS = !FINAL || CPC; // (CPC: the corrected step delivers the sums too - forward())
        double so_c = 1.0;
        if constexpr (CPC) so_c = so_cur;
        double dzx;
        {
            const Planes W0 = ws(0);
            dzx = xlane ? W0.ld(P_DX0) - W0.ld(P_Z) : 0.0;
        }
        double q = 1.0, s1 = 0.0, s2 = 0.0;
        constexpr int SW = FINAL ? SW_FWD_B : SW_FWD_A;
        sweep_sync();
        SeqIn nxt;
        seq_load<SW>(0, nxt);
        StageIn rnx;
        if constexpr (!LDSWS) row_load<SW>(row <= N ? row : N, ws_row(row <= N ? row : N, 0, false), rnx);
        for (int kb = 0; kb <= N; kb += BS) {
            // ---- the recursion over the block's stages, in all four rows alike; row r keeps the step of stage kb + r
            double mydz = 0.0;
            for (int j = 0; j < BS; j++) {
                const int k = kb + j;
                if (k > N) break; // wave-uniform
                const WPl W = ws_seq(k);
                const SeqIn cur = nxt;
                if (k < N) seq_load<SW>(k + 1, nxt);
                double lzu[NU > 0 ? NU : 1], aux;
                if constexpr (LDSWS) {
                    aux = W.ld(P_AUX);
                    if (k < N) sfor<0, NU>([&](auto l) { lzu[l] = W.ld(P_LZU + l); });
                } else {
                    aux = cur.aux;
                    sfor<0, NU>([&](auto l) { lzu[l] = cur.lzu[l]; });
                }
                double dz;
                if (k < N) {
                    double t[NU], du[NU];
                    sfor<0, NU>([&](auto l) {
                        t[l] = lanes::bcast<AXL_LU - l>(aux) + lanes::gsum(xlane ? lzu[l] * dzx : 0.0);
                    });
                    sfor<0, NU>([&](auto qq) { // back substitution with Luu'
                        constexpr int l = NU - 1 - qq;
                        double acc = t[l];
                        sfor<l + 1, NU>([&](auto m) { acc -= lanes::bcast<m>(lzu[l]) * du[m]; });
                        du[l] = acc * lanes::bcast<l>(lzu[l]); // (the diagonal entry is stored as its reciprocal)
                    });
                    dz = xlane ? dzx : 0.0;
                    sfor<0, NU>([&](auto l) { dz = (lane == l) ? -du[l] : dz; });
                } else {
                    dz = xlane ? dzx : 0.0;
                }
                W.st(FINAL ? P_DZ : P_DZA, dz);
                mydz = (row == j) ? dz : mydz;
                if (k < N) {
                    const double dxn = mat_apply(cur.mpk, dz, cur.rb * rbscale);
                    dzx = xlane ? dxn : 0.0;
                }
            }
            wide_sync();
            // ---- row phase: row r on the rows of stage kb + r
            {
                const int kr = kb + row;
                const bool own = kr <= N;
                const int k = own ? kr : N;
                StageIn in;
                if constexpr (LDSWS) row_load<SW>(k, ws_row(k, 0, false), in);
                else {
                    in = rnx;
                    const int k2 = kr + BS <= N ? kr + BS : N;
                    if (kb + BS <= N) row_load<SW>(k2, ws_row(k2, kb + BS, false), rnx); // wave-uniform: the next block's, in flight during its recursion
                }
                const double z = in.z, aux = in.aux;
                const double dz = mydz;
                const double dza = FINAL ? in.dza : dz;
                const double zbx = aux_zx(aux), zby = aux_zy(aux);
                if constexpr (!MERGE) { // (MERGE: the box rows are rows of the chunk below)
                    BoxRow br;
                    in.aux = aux;
                    box_from(in, k, br);
                    br.act = br.act && own;
                    double Gh, gam;
                    chain<CPC>(br, z, FINAL, dza, sigmu, Gh, gam, so_c);
                    br.expand(dz);
                    q = br.blocking(q);
                    if (SUMS) {
                        ex_put(row, 4 * KC, br.act ? br.ll * br.dtl + br.tl * br.dll + br.lu * br.dtu + br.tu * br.dlu : 0.0);
                        ex_put(row, 4 * KC + 1, br.act ? br.dll * br.dtl + br.dlu * br.dtu : 0.0);
                        if constexpr (SOFTBOX) {
                            const bool sb = br.act && br.soft;
                            ex_put(row, 4 * KC + 2, sb ? br.lsl * br.dtsl + br.tsl * br.dlsl + br.lsu * br.dtsu + br.tsu * br.dlsu : 0.0);
                            ex_put(row, 4 * KC + 3, sb ? br.dlsl * br.dtsl + br.dlsu * br.dtsu : 0.0);
                        }
                    }
                }
                if constexpr (KCH > 0) {
                sfor<0, KCH>([&](auto c) {
                    ObsRow o;
                    double cx, cy, Gh2, gam2;
                    obs_from<c>(in, k, zbx, zby, o, cx, cy);
                    o.act = o.act && own;
                    const double v = rowdot<c>(cx, cy, z - pos_sel(zbx, zby), z);
                    const double w = rowdot<c>(cx, cy, dz, dz);
                    const double wa = FINAL ? rowdot<c>(cx, cy, dza, dza) : w;
                    chain<CPC>(o, v, FINAL, wa, sigmu, Gh2, gam2, so_c);
                    o.expand(w);
                    q = o.blocking(q);
                    if (SUMS) {
                        ex_put(row, 4 * c, o.act ? o.ll * o.dtl + o.tl * o.dll + o.lu * o.dtu + o.tu * o.dlu : 0.0);
                        ex_put(row, 4 * c + 1, o.act ? o.dll * o.dtl + o.dlu * o.dtu : 0.0);
                        if constexpr (SOFT) {
                            ex_put(row, 4 * c + 2, o.act ? o.lsl * o.dtsl + o.tsl * o.dlsl + o.lsu * o.dtsu + o.tsu * o.dlsu : 0.0);
                            ex_put(row, 4 * c + 3, o.act ? o.dlsl * o.dtsl + o.dlsu * o.dtsu : 0.0);
                        }
                    }
                });
                } // (KCH > 0)
            }
            if (SUMS) { // the sums for mu_aff (CPC: and of the corrected step), stage by stage as the 16-lane sweep takes them
                wide_sync();
                for (int j = 0; j < BS; j++) {
                    if constexpr (!MERGE) { s1 += ex_get(j, 4 * KC); s2 += ex_get(j, 4 * KC + 1); }
                    if constexpr (SOFTBOX) { s1 += ex_get(j, 4 * KC + 2); s2 += ex_get(j, 4 * KC + 3); }
                    if constexpr (KCH > 0) {
                        sfor<0, KCH>([&](auto c) {
                            s1 += ex_get(j, 4 * c); s2 += ex_get(j, 4 * c + 1);
                            if constexpr (SOFT) { s1 += ex_get(j, 4 * c + 2); s2 += ex_get(j, 4 * c + 3); }
                        });
                    }
                }
                wide_sync();
            }
        }
        if constexpr (!LDSWS) lanes::drain_stores(); // (rows read each other's stores in the next sweep)
        alpha = 1.0 / xwave<true>(lanes::gmax(lanes::xrow_max(q))); // q >= 1: alpha = min(1, min over blocking pairs of -v/dv)
        if (SUMS) { S1 = lanes::gsum(s1); S2 = lanes::gsum(s2); }
    }

    // ------------------------------------------------------------------ NLP residuals (full SQP only)
    // Inf-norms of the KKT residuals of the NLP at the iterate the lineariser has just been run on, with the
    // multipliers and slacks the previous QP left in the workspace (acados ocp_nlp_res_compute): because the
    // QP is the linearisation AT this iterate, they are the QP's residuals at dz = 0.  first: no QP has been
    // solved yet in this call, all multipliers / slacks are zero.  One backward sweep, no prefetching.
    USV_DEV void nlp_residual(bool first, double *res) const
    {
        double rg = 0.0, rbn = 0.0, rd = 0.0, rm = 0.0, pin = 0.0;
        for (int k = N; k >= 0; k--) {
            const Planes W = ws(k);
            StageIn in;
            const double zb = zbar(k); // the iterate itself: its residuals are the QP's at z = 0
            const double zbx = KCH > 0 ? lanes::bcast<PXL>(zb) : 0.0, zby = KCH > 0 ? lanes::bcast<PYL>(zb) : 0.0;
            in.z = zb;
            in.aux = 0.0;
            sfor<0, 4>([&](auto e) { in.box[e] = 0.0; });
            if constexpr (SOFTBOX) sfor<0, 6>([&](auto e) { in.bxs[e] = 0.0; });
            if constexpr (KCH > 0) sfor<0, KCH>([&](auto c) { sfor<0, OBSN>([&](auto e) { in.obs[c][e] = 0.0; }); });
            if (!first) { // wave-uniform
                if constexpr (PACK) {
                    in.aux = W.ld(P_AUX); // the dense box rows of the last QP (its position lanes are stale: zbx / zby above)
                } else {
                    in.box[0] = W.ld(P_BLL); in.box[1] = W.ld(P_BLU); in.box[2] = W.ld(P_BTL); in.box[3] = W.ld(P_BTU);
                    if constexpr (SOFTBOX) sfor<0, 6>([&](auto e) { in.bxs[e] = W.ld(P_BS + e); });
                }
                if constexpr (KCH > 0) {
                    if (k >= 1 && k < N) {
                        sfor<0, KCH>([&](auto c) { sfor<0, OBSN>([&](auto e) { in.obs[c][e] = W.ld(P_OBS + c * OBSN + e); }); });
                    } else if (PACK && k == 0) {
                        sfor<0, 4>([&](auto e) { in.obs[KCH - 1][e] = W.ld(P_OBS + (KCH - 1) * OBSN + e); });
                    }
                }
            }
            if constexpr (KCH > 0) {
                if (k >= 1 && k < N && !pstat) sfor<0, KCH>([&](auto c) { obs_raw<c>(k, in.raw[c]); });
            }
            double bat[NX], mpk[MP::NPK];
            if (k < N) { mat_issue(k, mpk); mat_unpack(mpk, bat); }
            else sfor<0, NX>([&](auto j) { bat[j] = 0.0; });
            // rows: with no multipliers yet lambda = t = 0 (box_from / obs_from deliver 0 / 1 for inactive rows)
            BoxRow br;
            box_from(in, k, br);
            if (first) { br.tl = 0.0; br.tu = 0.0; if constexpr (SOFTBOX) { br.tsl = 0.0; br.tsu = 0.0; } }
            if (br.act) {
                rd = fmax(rd, fmax(fabs(zb + br.sl - br.dl - br.tl), fabs(br.du - zb + br.su - br.tu)));
                rm = fmax(rm, fmax(br.ll * br.tl, br.lu * br.tu));
                if constexpr (SOFTBOX) {
                    if (br.soft) {
                        rg = fmax(rg, fmax(fabs(br.Zl * br.sl + br.zl - br.ll - br.lsl), fabs(br.Zu * br.su + br.zu - br.lu - br.lsu)));
                        rd = fmax(rd, fmax(fabs(br.sl - br.bsl - br.tsl), fabs(br.su - br.bsu - br.tsu)));
                        rm = fmax(rm, fmax(br.lsl * br.tsl, br.lsu * br.tsu));
                    }
                }
            }
            double lx = 0.0, ly = 0.0;
            if constexpr (KCH > 0) {
                sfor<0, KCH>([&](auto c) {
                    ObsRow o;
                    double cx, cy;
                    obs_from<c, false>(in, k, zbx, zby, o, cx, cy);
                    if (first) { o.tl = 0.0; o.tu = 0.0; if constexpr (SOFT) { o.tsl = 0.0; o.tsu = 0.0; } }
                    if (o.act) {
                        rd = fmax(rd, fmax(fabs(o.sl - o.dl - o.tl), fabs(o.du + o.su - o.tu)));
                        rm = fmax(rm, fmax(o.ll * o.tl, o.lu * o.tu));
                        lx += (o.ll - o.lu) * cx; ly += (o.ll - o.lu) * cy;
                        if constexpr (SOFT) {
                            rg = fmax(rg, fmax(fabs(o.Zl * o.sl + o.zl - o.ll - o.lsl), fabs(o.Zu * o.su + o.zu - o.lu - o.lsu)));
                            rd = fmax(rd, fmax(fabs(o.sl - o.bsl - o.tsl), fabs(o.su - o.bsu - o.tsu)));
                            rm = fmax(rm, fmax(o.lsl * o.tsl, o.lsu * o.tsu));
                        }
                    }
                });
                lx = lanes::gsum(lx); ly = lanes::gsum(ly);
            }
            // stationarity: g + [B A]' pi_{k+1} - C'(ll - lu) (- pi_k on the x lanes), g = H zbar - M yref
            double t = W.ld(P_GQ);
            if constexpr (HDIAG) t = fma((k < N) ? hd_stage : hd_term, zb, t);
            else {
                const double *Hrow = (k < N ? S.Hc : S.He) + lane * LANES;
                sfor<0, NZ>([&](auto c) { lanes::fma_bc<c>(t, zb, Hrow[c]); });
            }
            sfor<0, NX>([&](auto j) {
                if constexpr (!out_unit(j)) lanes::fma_bc<NU + j>(t, pin, bat[j]);
            });
            if constexpr (M::OUT_UNIT != 0u) t += ounit ? pin : 0.0;
            t -= br.act ? br.ll - br.lu : 0.0;
            t -= isPX ? lx : (isPY ? ly : 0.0);
            const double pik = (xlane && !first) ? W.ld(P_PI) : 0.0;
            if (xlane && k >= 1) rg = fmax(rg, fabs(t - pik));
            if (ulane && k < N) rg = fmax(rg, fabs(t));
            if (k < N) rbn = fmax(rbn, xlane ? fabs(W.ld(P_RB0)) : 0.0);
            if (k == 0) rbn = fmax(rbn, xlane ? fabs(P.x0[(long)b * NX + (lane - NU)] - zb) : 0.0);
            pin = pik;
        }
        res[0] = lanes::gmax(rg); res[1] = lanes::gmax(rbn); res[2] = lanes::gmax(rd); res[3] = lanes::gmax(rm);
    }

    // ------------------------------------------------------------------ multiplier read-back (usvmpc_get "lam" / "t")
    // The inequality multipliers and slacks the last QP of this row's instance left in the workspace, written in acados'
    // row order (DevSpec::box_pos ...): what ocp_nlp_out_get(.., "lam" | "t") returns after acados_solve.  Rows that do not
    // exist at a stage (state bounds and obstacle rows at stage 0, everything at stage N) stay at the zeros the host put there.
    // Not part of a solve: its own kernel (usv_qp_export), run on demand.
    USV_DEV void export_rows() const
    {
        const bool real = g < nB && live;
        const int nrow = S.nbu + S.nbx + Kn, ns = S.nsbx + (SOFT ? Kn : 0), ns0 = 2 * nrow, nlam = P.nlam;
        for (int k = 0; k <= N; k++) {
            StageIn in = {};
            load_in<SW_FWD_A>(k, in);
            const double zbx = aux_zx(in.aux), zby = aux_zy(in.aux);
            double *L = P.lam_out + ((long)b * (N + 1) + k) * nlam, *T = P.t_out + ((long)b * (N + 1) + k) * nlam;
            BoxRow br;
            box_from(in, k, br); // (lane gathers: wave-uniform control flow)
            if (real && br.act) {
                const int i = S.box_pos[lane];
                L[i] = br.ll; L[nrow + i] = br.lu; T[i] = br.tl; T[nrow + i] = br.tu;
                if constexpr (SOFTBOX) {
                    if (br.soft) {
                        const int j = S.sbx_pos[lane];
                        L[ns0 + j] = br.lsl; L[ns0 + ns + j] = br.lsu; T[ns0 + j] = br.tsl; T[ns0 + ns + j] = br.tsu;
                    }
                }
            }
            if constexpr (KCH > 0) {
                sfor<0, KCH>([&](auto c) {
                    ObsRow o;
                    double cx, cy;
                    obs_from<c, false>(in, k, zbx, zby, o, cx, cy);
                    const int i = S.nbu + S.nbx + c * LANES + lane;
                    if (real && o.act) {
                        L[i] = o.ll; L[nrow + i] = o.lu; T[i] = o.tl; T[nrow + i] = o.tu;
                        if constexpr (SOFT) {
                            const int j = S.nsbx + c * LANES + lane;
                            L[ns0 + j] = o.lsl; L[ns0 + ns + j] = o.lsu; T[ns0 + j] = o.tsl; T[ns0 + ns + j] = o.tsu;
                        }
                    }
                });
            }
        }
    }

    // ------------------------------------------------------------------ results of a finished QP
    // RTI step and outputs of the rows selected by `fin` (acados ocp_nlp_update_variables; status 4 leaves the iterate
    // untouched).  real: the row holds an instance of the batch that this launch is solving.
    USV_DEV void finish(bool fin, bool real, int status, int iters, int phase)
    {
        const bool ok = (status == 0 || status == 1);
        const bool out = fin && real;
        const bool share = P.epoch != nullptr; // wave-uniform (kernel argument)
        double tmin = 1e300; // smallest t_l over this instance's obstacle rows: how close the QP solution sits to a keep-out circle
        for (int k = 0; k <= N; k++) {
            const Planes W = ws(k);
            const double z = W.ld(P_Z); // zbar + z: the new iterate
            if constexpr (KCH > 0) {
                if (k >= 1 && k < N) { // wave-uniform
                    sfor<0, KCH>([&](auto c) {
                        const double tl = W.ld(P_OBS + c * OBSN + 2);
                        tmin = (c * LANES + lane < Kn) ? fmin(tmin, tl) : tmin;
                    });
                }
            }
            if constexpr (LDSWS) {
                // the multipliers and slacks of this QP go back to the group's planes in HBM: that is where the multiplier
                // read-back (export_rows) and the residual test of a later full SQP (nlp_residual) look for them
                if (out) {
                    const lanes::Planes G = wsg(k);
                    G.st(P_AUX, W.ld(P_AUX));
                    G.st(P_PI, W.ld(P_PI));
                    if constexpr (!PACK) {
                        G.st(P_BLL, W.ld(P_BLL)); G.st(P_BLU, W.ld(P_BLU)); G.st(P_BTL, W.ld(P_BTL)); G.st(P_BTU, W.ld(P_BTU));
                        if constexpr (SOFTBOX) sfor<0, 6>([&](auto e) { G.st(P_BS + e, W.ld(P_BS + e)); });
                    }
                    if constexpr (KCH > 0) sfor<0, KCH * OBSN>([&](auto e) { G.st(P_OBS + e, W.ld(P_OBS + e)); });
                }
            }
            if constexpr (AUXLDS) { // (the read-back paths and a later full SQP find the dense box rows in the HBM plane)
                if (out) W.st(P_AUX, aux_ld(k, W));
            }
            if (out) {
                // (with a consumer waiting in another kernel - the next tick's lineariser - the iterate goes out past the L2)
                if (share) {
                    if (ok && xlane) lanes::st_shared(&P.x[((long)b * (N + 1) + k) * NX + (lane - NU)], z);
                    if (ok && ulane && k < N) lanes::st_shared(&P.u[((long)b * N + k) * NU + lane], z);
                } else {
                    if (ok && xlane) P.x[((long)b * (N + 1) + k) * NX + (lane - NU)] = z;
                    if (ok && ulane && k < N) P.u[((long)b * N + k) * NU + lane] = z;
                }
                if (k >= 1 && xlane && P.pi) P.pi[((long)b * N + (k - 1)) * NX + (lane - NU)] = W.ld(P_PI);
            }
            if constexpr (KCH > 0 && SOFT) {
                if (k >= 1 && k < N) {
                    sfor<0, KCH>([&](auto c) {
                        const int i = c * LANES + lane;
                        if (out && i < Kn && P.sl) {
                            const int p0 = P_OBS + c * OBSN;
                            P.sl[((long)b * N + k) * Kn + i] = W.ld(p0 + 4);
                            P.su[((long)b * N + k) * Kn + i] = W.ld(p0 + 5);
                        }
                    });
                } else if (k == 0) { // wave-uniform
                    // soft rows of stage 0 (see init): x_0 = x0 fixes their value, the slacks minimise their own penalty
                    const double aux = aux_ld(0, W);
                    const double zbx = aux_zx(aux), zby = aux_zy(aux);
                    const double e0 = z - pos_sel(zbx, zby); // x0 - xbar_0 on the position lanes
                    sfor<0, KCH>([&](auto c) {
                        const int i = c * LANES + lane;
                        double raw[3], d, ux, uy;
                        obs_raw<c>(0, raw);
                        obs_dist(zbx - raw[0], zby - raw[1], d, ux, uy);
                        const double v0 = ux * lanes::bcast<PXL>(e0) + uy * lanes::bcast<PYL>(e0);
                        double a = fmax(c_bsl[c], raw[2] - d - v0), q = fmax(c_bsu[c], d + v0 - c_uh[c]);
                        if (c_Zl[c] > 0.0) a = fmax(a, -c_zl[c] / c_Zl[c]);
                        if (c_Zu[c] > 0.0) q = fmax(q, -c_zu[c] / c_Zu[c]);
                        if (out && i < Kn && P.sl) {
                            P.sl[(long)b * N * Kn + i] = a;
                            P.su[(long)b * N * Kn + i] = q;
                        }
                    });
                }
            }
        }
        tmin = lanes::gmin(tmin);
        if (share) { // the instance's iterate is final: tell the lineariser of the next tick (every store of the wave has landed first)
            lanes::drain_stores();
            if (out && lane == 0) lanes::publish(P.epoch + b, P.tick);
        }
        if (out && lane == 0) {
            if (P.obs_tmin) P.obs_tmin[b] = tmin;
            if (!ok && P.fail_count) lanes::count_one(P.fail_count);
            P.status[b] = ok ? 0 : 4;
            P.qp_iter[b] = iters;
            P.qp_status[b] = status;
            if (phase > 0) {
                P.sqp_iter[b] += 1;
                if (!ok) P.sqp_state[b] = 4;
                else lanes::count_one(P.sqp_running);
            }
        }
    }

    // ------------------------------------------------------------------ hand-over of long runners
    // A persistent launch ends with a drain: the instances in flight when its queue runs dry finish one by one while more and more of the
    // device idles (profiles/r03_tail.txt, r05_handover.txt).  The idea: a faster pass for ONE instance - the latency mapping - for the ones
    // that run longest.  (Built and measured in round 5: over COLD planes in HBM the one-instance-per-wave sweeps are slower per pass than the
    // lone row they relieve; with the planes copied into LDS first - copy_in below - they are twice as fast.  Round 6: under the default QP solver
    // profile the hand-over pays for batches of a few thousand instances, with the follow-up kernel running beside the launch - usvmpc.hip
    // launch_qp has the policy; option "handover_iter".)  Once every instance of the launch has been handed out (the queue counter has passed the batch), a row whose
    // instance has done handover_iter iterations leaves it where it stands: everything an iteration hands to the next is in the
    // workspace planes already (the pending step in P_DZ / P_DZA, multipliers and slacks in the row planes, the iterate in P_Z); what
    // lives in registers - step length and centring target of the pending step, the residual scale, the iteration count - goes into
    // the instance's record, the group into the list.  The follow-up launch (usvmpc.hip: usv_qp_resume) picks the list up one instance
    // per wave and carries on from exactly that state, on the same planes: the mappings return the same bits, so WHERE an instance
    // is finished does not show in its result (scheduling only; tests/test_gpu_handover.py, emulator tests/test_wide_emu.py).
    // The resuming wave with its planes in LDS: everything the suspended solve has left in the group's planes in HBM comes in first (the planes of
    // the LDS map, stage by stage; all rows load, row 0 stores) - one pass over cold planes, after which a pass costs what it costs in LDS.
    USV_DEV void copy_in() const
    {
        if constexpr (WIDE && LDSWS) {
            for (int k = 0; k <= N; k++) {
                const lanes::Planes G = wsg(k);
                const Planes W = ws(k);
                double v[WL::NPT];
                sfor<0, WL::NPT>([&](auto p) { if constexpr (WideMap::at(p) >= 0) v[p] = G.ld(p); });
                sfor<0, WL::NPT>([&](auto p) { if constexpr (WideMap::at(p) >= 0) W.st(p, v[p]); });
            }
            wide_sync();
        }
    }
    static constexpr bool HAS_CPC = CPC;
    static constexpr bool CAN_SUSPEND = !WIDE && !LDSWS && HDIAG && !SOFTBOX && ((PACK && KCH == 1) || (!PACK && KCH == 0));
    USV_DEV void suspend(bool sel, int it, double a_prev, double sig_prev)
    {
        if constexpr (AUXLDS) { // (the aux plane of this launch lives in the wave's LDS: the resuming wave reads the HBM plane)
            for (int k = 0; k <= N; k++) {
                const Planes W = ws(k);
                const double a = aux_ld(k, W);
                if (sel) W.st(P_AUX, a);
            }
        }
        if (sel && lane == 0) {
            double *r = P.susp_rec + 4 * b;
            // (CPC: whether the pending step is a centring-only one rides in the sign of the iteration count, which is at least 1 here)
            double itv = (double)it;
            if constexpr (CPC) itv = ((double)so_prv == 0.0) ? -itv : itv;
            r[0] = a_prev; r[1] = sig_prev; r[2] = rbscale; r[3] = itv;
        }
        // With the follow-up kernel running beside this launch (co_ctl) the consumer may sit on another XCD and starts as soon as the entry
        // appears: everything this wave has stored for the instance - planes, record - leaves its L2 first.  (Wave-uniform: kernel argument.)
        const bool co = P.co_ctl != nullptr;
        if (co) lanes::release_agent();
        if (sel && lane == 0) {
            const int slot = lanes::fetch_add(P.susp_count);
            if (co) lanes::publish(P.susp_list + slot, (int)g);
            else P.susp_list[slot] = (int)g;
        }
    }

    // ------------------------------------------------------------------ driver
    // phase 0: one SQP-RTI iteration.  phase 1 / 2: one iteration of the full SQP (first / later): test the NLP
    // residuals, and unless the instance has converged (or finished earlier) solve the QP and take the step.
    // phase 3 (the latency mapping over planes in HBM): carry on with an RTI solve that another launch has suspended.
    // queue0 >= 0: the rows of this wave started on groups below queue0 and take further groups queue0, queue0 + 1, ...
    // from the counter P.queue as they finish (phase 0 only); queue0 < 0: every row keeps its first group.
    USV_DEV void solve(int phase, int queue0)
    {
        bool frozen = false;
        keep = false;
        const bool resume = phase == 3;
        if (resume) phase = 0;
        if (phase > 0) {
            const bool real0 = g < nB;
            frozen = !real0 || P.sqp_state[b] >= 0;
            if (!lanes::wave_any(!frozen)) return;
            double nr[4];
            nlp_residual(phase == 1, nr);
            if (!frozen) {
                const bool conv = nr[0] <= S.nlp_tol[0] && nr[1] <= S.nlp_tol[1] && nr[2] <= S.nlp_tol[2] && nr[3] <= S.nlp_tol[3];
                if (lane == 0) {
                    P.nlp_res[b * 4 + 0] = nr[0]; P.nlp_res[b * 4 + 1] = nr[1];
                    P.nlp_res[b * 4 + 2] = nr[2]; P.nlp_res[b * 4 + 3] = nr[3];
                    if (conv) P.sqp_state[b] = 0;
                }
                frozen = conv;
            }
            if (!lanes::wave_any(!frozen)) return;
        }
        keep = frozen;
        const bool bad0 = resume ? false : init(true);
        // ---- per-row state of the IPM (every row is in its own iteration)
        rbscale = 1.0;
        const bool own = WIDE ? true : live;     // (WIDE: all four rows iterate on the wave's instance, row 0 - live - writes)
        bool real = g < nB && !frozen && live;   // the row holds an instance whose results are to be written
        bool done = frozen || !own;              // nothing (more) to iterate on in this row
        bool pend = false;               // a step of the previous iteration is waiting to be applied
        bool fresh = false;              // cold-started after this pass's factorisation sweep: sits out the rest of the pass
        bool late = bad0 && !frozen && own;      // stopped after the factorisation sweep (step-length floor), or never started
                                         // (x0 inside a hard keep-out circle): results at the next pass
        int status = late ? 4 : 1, iters = 0, it = 0;
        done = done || late;
        Norms nm;
        double a_prev = 0.0, sig_prev = 0.0;
        const double nc = (double)S.nc;
        const bool refill = phase == 0 && queue0 >= 0; // wave-uniform
        // option "cond_pred_corr" (on in every profile but "r04"): the test is built into every kernel, the option switches it (off: no step is
        // ever refused, every so stays 1 and the sweeps return the bits of sweeps without it)
        const bool cpc = CPC && lanes::uniform(S.cpc) != 0;
        if constexpr (HAS_CPC) { so_prv = 1.0; so_cur = 1.0; }
        if constexpr (WIDE) {
            if (resume) { // the state the suspending row left (all rows of the wave read the same record)
                copy_in();
                const double *r = P.susp_rec + 4 * b;
                const double r3 = r[3];
                a_prev = r[0]; sig_prev = r[1]; rbscale = r[2];
                it = (int)fabs(r3);
                if constexpr (CPC) so_prv = (r3 < 0.0) ? 0.0 : 1.0;
                iters = it;
                pend = true;
            }
        }
        // (hand-over: kernel argument, wave-uniform; only launches that refill from a queue have a tail worth handing over)
        const int hand_it = (CAN_SUSPEND && phase == 0 && P.susp_count != nullptr) ? P.handover_iter : 0;
        for (;;) {
            backward<true>(nm, 0.0, pend && !done, a_prev, sig_prev);
            bool fin = late;
            if (!done) {
                // the residuals of the iterate this pass has just evaluated; the last ones written are the final ones
                if (real && lane == 0) {
                    P.res[b * 4 + 0] = nm.rg; P.res[b * 4 + 1] = nm.rb; P.res[b * 4 + 2] = nm.rd; P.res[b * 4 + 3] = nm.rm;
                }
                iters = it;
                if (nm.nan != nm.nan) { status = 3; fin = true; }
                else if (nm.rg <= S.tol_stat && nm.rb <= S.tol_eq && nm.rd <= S.tol_ineq && nm.rm <= S.tol_comp) {
                    status = 0; fin = true;
                } else if (it >= itmax) { status = 1; fin = true; }
            }
            done = done || fin;
            late = false;
            if (lanes::wave_any(fin)) { // wave-uniform
                finish(fin, real, status, iters, phase);
                real = fin ? false : real;
                bool took = false;
                if (refill) {
                    // one ticket per finished row; beyond the batch there is nothing left and the row stays idle
                    int gn = 0;
                    if constexpr (WIDE) { // one ticket for the wave
                        if (fin && lanes::block_row() == 0u && lane == 0) gn = queue0 + lanes::fetch_add(P.queue);
                        gn = xwave_first_i(lanes::wave_first_i(gn));
                    } else {
                        if (fin && lane == 0) gn = queue0 + lanes::fetch_add(P.queue);
                        gn = lanes::bcast_i<0>(gn);
                    }
                    const bool take = fin && gn < nB;
                    took = take;
                    if (lanes::wave_any(take)) { // wave-uniform
                        if constexpr (WW > 1) lanes::block_sync(); // (the parked constants are the workgroup's: no wave may still be reading the old instance's)
                        bind((long)gn, take);
                        const bool bad = init(take) && take;
                        real = take ? (WIDE ? live : true) : real;
                        done = take ? bad : done;
                        late = bad;
                        fresh = take;
                        pend = take ? false : pend;
                        rbscale = take ? 1.0 : rbscale;
                        it = take ? 0 : it;
                        iters = take ? 0 : iters;
                        status = take ? (bad ? 4 : 1) : status;
                        if constexpr (HAS_CPC) { const double sp = so_prv; so_prv = take ? 1.0 : sp; }
                        // (the freshly started row sits out the three remaining sweeps of this pass: parked meanwhile - see below - so that
                        // it does not stream its new group's planes for nothing; back at the end of the pass)
                        if constexpr (!LDSWS && !WIDE) voff = take ? stage_bytes + (unsigned)(lane * 8) : voff;
                    }
                }
                // A row that has nothing more to do in this launch stops streaming.  The sweeps run in lock step for the four rows of the
                // wave whatever their state, plane loads and stores included - and the launch ends with a drain of ~13 ms in which more
                // and more rows are idle beside rows that still iterate (profiles/r05_handover.txt).  The idle row's offset goes to the
                // end of the stage window: the buffer range check then answers its loads with 0 and drops its stores, no memory access
                // is made.  (RTI launches over planes in HBM; its results are out, nothing of its group is needed again.)
                if constexpr (!LDSWS && !WIDE) {
                    if (phase == 0) voff = (fin && !took) ? stage_bytes + (unsigned)(lane * 8) : voff;
                }
            }
            // (a row that has just taken an instance whose x0 sits inside a hard keep-out circle is done AND still owes its results: late)
            if (!lanes::wave_any(!done || late)) break;
            const bool run = !done && !fresh; // rows that take part in the rest of this pass
            const double mu = nc > 0.0 ? nm.musum / nc : 0.0;
            double a_aff = 1.0, S1 = 0.0, S2 = 0.0, a = 1.0, d1 = 0.0, d2 = 0.0;
            forward<false>(0.0, a_aff, S1, S2);
            double sigmu = 0.0, mu_aff = 0.0;
            if (nc > 0.0) {
                mu_aff = (nm.musum + a_aff * S1 + a_aff * a_aff * S2) / nc;
                const double sg = mu_aff / mu;
                sigmu = sg * sg * sg * mu;
            }
            if constexpr (HAS_CPC) so_cur = 1.0;
            backward<false>(nm, sigmu, false, 0.0, 0.0);
            forward<true>(sigmu, a, d1, d2);
            if constexpr (HAS_CPC) {
                // HPIPM's conditional predictor-corrector (d_ocp_qp_ipm_arg.cond_pred_corr, on in its SPEED / BALANCE / ROBUST modes; as
                // recalled - DESIGN.md section 2): a corrected step that leaves the duality measure above cpc_factor x the predictor's
                // mu_aff is refused and the centring-only step (the corrector's target without its second-order term) taken instead.
                // Per row; the sweeps are the wave's, so the rows that keep their step recompute it (so = 1: the same bits).  (WIDE: the
                // rows and waves of a workgroup hold one instance and the same values - the branch is uniform over the workgroup.)
                const double mu_pc = nc > 0.0 ? (nm.musum + a * d1 + a * a * d2) / nc : 0.0;
                const bool refuse = cpc && run && nc > 0.0 && mu_pc > S.cpc_factor * mu_aff;
                if (lanes::wave_any(refuse)) { // wave-uniform
                    so_cur = refuse ? 0.0 : 1.0; // (LDS operations of a wave execute in order: the sweeps below read what is written here)
                    double a2 = 1.0;
                    backward<false>(nm, sigmu, false, 0.0, 0.0);
                    forward<true>(sigmu, a2, d1, d2);
                    a = refuse ? a2 : a;
                }
            }
            if (run && a < S.alpha_min) { status = 2; done = true; late = true; iters = it; }
            a_prev = run ? a * ((1.0 - a) * 0.99 + a * 0.9999999) : a_prev;
            sig_prev = run ? sigmu : sig_prev;
            if constexpr (HAS_CPC) { const double sp = so_prv, sc = so_cur; so_prv = run ? sc : sp; }
            pend = run ? true : pend;
            it = run ? it + 1 : it;
            if constexpr (!LDSWS && !WIDE) voff = fresh ? lanes::Planes::lane_offset(g, NPL, lane) : voff;
            fresh = false;
            if constexpr (CAN_SUSPEND) {
                if (hand_it > 0) { // wave-uniform
                    // every instance of the launch has been handed out (a launch without a queue: from the start)
                    const bool drained = !refill || lanes::observe(P.queue) >= nB - queue0;
                    const bool sus = drained && !done && real && it >= hand_it;
                    if (lanes::wave_any(sus)) { // wave-uniform
                        suspend(sus, it, a_prev, sig_prev);
                        done = done || sus;
                        real = sus ? false : real;
                        voff = sus ? stage_bytes + (unsigned)(lane * 8) : voff; // (parked: see above - the follow-up launch owns the planes now)
                    }
                }
            }
        }
    }
};

} // namespace usv

#pragma clang fp contract(fast)
