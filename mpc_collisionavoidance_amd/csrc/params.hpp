// params.hpp — plain-data structs shared by the host C-ABI layer and the device kernels.
#pragma once

namespace usv {

constexpr int LANES = 16;   // lanes per OCP instance (one DPP row)
constexpr int KMAX = 32;    // obstacle rows per stage (2 chunks of 16 lanes)

// Constants of one OCP definition (identical for every instance of the batch). Lives in device
// memory; row-indexed tables are read per lane.
struct DevSpec {
    double Hc[LANES * LANES];  // dt * [Vu Vx]' W [Vu Vx], [u;x] ordering, 16x16 zero padded
    double He[LANES * LANES];  // terminal Vx_e' W_e Vx_e placed at the x rows/cols
    double Mc[LANES * LANES];  // dt * [Vu Vx]' W        (nz x ny)
    double Me[LANES * LANES];  // Vx_e' W_e at the x rows (nz x ny_e)
    double lb[LANES], ub[LANES];  // box bounds per variable of [u;x]
    int has_b[LANES];             // 1: variable carries a box constraint
    // Box rows packed into the last obstacle chunk's (lambda, t) planes + at most one dense plane (qp_ipm.hpp):
    int boxpack;                  // 1: packing in use (option "pack_box_rows" toggles it when boxpack_ok)
    int boxpack_ok;               //    whether the rows fit at all
    int box_dense;                // 1: some rows are dense (plane P_BLL is in use)
    int aux_dense4;               // 4 x the number of dense rows: they occupy lanes 0 .. aux_dense4-1 of the aux plane
    int box_slot[LANES];          // variable r -> lane its value 0 is stored in
    int box_step[LANES];          // variable r -> 0 (slot row: all four values in that lane) | 1 (dense row)
    int slot_var[LANES];          // lane L -> variable whose row (or row element) it stores
    int slot_is[LANES];           // lane L: 0 nothing, 1 slot lane, 2 dense lane
    double uh[KMAX];
    double lsl[KMAX], lsu[KMAX];  // lower bounds of the soft slacks (lsh, ush)
    double zl[KMAX], zu[KMAX], Zl[KMAX], Zu[KMAX];  // slack penalties, already scaled by dt
    double dt;
    int N, K, B, Bp;              // horizon, obstacles, batch, batch padded to a multiple of 4
    int ny, ny_e;
    int nc;                       // number of (lambda, t) pairs of the whole QP
    int p_static;                 // 1: every stage uses the obstacle set / lh of stage 0 (the ROS node's usage)
    int hdiag;                    // 1: Hc and He are diagonal (true for every OCP of the reference)
    int iter_max;
    double mu0, thr0, tol_stat, tol_eq, tol_ineq, tol_comp, alpha_min;
    int sim_steps;                // RK4 steps per shooting interval (sim_method_num_steps)
    int npt;                      // workspace planes per stage (WsLayout::NPT of the QP kernel in use)
    // soft state bounds (acados idxsbx): per variable of [u;x], penalties already scaled by dt
    int any_bsoft;
    int bsoft[LANES];
    double b_zl[LANES], b_zu[LANES], b_Zl[LANES], b_Zu[LANES], b_lsl[LANES], b_lsu[LANES];
    double nlp_tol[4];            // full SQP: exit tolerances on the NLP residuals (stat, eq, ineq, comp)
    // multiplier read-back (usvmpc_get "lam" / "t"): rows of a stage in acados' order [bu.., bx.., h..], nrow = nbu + nbx + K;
    // slack rows [sbx.., sh..], ns = nsbx + (soft ? K : 0); a stage's vector is [lower(nrow) | upper(nrow) | lower slack(ns) | upper slack(ns)]
    int nbu, nbx, nsbx;
    int box_pos[LANES];           // variable r of [u;x] -> position of its row in [bu.., bx..] (only where has_b)
    int sbx_pos[LANES];           // variable r -> position of its slack pair among the soft state bounds (only where bsoft)
    // option "cond_pred_corr" (HPIPM's conditional predictor-corrector; qp_ipm.hpp QpIpm::solve): 0 off (default)
    int cpc;
    double cpc_factor;            // 2.0
};

// How the stage matrix [B A] (nx x nz) is kept in HBM.  Only entries that carry information are stored: the model
// states which entries of the discrete sensitivities are structurally non-zero (M::SENS[j]: bit c <=> d x+_j / d z_c,
// z = [u;x], can be non-zero and is not the exact unit diagonal; M::DIAG_ONE bit j <=> d x+_j / d x_j == 1 exactly) -
// the pattern of (I + J + J^2 + ...)[Ju | I] for the continuous Jacobian J, valid for any explicit RK scheme and any
// number of steps.  Everything else is an exact 0 (or the exact 1 of DIAG_ONE) and is rebuilt from the pattern.
// The stored entries form one stream, row after row, within a row by increasing variable index, 16 per plane: entry
// number e sits in plane e / 16, lane e % 16.  The lineariser packs with lane gathers, the sweeps unpack the same way
// (qp_ipm.hpp).  usv_model_pf_ca: 71 of 14 x 16 entries = 5 planes (a plane per row: 14; dense 11 x 9 block: 7).
template <class M>
struct MatPack {
    static constexpr int NX = M::NX, NU = M::NU, NZ = NX + NU;
    static constexpr unsigned row_mask(int j) { return M::SENS[j]; }
    static constexpr int count(int j) { return __builtin_popcount(M::SENS[j]); }
    static constexpr int start(int j) // stream position of row j's first entry
    {
        int s = 0;
        for (int i = 0; i < j; i++) s += __builtin_popcount(M::SENS[i]);
        return s;
    }
    static constexpr bool diag_one(int j) { return ((M::DIAG_ONE >> j) & 1u) != 0u; }
    static constexpr unsigned col_mask() // variables that appear in some row's pattern
    {
        unsigned m = 0;
        for (int j = 0; j < NX; j++) m |= M::SENS[j];
        return m;
    }
    static constexpr int NE = start(NX);                                // stored entries per stage
    static constexpr int NPK = (NE + 15) / 16;                          // planes per stage
    static constexpr int nth(unsigned mask, int i)                      // position of the i-th set bit
    {
        for (int b = 0; b < 32; b++)
            if ((mask >> b) & 1u) {
                if (i == 0) return b;
                i--;
            }
        return 0;
    }
    static constexpr int rank(unsigned mask, int b) { return __builtin_popcount(mask & ((1u << b) - 1u)); }
    // consistency of the coarse traits with the pattern: a unit row stores nothing and has the unit diagonal; a unit
    // column appears in no row
    static constexpr bool consistent()
    {
        unsigned cols = 0;
        for (int j = 0; j < NX; j++) {
            cols |= M::SENS[j];
            if (((M::OUT_UNIT >> j) & 1u) && (M::SENS[j] != 0u || !diag_one(j))) return false;
            if (((M::SENS[j] >> (NU + j)) & 1u) && diag_one(j)) return false;
            if (M::SENS[j] >> NZ) return false;
        }
        for (int c = 0; c < NZ; c++)
            if (((M::IN_UNIT >> c) & 1u) && (((cols >> c) & 1u) || (c >= NU && !diag_one(c - NU)))) return false;
        return true;
    }
    static_assert(consistent(), "OUT_UNIT / IN_UNIT disagree with SENS / DIAG_ONE");
};

// Plane map of one stage of the solver workspace `ws` (lane-major planes, one window per stage).  The
// lineariser writes the last planes (dynamics residual, cost gradient, packed [B A]); everything lives in ONE
// window per stage so that the QP kernel needs one buffer descriptor and compile-time plane numbers (separate
// arrays cost scalar registers, and once those ran out the compiler moved plane offsets to vector registers
// and wrapped the loads in waterfall loops).
// planes of the exchange area of the wide mapping (qp_ipm.hpp, WIDE: [4 rows][WIDE_EX_PLANES][16 lanes] behind the instance's planes in LDS)
// (kch obstacle chunks per stage: the complementarity sums of a stage are handed over chunk by chunk - qp_ipm.hpp EX_*)
// (softbox: soft state bounds - their slack pairs' sums travel in planes of their own)
constexpr int wide_ex_fwd(int kch, bool softbox) { return 4 * (kch > 1 ? kch : 1) + 2 + (softbox ? 2 : 0); }
constexpr int wide_ex_planes(int kch, bool softbox = false)      // ... with the solver's planes in LDS
{
    const int z = 4 + 2 * (kch > 1 ? kch : 1) + 1 + (softbox ? 1 : 0);
    return z > wide_ex_fwd(kch, softbox) ? z : wide_ex_fwd(kch, softbox);
}
constexpr int wide_ex_planes_hbm(int kch, bool softbox = false)  // ... in HBM (two more: qp_ipm.hpp)
{
    const int a = 4 + 2 * (kch > 1 ? kch : 1) + 1 + (softbox ? 1 : 0) + 2;
    return a > wide_ex_fwd(kch, softbox) ? a : wide_ex_fwd(kch, softbox);
}

template <class M, int KCH, bool SOFT, bool SOFTBOX = false>
struct WsLayout {
    // P_Z   : the QP iterate in absolute form, zbar + z (what the box rows and the Hessian product need); the step z of an
    //         obstacle row's position lanes is recovered with the linearisation point kept in P_AUX
    // P_AUX : the small per-stage items, so that no sweep loads a whole plane for two values -
    //         lanes 0..: box rows stored densely (four values per row, when packing is on: host_spec.hpp),
    //         lane AXL_ZX / AXL_ZY: position (px, py) of the linearisation point (obstacle-row geometry),
    //         lane AXL_RG - l: stationarity residual of control l, lane AXL_LU - l: gain right-hand side of control l
    // P_PB  : P_{k+1} b_k (x lanes);  P_PI : dynamics multiplier pi_k (x lanes);  P_DX0 (stage 0): x0
    enum : int { P_Z = 0, P_AUX, P_DZA, P_DZ, P_DX0, P_PB, P_PI, P_BLL, P_BLU, P_BTL, P_BTU, P_OBS };
    enum : int { AXL_ZX = 15, AXL_ZY = 14, AXL_RG = 13, AXL_LU = 11, AUX_DENSE_LANES = 8 };
    static constexpr int OBSN = SOFT ? 10 : 4;
    static constexpr int P_LZU = P_OBS + KCH * OBSN;
    static constexpr int P_RB0 = P_LZU + M::NU; // b_k of the linearisation point (x lanes)
    static constexpr int P_GQ = P_RB0 + 1;      // cost gradient
    static constexpr int P_MAT = P_GQ + 1;      // packed [B A] (MatPack<M>::NPK planes)
    static constexpr int P_BS = P_MAT + MatPack<M>::NPK; // soft state bounds only: sl, su, lsl, lsu, tsl, tsu of the box rows
    static constexpr int NPT = P_BS + (SOFTBOX ? 6 : 0);  // (DevSpec::npt at run time: the lineariser does not know SOFTBOX)
};

// Device pointers of one solver handle.
struct DevPtrs {
    const DevSpec *spec;
    const int *perm;      // [B] group -> instance (difficulty binning); nullptr = identity
    // caller-visible arrays, natural layout, instance-major
    double *x;            // [B][N+1][nx]
    double *u;            // [B][N][nu]
    const double *x0;     // [B][nx]
    const double *yref;   // [B][N][ny]
    const double *yref_e; // [B][ny_e]
    const double *p;      // [B][N+1][2K]
    const double *lh;     // [B][N][K]
    double *sl, *su;      // [B][N][K]   soft slack values of the last QP
    double *pi;           // [B][N][nx]  dynamics multipliers pi_1..pi_N of the last QP
    int *status;          // [B]
    int *qp_iter;         // [B]
    int *qp_status;       // [B]   0 ok, 1 max iter, 2 min step, 3 nan, 4 x0 violates a hard stage-0 obstacle row
    double *res;          // [B][4]      final QP residuals (stat, eq, ineq, comp)
    double *obs_tmin;     // [B]         smallest lower-side slack t_l over the obstacle rows of the last QP (1e300 without rows)
    int *fail_count;      // [1]         instances of THIS launch whose solve ended with status != 0 (host points it at a ring slot)
    // full SQP (usvmpc_solve_sqp): per-instance state between the iterations of one call
    double *nlp_res;      // [B][4]      NLP residuals of the current iterate (stat, eq, ineq, comp)
    int *sqp_iter;        // [B]         SQP iterations taken
    int *sqp_state;       // [B]         -1 running, else the final acados status (0 converged, 2 max iter, 4 QP failure)
    int *sqp_running;     // [1]         instances still running after the last launch
    // solver workspace: lane-major planes [N+1 stages][Bp groups][WsLayout::NPT planes][16 lanes] - element e of stage k,
    // group g, lane r at (((k * Bp + g) * NPT + e) * 16 + r: linearisation output + QP state
    double *ws;
    int *queue;           // [1]  groups handed out beyond the waves' first four (work queue of the QP kernel)
    // pipelined lineariser (usvmpc.hip, option "pipeline_linearize"): per-instance hand-over between the QP launch of tick t and the
    // lineariser of tick t + 1 running in its tail on a second stream
    int *epoch;           // [B]  tick whose results (x, u) are final for the instance; nullptr: no hand-over in this launch
    int *redo;            // [B][redo_words]  bit k: the speculative lineariser skipped stage k of the instance: the fix-up pass does it
    int redo_words;       //      (N + 1 + 31) / 32
    const int *perm_cur;  // [B]  speculative lineariser: the group -> instance map of the QP launch that is still running
    int tick;             // the solve this launch belongs to (QP, fix-up) / whose results the speculative lineariser waits for
    // multiplier read-back (kernel usv_qp_export): [B][N+1][nlam] each, nlam = 2 (nrow + ns) - DevSpec
    double *lam_out, *t_out;
    int nlam;
    // hand-over of long runners (qp_ipm.hpp QpIpm::suspend / resume; usvmpc.hip launch_qp, option "handover_iter"): once the queue of a
    // launch is empty, a row whose instance has passed handover_iter IPM iterations leaves it - state in the workspace planes, the
    // scalars of the iteration in susp_rec - to the follow-up launch, which finishes it on the latency mapping (one instance per wave)
    int *susp_count;      // [1]     instances suspended by this launch; nullptr: no hand-over
    int *susp_list;       // [B]     their groups
    double *susp_rec;     // [B][4]  per instance: step length and centring target of the pending step, residual scale, iterations done
    int handover_iter;
    // ... with the follow-up kernel running BESIDE the draining launch (usvmpc.hip usv_qp_resume_co, option "handover_co"): entries are
    // published one by one (susp_list starts at -1 everywhere; a consumer claims entry i by compare-and-swap to -2 - group), the planes and
    // the record go out past the suspending wave's L2 first (the XCDs' L2s are not coherent: agent-scope release before the entry appears,
    // agent-scope acquire in the consumer).  co_ctl: [0] tickets taken, [1] the main launch has ended, [2] main workgroups that have started,
    // [3] entries the co-resident kernel finished, [4] polls that ran into the spin limit (a bounded wait: what it leaves is done by the
    // launch behind the main one); nullptr: entries are only read after the launch has ended.
    int *co_ctl;
};

} // namespace usv
