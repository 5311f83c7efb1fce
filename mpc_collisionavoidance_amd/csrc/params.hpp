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
};

// Device pointers of one solver handle.
// Which copy of the stage matrix the forward sweeps stream: the rows of [B A] (a second, transposed set
// of planes written by the lineariser; one plane per column that is not a unit vector) or the [B A]'
// planes of the backward sweeps (one per row that is not a unit vector, reduced across the lanes).
// The cheaper one in bytes wins; on a tie the single copy.
template <class M>
constexpr bool fwd_rows()
{
    return (M::NX + M::NU - __builtin_popcount(M::IN_UNIT)) < (M::NX - __builtin_popcount(M::OUT_UNIT));
}

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
    int *qp_status;       // [B]   0 ok, 1 max iter, 2 min step, 3 nan
    double *res;          // [B][4]      final QP residuals (stat, eq, ineq, comp)
    // linearisation output, lane-major planes: element (k, e) of group g, lane r at
    // ((k*E + e) * Bp + g) * 16 + r
    double *BAt;          // [N][nx]   row r of [B A]'   (lane r = variable r of [u;x])
    double *ABr;          // [N][nz]   row j of [B A] (lane nu+j = state j); only if fwd_rows<M>(), else nullptr
    double *rb0;          // [N]       dynamics residual b_k (x lanes)
    double *gq;           // [N+1]     cost gradient
    // QP workspace, lane-major planes [N+1][NPL]
    double *ws;
};

} // namespace usv
