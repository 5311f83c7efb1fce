// host_spec.hpp — host-side translation of the caller's OCP description (usvmpc_desc, i.e. what
// an AcadosOcp carries) into the constant tables the kernels read (DevSpec).
#pragma once
#include "../../include/usvmpc.h"
#include "params.hpp"
#include <cstring>
#include <cmath>
#include <string>

namespace usv {

inline int model_dims(int model, int &nx, int &nu)
{
    switch (model) {
    case USVMPC_MODEL_USV: nx = 5; nu = 2; return 0;
    case USVMPC_MODEL_GUIDANCE_CA1: nx = 8; nu = 1; return 0;
    case USVMPC_MODEL_PF_CA: nx = 14; nu = 2; return 0;
#ifdef USV_GEN_NX
    case USVMPC_MODEL_GENERATED: nx = USV_GEN_NX; nu = USV_GEN_NU; return 0;
#endif
    }
    return -1;
}

// Returns "" on success, else a description of what is wrong with the description.
inline std::string build_spec(const usvmpc_desc &d, DevSpec &S)
{
    int nx, nu;
    if (model_dims(d.model, nx, nu)) return "unknown model id";
    if (d.N < 2) return "N must be >= 2";
    if (!(d.Tf > 0.0)) return "Tf must be positive";
    if (d.batch < 1) return "batch must be >= 1";
    if (d.K < 0 || d.K > KMAX) return "K out of range (0..32)";
    if (d.model == USVMPC_MODEL_USV && d.K != 0) return "model usv_model has no obstacle rows (K must be 0)";
    if (d.nbu < 0 || d.nbu > nu || d.nbx < 0 || d.nbx > nx) return "nbu/nbx out of range";
    // The IPM forms products of two slacks / two multipliers of a row (paired reciprocals, qp_ipm.hpp): bounds standing in for
    // "none" must stay representable next to them.  acados' own ACADOS_INFTY is 1e10.
    {
        const double BIG = 1e100;
        for (int i = 0; i < d.nbu; i++) if (!(std::fabs(d.lbu[i]) <= BIG && std::fabs(d.ubu[i]) <= BIG)) return "lbu / ubu beyond 1e100 (or NaN)";
        for (int i = 0; i < d.nbx; i++) if (!(std::fabs(d.lbx[i]) <= BIG && std::fabs(d.ubx[i]) <= BIG)) return "lbx / ubx beyond 1e100 (or NaN)";
        for (int i = 0; i < d.K; i++) if (!(std::fabs(d.uh[i]) <= BIG)) return "uh beyond 1e100 (or NaN)";
    }
    const int nz = nx + nu, ny = nx + nu, ny_e = nx;
    std::memset(&S, 0, sizeof(S));
    S.dt = d.Tf / d.N;
    S.N = d.N; S.K = d.K; S.B = d.batch; S.Bp = (d.batch + 3) / 4 * 4;
    S.ny = ny; S.ny_e = ny_e;
    // V = [Vu Vx] (ny x nz); Hc = dt V'WV; Mc = dt V'W
    double V[LANES * LANES] = {0}, WV[LANES * LANES] = {0};
    for (int i = 0; i < ny; i++) {
        for (int j = 0; j < nu; j++) V[i * LANES + j] = d.Vu[i * nu + j];
        for (int j = 0; j < nx; j++) V[i * LANES + nu + j] = d.Vx[i * nx + j];
    }
    for (int i = 0; i < ny; i++)
        for (int j = 0; j < nz; j++) {
            double a = 0;
            for (int k = 0; k < ny; k++) a += d.W[i * ny + k] * V[k * LANES + j];
            WV[i * LANES + j] = a;
        }
    for (int i = 0; i < nz; i++) {
        for (int j = 0; j < nz; j++) {
            double a = 0;
            for (int k = 0; k < ny; k++) a += V[k * LANES + i] * WV[k * LANES + j];
            S.Hc[i * LANES + j] = S.dt * a;
        }
        for (int y = 0; y < ny; y++) {
            double a = 0;
            for (int k = 0; k < ny; k++) a += V[k * LANES + i] * d.W[k * ny + y];
            S.Mc[i * LANES + y] = S.dt * a;
        }
    }
    // terminal: He = Vx_e' W_e Vx_e, Me = Vx_e' W_e, placed at the x rows/cols
    for (int i = 0; i < nx; i++) {
        for (int j = 0; j < nx; j++) {
            double a = 0;
            for (int k = 0; k < ny_e; k++)
                for (int l = 0; l < ny_e; l++) a += d.Vx_e[k * nx + i] * d.W_e[k * ny_e + l] * d.Vx_e[l * nx + j];
            S.He[(nu + i) * LANES + nu + j] = a;
        }
        for (int y = 0; y < ny_e; y++) {
            double a = 0;
            for (int k = 0; k < ny_e; k++) a += d.Vx_e[k * nx + i] * d.W_e[k * ny_e + y];
            S.Me[(nu + i) * LANES + y] = a;
        }
    }
    for (int i = 0; i < d.nbu; i++) {
        const int j = d.idxbu[i];
        if (j < 0 || j >= nu) return "idxbu out of range";
        S.has_b[j] = 1; S.lb[j] = d.lbu[i]; S.ub[j] = d.ubu[i];
    }
    for (int i = 0; i < d.nbx; i++) {
        const int j = d.idxbx[i];
        if (j < 0 || j >= nx) return "idxbx out of range";
        S.has_b[nu + j] = 1; S.lb[nu + j] = d.lbx[i]; S.ub[nu + j] = d.ubx[i];
    }
    S.nbu = d.nbu; S.nbx = d.nbx;
    for (int i = 0; i < d.nbu; i++) S.box_pos[d.idxbu[i]] = i;
    for (int i = 0; i < d.nbx; i++) S.box_pos[nu + d.idxbx[i]] = d.nbu + i;
    for (int i = 0; i < d.nbx; i++) {
        if (!d.sbx[i]) continue;
        const int r = nu + d.idxbx[i];
        S.sbx_pos[r] = S.nsbx++;
        S.any_bsoft = 1;
        S.bsoft[r] = 1;
        S.b_zl[r] = S.dt * d.zl_bx[i]; S.b_zu[r] = S.dt * d.zu_bx[i];
        S.b_Zl[r] = S.dt * d.Zl_bx[i]; S.b_Zu[r] = S.dt * d.Zu_bx[i];
        S.b_lsl[r] = d.lsbx[i]; S.b_lsu[r] = d.usbx[i];
    }
    for (int i = 0; i < d.K; i++) {
        S.uh[i] = d.uh[i];
        S.lsl[i] = d.lsh[i]; S.lsu[i] = d.ush[i];
        S.zl[i] = S.dt * d.zl[i]; S.zu[i] = S.dt * d.zu[i];
        S.Zl[i] = S.dt * d.Zl[i]; S.Zu[i] = S.dt * d.Zu[i];
    }
    {   // Box rows ride in the idle lanes (>= k_last) of the last obstacle chunk; rows that do not fit there
        // are stored densely (four values in four consecutive lanes, from lane 0) in one plane.  The two lane
        // sets must be disjoint (one gather serves both), which bounds the dense rows by k_last / 4.
        const int kch = (d.K + LANES - 1) / LANES;
        const int k_last = d.K - (kch - 1) * LANES;
        int nb = 0;
        for (int r = 0; r < LANES; r++) nb += S.has_b[r];
        const int nslot = kch > 0 ? LANES - k_last : 0;
        const int ndense = nb > nslot ? nb - nslot : 0;
        // (soft state bounds carry six more values per row: they keep planes of their own)
        // (the dense rows live in lanes 0..7 of the per-stage aux plane, whose upper lanes hold other small items:
        // WsLayout in params.hpp)
        S.boxpack_ok = (kch > 0 && nb > 0 && ndense <= 2 && 4 * ndense <= k_last && !S.any_bsoft) ? 1 : 0;
        S.boxpack = S.boxpack_ok;
        if (S.boxpack_ok) {
            S.box_dense = ndense > 0;
            S.aux_dense4 = 4 * ndense;
            int slot = k_last, j = 0;
            for (int r = 0; r < LANES; r++) {
                if (!S.has_b[r]) continue;
                if (slot < LANES) {
                    S.box_slot[r] = slot; S.box_step[r] = 0; S.slot_var[slot] = r; S.slot_is[slot] = 1; slot++;
                } else {
                    S.box_slot[r] = 4 * j; S.box_step[r] = 1;
                    for (int e = 0; e < 4; e++) { S.slot_var[4 * j + e] = r; S.slot_is[4 * j + e] = 2; }
                    j++;
                }
            }
        }
    }
    S.hdiag = 1;
    for (int i = 0; i < LANES; i++)
        for (int j = 0; j < LANES; j++)
            if (i != j && (S.Hc[i * LANES + j] != 0.0 || S.He[i * LANES + j] != 0.0)) S.hdiag = 0;
    S.nc = d.N * d.nbu * 2 + (d.N - 1) * (d.nbx * 2 + d.K * (d.soft ? 4 : 2));
    for (int i = 0; i < d.nbx; i++) S.nc += d.sbx[i] ? (d.N - 1) * 2 : 0;
    S.iter_max = d.qp_iter_max;
    S.mu0 = d.mu0; S.thr0 = d.thr0;
    S.tol_stat = d.tol_stat; S.tol_eq = d.tol_eq; S.tol_ineq = d.tol_ineq; S.tol_comp = d.tol_comp;
    S.alpha_min = d.alpha_min;
    if (S.iter_max < 1) return "qp_iter_max must be >= 1";
    if (d.sim_num_steps < 0 || d.sim_num_steps > 64) return "sim_num_steps out of range (1..64)";
    S.sim_steps = d.sim_num_steps > 0 ? d.sim_num_steps : 1;
    if (d.hpipm_mode < USVMPC_HPIPM_BALANCE || d.hpipm_mode > USVMPC_HPIPM_R04) return "hpipm_mode must be one of USVMPC_HPIPM_*";
    if (d.cpc_factor < 0.0) return "cpc_factor must be positive (0: the default, 2)";
    S.cpc = d.cond_pred_corr != 0 ? 1 : 0;
    S.cpc_factor = d.cpc_factor > 0.0 ? d.cpc_factor : 2.0;
    S.nlp_tol[0] = d.nlp_tol_stat > 0.0 ? d.nlp_tol_stat : 1e-6;
    S.nlp_tol[1] = d.nlp_tol_eq > 0.0 ? d.nlp_tol_eq : 1e-6;
    S.nlp_tol[2] = d.nlp_tol_ineq > 0.0 ? d.nlp_tol_ineq : 1e-6;
    S.nlp_tol[3] = d.nlp_tol_comp > 0.0 ? d.nlp_tol_comp : 1e-6;
    if (d.nlp_max_iter < 0) return "nlp_max_iter must be >= 0";
    return "";
}

// entries per stage of the multiplier vectors usvmpc_get "lam" / "t" return (DevSpec: 2 (nrow + ns))
inline int lam_len(const DevSpec &S, bool soft) { return 2 * (S.nbu + S.nbx + S.K + S.nsbx + (soft ? S.K : 0)); }

// The fields of the descriptor a QP solver profile governs (include/usvmpc.h, USVMPC_HPIPM_*; DESIGN.md section 2: HPIPM's mode values as set by
// d_ocp_qp_ipm_arg_set_default and acados' overwrites in ocp_qp_hpipm_opts_initialize_default, as recalled).  The oracle has the same table
// (oracle/usv_oracle.c, usv_opts_profile) plus the two fields only it implements (itref_corr_max).
inline bool hpipm_profile(usvmpc_desc &d, int mode)
{
    if (mode < USVMPC_HPIPM_BALANCE || mode > USVMPC_HPIPM_R04) return false;
    const bool r04 = mode == USVMPC_HPIPM_R04;
    d.hpipm_mode = mode;
    d.qp_iter_max = 50;                                                                  // acados (HPIPM: 15 / 30 / 100)
    d.tol_stat = 1e-6; d.tol_eq = 1e-8; d.tol_ineq = 1e-8; d.tol_comp = 1e-8;            // acados (HPIPM: 1e-8 each)
    d.mu0 = r04 ? 10.0 : 1.0;                                                            // acados: 1 (HPIPM: 10 / 10 / 100)
    d.alpha_min = r04 ? 1e-12 : 1e-8;                                                    // acados: 1e-8 (HPIPM: 1e-12)
    d.cond_pred_corr = r04 ? 0 : 1;                                                      // HPIPM: 1 in SPEED, BALANCE and ROBUST
    d.cpc_factor = 2.0;
    return true;
}

inline void default_options(usvmpc_desc &d)
{
    hpipm_profile(d, USVMPC_HPIPM_BALANCE);
    d.thr0 = 0.1;
    d.sim_num_steps = 1;
    d.nlp_max_iter = 100;
    d.nlp_tol_stat = d.nlp_tol_eq = d.nlp_tol_ineq = d.nlp_tol_comp = 1e-6;
}

// planes of the solver workspace per stage for a (model, KCH, soft) combination: WsLayout::NPT (params.hpp);
// mat_planes = MatPack<M>::NPK of the model
inline int ws_planes(int nx, int nu, int kch, bool soft, int mat_planes, bool softbox = false)
{
    (void)nx;
    return 11 + kch * (soft ? 10 : 4) + nu + 2 + mat_planes + (softbox ? 6 : 0);
}

} // namespace usv
