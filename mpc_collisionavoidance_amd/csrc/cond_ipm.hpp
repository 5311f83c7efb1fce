// cond_ipm.hpp — partial condensing on the device (BASELINE.json configs[4]: N = 80 -> N2 = 10).
//
// What acados' qp_solver = PARTIAL_CONDENSING_HPIPM does when qp_solver_cond_N = N2 < N
// (/root/reference/catkin_ws/src/nmpc_ca/scripts/usv_pf_ca/acados_settings.py:172 selects the solver; the reference leaves
// cond_N at N, for which qp_ipm.hpp IS the solver): HPIPM d_part_cond_qp turns Mb consecutive stages (N / N2 of them, one more in
// the first N mod N2 blocks) into ONE stage
// whose state is x_k0 and whose input is the stack u_hat = (u_k0 .. u_k0+Mb-1); the intermediate states are eliminated
// through the dynamics, x_{k0+j} = Phi_j x_k0 + Gam_j u_hat + c_j.  The block's cost becomes a dense (nx + Mb nu)^2 Hessian,
// its dynamics a dense nx x (nx + Mb nu) matrix, and every inequality row of an intermediate stage a general row in
// (u_hat, x_k0).  The QP over the N2 dense stages is solved by the same Mehrotra predictor-corrector IPM as the uncondensed
// one (cold start, step rule, exit test: qp_ipm.hpp / oracle/usv_oracle.c), Riccati recursion over dense stages
// (Cholesky of the (Mb nu)^2 input block), and the solution is expanded back (d_part_cond_qp_expand_sol: states by the
// original dynamics, dynamics multipliers of the intermediate stages by the adjoint recursion).
//
// The general rows are kept in FACTORED form: a row of original stage k0+j is c' z_{k0+j} with c sparse (one entry for a
// bound, the two position entries for an obstacle row) and z_{k0+j} = T_j w + d_j, so C w, C'v and C' diag(g) C are formed
// as   expand (S_j w)  ->  sparse row  ->  S_j' (.)   with only the rows of S_j = [Gam_j Phi_j] that some row touches
// (bounded states and the position: 7 of 14 for usv_model_pf_ca).  Same numbers as the dense C of HPIPM up to rounding,
// 200 x 30 doubles per block less to stream.
//
// Mapping (different from qp_ipm.hpp, whose row-per-lane layout ends at 16 variables per stage): ONE instance per team of
// NT threads (a wave or a workgroup), the block's matrices in LDS, small dense kernels written as team-parallel loops
// over output elements; the per-instance condensed data (S_j rows, H0, factors, row multipliers) lives in a per-TEAM
// scratch area in HBM that is reused for every instance the team pulls from the queue.  Obstacle rows hard or soft (SOFT: slacks
// eliminated row by row as in qp_ipm.hpp's RowCalc); soft state bounds are not built.
//
// Parity: oracle/condense.py (numpy: part_cond + the oracle's IPM on the dense stages + expand) - tests/test_condensing.py
// (CPU: this file compiled by the host compiler for the lane emulator - one thread plays the team; -m gpu: the kernel).
#pragma once
#include "lanes.hpp"
#include "params.hpp"
#include "qp_ipm.hpp"
#include "cond_dims.hpp"

namespace usv {

// ---- the threads that work on one instance
#if !defined(__HIPCC__) // compiled by a host compiler (the lane emulator's build, tests/emu): one thread plays the whole team
struct CondTeam {
    static constexpr int NT = 1;
    static int tid() { return 0; }
    static void sync() {}
    static double rmax(double v, double *) { return v; }
    static double rsum(double v, double *) { return v; }
};
#define USV_CDEV inline
#else
template <int NT_>
struct CondTeam {
    static constexpr int NT = NT_;
    static_assert(NT % 64 == 0, "whole waves");
    __device__ static int tid() { return (int)threadIdx.x; }
    __device__ static void sync() { __syncthreads(); }
    __device__ static double rmax(double v, double *red)
    {
        for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        if constexpr (NT > 64) {
            sync();
            if ((tid() & 63) == 0) red[tid() >> 6] = v;
            sync();
            v = red[0];
            for (int i = 1; i < NT / 64; i++) v = fmax(v, red[i]);
        }
        return v;
    }
    __device__ static double rsum(double v, double *red)
    {
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if constexpr (NT > 64) {
            sync();
            if ((tid() & 63) == 0) red[tid() >> 6] = v;
            sync();
            v = red[0];
            for (int i = 1; i < NT / 64; i++) v += red[i];
        }
        return v;
    }
    // value of lane `src` (wave-uniform) of the calling wave
    __device__ static double lane_value(double v, int src)
    {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    }
};
#define USV_CDEV __device__ __forceinline__
#endif

template <class M, int KCH, bool SOFT, class TM>
struct CondIpm {
    static constexpr int NX = M::NX, NU = M::NU, NZ = NX + NU, NT = TM::NT;
    using MP = MatPack<M>;
    using WL = WsLayout<M, KCH, SOFT, false>;
    using Row = RowCalc<SOFT, SOFT>; // (soft obstacle rows next to hard box rows: the mixed form)
    static constexpr int NRA = SOFT ? 14 : 8; // values per row in the scratch area
    struct EntTab { unsigned char row[MP::NE > 0 ? MP::NE : 1], col[MP::NE > 0 ? MP::NE : 1]; };
    static constexpr EntTab make_tab()
    {
        EntTab t{};
        int s = 0;
        for (int j = 0; j < NX; j++)
            for (int c = 0; c < NZ; c++)
                if ((MP::row_mask(j) >> c) & 1u) { t.row[s] = (unsigned char)j; t.col[s] = (unsigned char)c; s++; }
        return t;
    }

    // Everything the sweeps read per element is kept out of global memory: DevSpec and CondDims live in HBM and the compiler must
    // assume the scratch stores alias them, so every `S.field` / `D.field` there would be a load of its own inside the loops.  The
    // scalars are copied into members once (registers), the short tables into LDS.
    struct DimsLocal {
        int Mb, N2, N1, R1, nuh, nzh, nxr, R, nrows, nbu, nbx, ipx, ipy;
        int o_SR, o_cr, o_BA, o_bt, o_H0, o_g0, o_row, o_Luu, o_P, o_Pb, o_w, o_pi, o_rg, o_rb, o_dwa, o_dw, o_dpi, o_p, o_lus, o_dg;
        long blk;
        const int *xr, *uvar, *xvar; // LDS
    };
    struct SpecLocal {
        const double *Hc, *He;        // global (condensing / expansion only)
        const double *HcD, *lb, *ub, *uh; // LDS: Hessian diagonal, box bounds per variable of [u;x], upper bounds of the obstacle rows
        const double *zl, *zu, *Zl, *Zu, *bsl, *bsu; // LDS: slack penalties (scaled by dt) and lower bounds of the slacks, per obstacle row
        const int *box_pos;           // LDS
        int N, K, B, Bp, npt, hdiag, p_static, iter_max, nbu, nbx, nc;
        double thr0, mu0, tol_stat, tol_eq, tol_ineq, tol_comp, alpha_min;
        int cpc;            // option "cond_pred_corr" (qp_ipm.hpp, QpIpm::solve: the same rule on the dense stages)
        double cpc_factor;
    };
    const DevPtrs &P;
    SpecLocal S;
    DimsLocal D;
    const int *tri; // LDS: element e of a lower triangle -> (row << 8) | column
    double *cw; // the team's scratch area in HBM
    int tid, N, Kn, Mb, N2, nuh, nzh, nxr, R, nrows;
    long g, b;
    // LDS
    double *Gm, *SRm, *Sm, *Sn, *Tm, *BAm, *PBm, *Pn, *BAk, *del, *delo, *dela, *delf, *yxr, *yxg, *wd, *wxy, *yur, *yug, *wu, *obuf, *red;
    double *vw, *vdwa, *vdw, *vr, *vgt, *vrq, *vg0, *vt, *vpi, *vpin, *vxn, *vpv, *vPb, *vrb, *vbt, *vdx, *vdxn, *vtmp, *vlus, *vq, *vgk, *vzb, *vdz, *vcr, *vdg;
    struct Norms { double rg, rb, rd, rm, musum; bool bad; };
    // second-order factor of the corrector targets of this pass / of the pending step: 0 where the corrected step was refused and the
    // centring-only one taken (the team holds one instance: a scalar)
    double so_cur = 1.0, so_prv = 1.0;

    USV_CDEV CondIpm(const DevPtrs &P_, const CondDims &Dg, double *scratch, double *lds) : P(P_), cw(scratch)
    {
        tid = TM::tid();
        const DevSpec &Sg = *P_.spec;
        S.Hc = Sg.Hc; S.He = Sg.He;
        S.N = Sg.N; S.K = Sg.K; S.B = Sg.B; S.Bp = Sg.Bp; S.npt = Sg.npt; S.hdiag = Sg.hdiag; S.p_static = Sg.p_static;
        S.iter_max = Sg.iter_max; S.nbu = Sg.nbu; S.nbx = Sg.nbx; S.nc = Sg.nc;
        S.thr0 = Sg.thr0; S.mu0 = Sg.mu0; S.tol_stat = Sg.tol_stat; S.tol_eq = Sg.tol_eq; S.tol_ineq = Sg.tol_ineq; S.tol_comp = Sg.tol_comp;
        S.alpha_min = Sg.alpha_min;
        S.cpc = Sg.cpc; S.cpc_factor = Sg.cpc_factor;
        D.Mb = Dg.Mb; D.N2 = Dg.N2; D.N1 = Dg.N1; D.R1 = Dg.R1; D.nuh = Dg.nuh; D.nzh = Dg.nzh; D.nxr = Dg.nxr; D.R = Dg.R; D.nrows = Dg.nrows; D.nbu = Dg.nbu; D.nbx = Dg.nbx;
        D.ipx = Dg.ipx; D.ipy = Dg.ipy;
        D.o_SR = (int)Dg.o_SR; D.o_cr = (int)Dg.o_cr; D.o_BA = (int)Dg.o_BA; D.o_bt = (int)Dg.o_bt; D.o_H0 = (int)Dg.o_H0; D.o_g0 = (int)Dg.o_g0;
        D.o_row = (int)Dg.o_row; D.o_Luu = (int)Dg.o_Luu; D.o_P = (int)Dg.o_P; D.o_Pb = (int)Dg.o_Pb; D.o_w = (int)Dg.o_w; D.o_pi = (int)Dg.o_pi;
        D.o_rg = (int)Dg.o_rg; D.o_rb = (int)Dg.o_rb; D.o_dwa = (int)Dg.o_dwa; D.o_dw = (int)Dg.o_dw; D.o_dpi = (int)Dg.o_dpi; D.o_p = (int)Dg.o_p;
        D.o_lus = (int)Dg.o_lus; D.o_dg = (int)Dg.o_dg; D.blk = Dg.blk;
        N = S.N; Kn = S.K; Mb = D.Mb; N2 = D.N2; nuh = D.nuh; nzh = D.nzh; nxr = D.nxr; R = D.R; nrows = D.nrows;
        double *q = lds;
        auto take = [&](long n) { double *at = q; q += n; return at; };
        {   // the short tables
            int *it = reinterpret_cast<int *>(take(2 * LANES + (nzh * (nzh + 1) / 2 + 1) / 2 + 1));
            int *xr_ = it, *uvar_ = it + LANES, *xvar_ = it + 2 * LANES, *bpos_ = it + 3 * LANES, *tri_ = it + 4 * LANES;
            double *sd = take(3 * LANES + 7 * KMAX);
            for (int e = tid; e < LANES; e += NT) {
                xr_[e] = Dg.xr[e]; uvar_[e] = Dg.uvar[e]; xvar_[e] = Dg.xvar[e]; bpos_[e] = Sg.box_pos[e];
                sd[e] = Sg.Hc[e * (LANES + 1)]; sd[LANES + e] = Sg.lb[e]; sd[2 * LANES + e] = Sg.ub[e];
            }
            for (int e = tid; e < KMAX; e += NT) {
                double *so = sd + 3 * LANES;
                so[e] = Sg.uh[e]; so[KMAX + e] = Sg.zl[e]; so[2 * KMAX + e] = Sg.zu[e]; so[3 * KMAX + e] = Sg.Zl[e]; so[4 * KMAX + e] = Sg.Zu[e];
                so[5 * KMAX + e] = Sg.lsl[e]; so[6 * KMAX + e] = Sg.lsu[e];
            }
            for (int a_ = tid; a_ < nzh; a_ += NT)
                for (int c = 0; c <= a_; c++) tri_[a_ * (a_ + 1) / 2 + c] = (a_ << 8) | c;
            D.xr = xr_; D.uvar = uvar_; D.xvar = xvar_; S.box_pos = bpos_; tri = tri_;
            S.HcD = sd; S.lb = sd + LANES; S.ub = sd + 2 * LANES; S.uh = sd + 3 * LANES;
            S.zl = S.uh + KMAX; S.zu = S.uh + 2 * KMAX; S.Zl = S.uh + 3 * KMAX; S.Zu = S.uh + 4 * KMAX; S.bsl = S.uh + 5 * KMAX; S.bsu = S.uh + 6 * KMAX;
        }
        Gm = take((long)(nzh + 1) * nzh);
        const long nsr = (long)Mb * nxr * nzh, ncn = 2L * NX * nzh + (long)NZ * nzh;
        SRm = take(nsr > ncn ? nsr : ncn);
        Sm = SRm; Sn = SRm + (long)NX * nzh; Tm = SRm + 2L * NX * nzh; // (condense phase only)
        BAm = take((long)NX * nzh); PBm = take((long)NX * nzh);
        Pn = take(NX * NX); BAk = take(NX * NZ);
        del = take(Mb * nxr); delo = take(Mb * nxr); dela = take(Mb * nxr); delf = take(Mb * nxr);
        yxr = take(Mb * nxr); yxg = take(Mb * nxr); wd = take(Mb * nxr); wxy = take(Mb);
        yur = take(nuh); yug = take(nuh); wu = take(nuh);
        obuf = take(4 * NT); red = take(64);
        vw = take(nzh); vdwa = take(nzh); vdw = take(nzh); vr = take(nzh); vgt = take(nzh); vrq = take(nzh); vg0 = take(nzh); vt = take(nzh);
        vpi = take(NX); vpin = take(NX); vxn = take(NX); vpv = take(NX); vPb = take(NX); vrb = take(NX); vbt = take(NX); vdx = take(NX);
        vdxn = take(NX); vtmp = take(NX); vq = take(NX); vlus = take(nuh); vgk = take(NZ); vzb = take(NZ); vdz = take(Mb * NZ); vcr = take(Mb * nxr);
        take(NX); vdg = take(nuh);
#if defined(__HIPCC__)
        if (q - lds > Dg.lds_doubles) __builtin_trap(); // (cond_dims.hpp sizes the launch's LDS: the two counts must agree)
#endif
    }

    USV_CDEV double *blk(int i) const { return cw + (long)i * D.blk; }
    USV_CDEV const double *plane(int k, int e) const { return P.ws + (((long)k * S.Bp + g) * S.npt + e) * LANES; }

    // ---- stage k of the lineariser's output: dense [B A] (NX x NZ) into BAk, b_k into vq, g_k = GQ + Hc zbar into vgk, zbar into vzb
    USV_CDEV void load_stage(int k)
    {
        static constexpr EntTab TAB = make_tab();
        TM::sync();
        for (int e = tid; e < NX * NZ; e += NT) {
            const int j = e / NZ, c = e - j * NZ;
            BAk[e] = (c == NU + j && ((M::DIAG_ONE >> j) & 1u)) ? 1.0 : 0.0;
        }
        for (int e = tid; e < NZ; e += NT) {
            double zb = 0.0;
            if (e < NU) zb = (k < N) ? P.u[((long)b * N + k) * NU + e] : 0.0;
            else zb = P.x[((long)b * (N + 1) + k) * NX + (e - NU)];
            vzb[e] = zb;
        }
        TM::sync();
        if (k < N) {
            for (int s = tid; s < MP::NE; s += NT) BAk[TAB.row[s] * NZ + TAB.col[s]] = plane(k, WL::P_MAT + s / 16)[s % 16];
            for (int e = tid; e < NX; e += NT) vq[e] = plane(k, WL::P_RB0)[NU + e];
        }
        const double *Hm = (k < N) ? S.Hc : S.He;
        for (int e = tid; e < NZ; e += NT) {
            double a = plane(k, WL::P_GQ)[e];
            if (S.hdiag && k < N) a = fma(S.HcD[e], vzb[e], a);
            else for (int c = 0; c < NZ; c++) a = fma(Hm[e * LANES + c], vzb[c], a);
            vgk[e] = a;
        }
        TM::sync();
    }

    // Block i covers the stages k0(i) .. k0(i) + mbi(i) - 1 (HPIPM's partition).  Everything is sized for the longest block (Mb stages);
    // a shorter block is PADDED: its surplus inputs get a unit Hessian diagonal, no rows and no effect on anything (they stay at 0),
    // its surplus sensitivity rows are zero - the other numbers of the block are exactly those of the short block.
    USV_CDEV int mbi(int i) const { return D.N1 + (i < D.R1 ? 1 : 0); }
    USV_CDEV int k0(int i) const { return i * D.N1 + (i < D.R1 ? i : D.R1); }
    // row q of stage j of block i: which kind, and whether the stage has it
    USV_CDEV bool row_active(int i, int j, int q) const { return j < mbi(i) && (q < D.nbu || (k0(i) + j) >= 1); }

    // ------------------------------------------------------------------ condensing (HPIPM d_part_cond_qp restated) + cold start
    // Returns whether x0 violates a hard obstacle row of stage 0 (qp_ipm.hpp init(): acados' QP is then infeasible).
    USV_CDEV bool condense()
    {
        double bad0 = 0.0;
        for (int i = 0; i < N2; i++) {
            double *W = blk(i);
            const int k0_ = k0(i), mb = mbi(i);
            TM::sync();
            for (int e = tid; e < NX * nzh; e += NT) { const int s = e / nzh, c = e - s * nzh; Sm[e] = (c == nuh + s) ? 1.0 : 0.0; }
            for (int e = tid; e < (nzh + 1) * nzh; e += NT) Gm[e] = 0.0;
            for (int e = tid; e < nzh; e += NT) vg0[e] = 0.0;
            for (int e = tid; e < NX; e += NT) vbt[e] = 0.0; // c_j
            TM::sync();
            for (int j = mb; j < Mb; j++) { // padding of a short block (see mbi)
                for (int e = tid; e < nxr * nzh; e += NT) W[D.o_SR + (long)j * nxr * nzh + e] = 0.0;
                for (int e = tid; e < nxr; e += NT) W[D.o_cr + j * nxr + e] = 0.0;
                for (int l = tid; l < NU; l += NT) Gm[(j * NU + l) * nzh + j * NU + l] = 1.0;
                for (int q = tid; q < R; q += NT) {
                    double *rw = W + D.o_row + (long)j * R + q;
                    rw[0] = 0.0; rw[nrows] = 0.0; rw[2 * nrows] = 1.0; rw[3 * nrows] = 1.0;
                    rw[4 * nrows] = -1.0; rw[5 * nrows] = 1.0; rw[6 * nrows] = 0.0; rw[7 * nrows] = 0.0;
                    if constexpr (SOFT) { rw[8 * nrows] = 0.0; rw[9 * nrows] = 0.0; rw[10 * nrows] = 0.0; rw[11 * nrows] = 0.0; rw[12 * nrows] = 1.0; rw[13 * nrows] = 1.0; }
                }
            }
            for (int j = 0; j < mb; j++) {
                const int k = k0_ + j;
                load_stage(k);
                // rows of S_j that some inequality row touches, and the offset c_j there
                for (int e = tid; e < nxr * nzh; e += NT) { const int r = e / nzh, c = e - r * nzh; W[D.o_SR + (long)j * nxr * nzh + e] = Sm[D.xr[r] * nzh + c]; }
                for (int e = tid; e < nxr; e += NT) W[D.o_cr + j * nxr + e] = vbt[D.xr[e]];
                // cost: H0 += T' Hc T, g0 += T' (g_k + Hc d),  T = [E_j; S_j], d = [0; c_j]
                for (int e = tid; e < NZ; e += NT) {
                    double a = vgk[e];
                    if (S.hdiag) a = (e >= NU) ? fma(S.HcD[e], vbt[e - NU], a) : a;
                    else for (int s = 0; s < NX; s++) a = fma(S.Hc[e * LANES + NU + s], vbt[s], a);
                    vt[e] = a; // gy
                }
                if (!S.hdiag) // HT = Hc T  (NZ x nzh)
                    for (int e = tid; e < NZ * nzh; e += NT) {
                        const int p = e / nzh, c = e - p * nzh;
                        double a = 0.0;
                        for (int l = 0; l < NU; l++) a += (c == j * NU + l) ? S.Hc[p * LANES + l] : 0.0;
                        for (int s = 0; s < NX; s++) a = fma(S.Hc[p * LANES + NU + s], Sm[s * nzh + c], a);
                        Tm[e] = a;
                    }
                TM::sync();
                for (int e = tid; e < nzh * nzh; e += NT) {
                    const int a_ = e / nzh, c = e - a_ * nzh;
                    double acc = Gm[e];
                    if (S.hdiag) {
                        for (int s = 0; s < NX; s++) acc = fma(Sm[s * nzh + a_] * S.HcD[NU + s], Sm[s * nzh + c], acc);
                        if (a_ == c && a_ >= j * NU && a_ < (j + 1) * NU) acc += S.HcD[a_ - j * NU];
                    } else {
                        if (a_ >= j * NU && a_ < (j + 1) * NU) acc += Tm[(a_ - j * NU) * nzh + c];
                        for (int s = 0; s < NX; s++) acc = fma(Sm[s * nzh + a_], Tm[(NU + s) * nzh + c], acc);
                    }
                    Gm[e] = acc;
                }
                for (int e = tid; e < nzh; e += NT) {
                    double acc = vg0[e];
                    if (e >= j * NU && e < (j + 1) * NU) acc += vt[e - j * NU];
                    for (int s = 0; s < NX; s++) acc = fma(Sm[s * nzh + e], vt[NU + s], acc);
                    vg0[e] = acc;
                }
                // inequality rows of the stage: constants in step coordinates, cold start (mu0 / thr0 as qp_ipm.hpp init())
                for (int q = tid; q < R; q += NT) {
                    const bool act = row_active(i, j, q);
                    double dl = -1.0, du = 1.0, cx = 0.0, cy = 0.0, v0 = 0.0;
                    if (q < D.nbu) {
                        const int l = D.uvar[q];
                        dl = S.lb[l] - vzb[l]; du = S.ub[l] - vzb[l];
                    } else if (q < D.nbu + D.nbx) {
                        const int s = D.xr[D.xvar[q - D.nbu]];
                        if (act) { dl = S.lb[NU + s] - vzb[NU + s]; du = S.ub[NU + s] - vzb[NU + s]; v0 = vbt[s]; }
                    } else {
                        const int o = q - D.nbu - D.nbx;
                        const int kk = S.p_static ? 0 : k;
                        const double *pk = P.p + ((long)b * (N + 1) + kk) * 2 * Kn;
                        const double lhv = P.lh[((long)b * N + (kk < N ? kk : N - 1)) * Kn + o];
                        double d, ux, uy;
                        obs_dist(vzb[NU + M::IPX] - pk[2 * o], vzb[NU + M::IPY] - pk[2 * o + 1], d, ux, uy);
                        if (act) {
                            dl = lhv - d; du = S.uh[o] - d; cx = ux; cy = uy;
                            v0 = ux * vbt[M::IPX] + uy * vbt[M::IPY];
                        } else if (!SOFT && k == 0) { // x0 inside a hard keep-out circle (or beyond uh)
                            const double e0x = P.x0[(long)b * NX + M::IPX] - vzb[NU + M::IPX], e0y = P.x0[(long)b * NX + M::IPY] - vzb[NU + M::IPY];
                            const double v = ux * e0x + uy * e0y;
                            if (lhv - d - v > S.tol_ineq || d + v - S.uh[o] > S.tol_ineq) bad0 = 1.0;
                        }
                    }
                    double *rw = W + D.o_row + (long)j * R + q;
                    double tl = 1.0, tu = 1.0, ll = 0.0, lu = 0.0;
                    if (act) {
                        tl = fmax(v0 - dl, S.thr0); tu = fmax(du - v0, S.thr0);
                        ll = S.mu0 / tl; lu = S.mu0 / tu;
                    }
                    rw[0] = ll; rw[nrows] = lu; rw[2 * nrows] = tl; rw[3 * nrows] = tu;
                    rw[4 * nrows] = dl; rw[5 * nrows] = du; rw[6 * nrows] = cx; rw[7 * nrows] = cy;
                    if constexpr (SOFT) { // slack pairs of a soft obstacle row (qp_ipm.hpp init()): s = 0, t_s = max(0 - ls, thr0), lam_s = mu0 / t_s
                        double tsl = 1.0, tsu = 1.0, lsl = 0.0, lsu = 0.0;
                        if (act && q >= D.nbu + D.nbx) {
                            const int o = q - D.nbu - D.nbx;
                            tsl = fmax(0.0 - S.bsl[o], S.thr0); tsu = fmax(0.0 - S.bsu[o], S.thr0);
                            lsl = S.mu0 / tsl; lsu = S.mu0 / tsu;
                        }
                        rw[8 * nrows] = 0.0; rw[9 * nrows] = 0.0; rw[10 * nrows] = lsl; rw[11 * nrows] = lsu; rw[12 * nrows] = tsl; rw[13 * nrows] = tsu;
                    }
                }
                // S_{j+1} = A_k S_j + B_k E_j,  c_{j+1} = A_k c_j + b_k
                for (int e = tid; e < NX * nzh; e += NT) {
                    const int s = e / nzh, c = e - s * nzh;
                    double a = 0.0;
                    for (int l = 0; l < NU; l++) a += (c == j * NU + l) ? BAk[s * NZ + l] : 0.0;
                    for (int m = 0; m < NX; m++) a = fma(BAk[s * NZ + NU + m], Sm[m * nzh + c], a);
                    Sn[e] = a;
                }
                for (int e = tid; e < NX; e += NT) {
                    double a = vq[e];
                    for (int m = 0; m < NX; m++) a = fma(BAk[e * NZ + NU + m], vbt[m], a);
                    vtmp[e] = a;
                }
                TM::sync();
                for (int e = tid; e < NX * nzh; e += NT) Sm[e] = Sn[e];
                for (int e = tid; e < NX; e += NT) vbt[e] = vtmp[e];
                TM::sync();
            }
            for (int e = tid; e < NX * nzh; e += NT) W[D.o_BA + e] = Sm[e];
            for (int e = tid; e < NX; e += NT) { W[D.o_bt + e] = vbt[e]; W[D.o_pi + e] = 0.0; W[D.o_dpi + e] = 0.0; }
            for (int e = tid; e < nzh * nzh; e += NT) W[D.o_H0 + e] = Gm[e];
            for (int e = tid; e < nzh; e += NT) { W[D.o_g0 + e] = vg0[e]; W[D.o_w + e] = 0.0; W[D.o_dwa + e] = 0.0; W[D.o_dw + e] = 0.0; }
        }
        {   // terminal stage: H = He (x block), g = GQ_N + He xbar_N, no rows
            double *W = blk(N2);
            load_stage(N);
            for (int e = tid; e < NX; e += NT) { W[D.o_g0 + e] = vgk[NU + e]; W[D.o_w + e] = 0.0; W[D.o_pi + e] = 0.0; W[D.o_dpi + e] = 0.0; W[D.o_dw + e] = 0.0; }
        }
        TM::sync();
        return TM::rmax(bad0, red) > 0.5;
    }

    // ---- small dense pieces on LDS operands (team-parallel over output elements; call between syncs)
    // out[j][r] = SRm[j][r][:] . v (+ cr)
    USV_CDEV void expand_rows(double *out, const double *v, const double *cr) const
    {
        for (int e = tid; e < Mb * nxr; e += NT) {
            double a = cr ? cr[e] : 0.0;
            const double *srow = SRm + (long)e * nzh;
#pragma unroll 6
            for (int c = 0; c < nzh; c++) a = fma(srow[c], v[c], a);
            out[e] = a;
        }
    }
    // out[c] += sum_{j,r} SRm[j][r][c] yx[j][r]  (+ yu at the u entries)
    USV_CDEV void rows_transposed(double *out, const double *yx, const double *yu) const
    {
        for (int c = tid; c < nzh; c += NT) {
            double a = out[c] + (c < nuh ? yu[c] : 0.0);
#pragma unroll 8
            for (int m = 0; m < Mb * nxr; m++) a = fma(SRm[(long)m * nzh + c], yx[m], a);
            out[c] = a;
        }
    }
    // the block's rows at the current iterate: one pass of NT rows at a time; obstacle rows of a stage are summed in row order
    // (deterministic) into the stage's position slots.  F(row index, j, q, Row &r, v, wa, wf) does the per-row work and returns
    // (yr, yg, Gh): coefficients of c in the residual / in the reduced gradient, and of c c' in the Hessian.
    template <class F>
    USV_CDEV void row_pass(int i, double *W, bool slots, F f)
    {
        if (slots) {
            for (int e = tid; e < Mb * nxr; e += NT) { yxr[e] = 0.0; yxg[e] = 0.0; wd[e] = 0.0; }
            for (int e = tid; e < Mb; e += NT) wxy[e] = 0.0;
            for (int e = tid; e < nuh; e += NT) { yur[e] = 0.0; yug[e] = 0.0; wu[e] = 0.0; }
        }
        TM::sync();
        for (int base = 0; base < nrows; base += NT) {
            const int e = base + tid;
            const bool has = e < nrows;
            const int j = has ? e / R : 0, q = has ? e - j * R : 0;
            double yr = 0.0, yg = 0.0, Gh = 0.0, cx = 0.0, cy = 0.0;
            bool obs = false;
            if (has) {
                double *rw = W + D.o_row + e;
                Row r;
                r.neutral(); // (sl = su = 0: the hard-row form still adds them)
                r.ll = rw[0]; r.lu = rw[nrows]; r.tl = rw[2 * nrows]; r.tu = rw[3 * nrows];
                r.dl = rw[4 * nrows]; r.du = rw[5 * nrows];
                r.act = row_active(i, j, q);
                double v, wa, wf;
                if (q < D.nbu) {
                    const int c = j * NU + D.uvar[q];
                    v = vw[c]; wa = vdwa[c]; wf = vdw[c];
                } else if (q < D.nbu + D.nbx) {
                    const int m = j * nxr + D.xvar[q - D.nbu];
                    v = del[m]; wa = dela[m]; wf = delf[m];
                } else {
                    obs = true;
                    if constexpr (SOFT) {
                        const int o = q - D.nbu - D.nbx;
                        r.soft = r.act;
                        r.sl = rw[8 * nrows]; r.su = rw[9 * nrows]; r.lsl = rw[10 * nrows]; r.lsu = rw[11 * nrows]; r.tsl = rw[12 * nrows]; r.tsu = rw[13 * nrows];
                        r.zl = S.zl[o]; r.zu = S.zu[o]; r.Zl = S.Zl[o]; r.Zu = S.Zu[o]; r.bsl = S.bsl[o]; r.bsu = S.bsu[o];
                    }
                    cx = rw[6 * nrows]; cy = rw[7 * nrows];
                    const int mx = j * nxr + D.ipx, my = j * nxr + D.ipy;
                    v = cx * del[mx] + cy * del[my]; wa = cx * dela[mx] + cy * dela[my]; wf = cx * delf[mx] + cy * delf[my];
                }
                f(e, j, q, r, rw, v, wa, wf, yr, yg, Gh);
                if (!r.act) { yr = 0.0; yg = 0.0; Gh = 0.0; }
                if (slots && !obs) {
                    if (q < D.nbu) { const int c = j * NU + D.uvar[q]; yur[c] = yr; yug[c] = yg; wu[c] = Gh; }
                    else { const int m = j * nxr + D.xvar[q - D.nbu]; yxr[m] = yr; yxg[m] = yg; wd[m] = Gh; }
                }
            }
            if (slots && Kn > 0) {
                obuf[tid] = obs ? cx * yr : 0.0; obuf[NT + tid] = obs ? cy * yr : 0.0;
                obuf[2 * NT + tid] = obs ? cx * yg : 0.0; obuf[3 * NT + tid] = obs ? cy * yg : 0.0;
                TM::sync();
                // stages this pass touches: j_lo .. j_hi; one thread per (stage, quantity)
                const int j_lo = base / R, j_hi = (base + NT - 1 < nrows ? base + NT - 1 : nrows - 1) / R;
                for (int t = tid; t < (j_hi - j_lo + 1) * 4; t += NT) {
                    const int jj = j_lo + t / 4, w = t & 3;
                    int e0 = jj * R + D.nbu + D.nbx, e1 = e0 + Kn;
                    if (e0 < base) e0 = base;
                    if (e1 > base + NT) e1 = base + NT;
                    double a = 0.0;
                    for (int ee = e0; ee < e1; ee++) a += obuf[w * NT + (ee - base)];
                    double *dst = (w == 0) ? &yxr[jj * nxr + D.ipx] : (w == 1) ? &yxr[jj * nxr + D.ipy] : (w == 2) ? &yxg[jj * nxr + D.ipx] : &yxg[jj * nxr + D.ipy];
                    *dst += a;
                }
                TM::sync();
                obuf[tid] = obs ? cx * cx * Gh : 0.0; obuf[NT + tid] = obs ? cy * cy * Gh : 0.0; obuf[2 * NT + tid] = obs ? cx * cy * Gh : 0.0;
                TM::sync();
                for (int t = tid; t < (j_hi - j_lo + 1) * 3; t += NT) {
                    const int jj = j_lo + t / 3, w = t % 3;
                    int e0 = jj * R + D.nbu + D.nbx, e1 = e0 + Kn;
                    if (e0 < base) e0 = base;
                    if (e1 > base + NT) e1 = base + NT;
                    double a = 0.0;
                    for (int ee = e0; ee < e1; ee++) a += obuf[w * NT + (ee - base)];
                    double *dst = (w == 0) ? &wd[jj * nxr + D.ipx] : (w == 1) ? &wd[jj * nxr + D.ipy] : &wxy[jj];
                    *dst += a;
                }
                TM::sync();
            }
        }
        TM::sync();
    }

    USV_CDEV void load_block(int i, double *W, bool hess)
    {
        TM::sync();
        for (int e = tid; e < Mb * nxr * nzh; e += NT) SRm[e] = W[D.o_SR + e];
        for (int e = tid; e < Mb * nxr; e += NT) vcr[e] = W[D.o_cr + e];
        for (int e = tid; e < NX * nzh; e += NT) BAm[e] = W[D.o_BA + e];
        for (int e = tid; e < nzh; e += NT) { vw[e] = W[D.o_w + e]; vdwa[e] = W[D.o_dwa + e]; vdw[e] = W[D.o_dw + e]; }
        if (hess) {
            for (int e = tid; e < nzh * nzh; e += NT) Gm[e] = W[D.o_H0 + e];
            for (int e = tid; e < nzh; e += NT) vg0[e] = W[D.o_g0 + e];
        }
        (void)i;
        TM::sync();
    }

    // lus = Luu^-1 rq_u,  pv = rq_x - Lxu lus   (L: nzh x nuh, row-major, in LDS at Gm with row stride nzh; vdg: 1 / diagonal)
    // On the device ONE wave does the substitution - lane r owns entry r, the pivot entry travels by readlane - instead of two
    // workgroup barriers per column: the sweeps are bound by their barrier count, not by arithmetic.
    USV_CDEV void solve_forward()
    {
        TM::sync();
        if constexpr (NT == 1) {
            for (int c = 0; c < nuh; c++) {
                const double y = vrq[c] * vdg[c];
                vrq[c] = y;
                for (int r = c + 1; r < nzh; r++) vrq[r] -= Gm[r * nzh + c] * y;
            }
        } else {
            if (tid < 64) {
                const int r = tid < nzh ? tid : nzh - 1;
                double y = vrq[r];
                const double dg = vdg[r < nuh ? r : 0];
                for (int c = 0; c < nuh; c++) {
                    const double yc = TM::lane_value(y, c) * TM::lane_value(dg, c);
                    const double l = Gm[r * nzh + c];
                    y = (r == c) ? yc : (r > c ? fma(-l, yc, y) : y);
                }
                if (tid < nzh) vrq[tid] = y;
            }
        }
        TM::sync();
    }
    // vt[0 .. nuh) <- Luu^-T vt   (back substitution, same arrangement)
    USV_CDEV void solve_backward()
    {
        TM::sync();
        if constexpr (NT == 1) {
            for (int c = nuh - 1; c >= 0; c--) {
                const double y = vt[c] * vdg[c];
                vt[c] = y;
                for (int r = 0; r < c; r++) vt[r] -= Gm[c * nzh + r] * y;
            }
        } else {
            if (tid < 64) {
                const int r = tid < nuh ? tid : nuh - 1;
                double y = vt[r];
                const double dg = vdg[r];
                for (int c = nuh - 1; c >= 0; c--) {
                    const double yc = TM::lane_value(y, c) * TM::lane_value(dg, c);
                    const double l = Gm[c * nzh + r];
                    y = (r == c) ? yc : (r < c ? fma(-l, yc, y) : y);
                }
                if (tid < nuh) vt[tid] = y;
            }
        }
        TM::sync();
    }

    // ------------------------------------------------------------------ backward sweep with factorisation
    // pend: the step of the previous iteration (dw, dpi, rows from dwa / sigmu_prev / dw) is applied first.
    USV_CDEV Norms backward_factor(bool pend, double a_prev, double sig_prev)
    {
        Norms nm{0.0, 0.0, 0.0, 0.0, 0.0, false};
        double badf = 0.0;
        {   // terminal stage
            double *W = blk(N2);
            TM::sync();
            for (int e = tid; e < NX; e += NT) {
                double w = W[D.o_w + e], pi = W[D.o_pi + e];
                if (pend) { w = fma(a_prev, W[D.o_dw + e], w); pi = fma(a_prev, W[D.o_dpi + e], pi); W[D.o_w + e] = w; W[D.o_pi + e] = pi; }
                vxn[e] = w; vpin[e] = pi;
            }
            TM::sync();
            for (int e = tid; e < NX; e += NT) {
                double r = W[D.o_g0 + e] - vpin[e];
                for (int c = 0; c < NX; c++) r = fma(S.He[(NU + e) * LANES + NU + c], vxn[c], r);
                W[D.o_rg + e] = r;
                vpv[e] = r;
                nm.rg = fmax(nm.rg, fabs(r));
                if (r != r) badf = fmax(badf, 1.0);
            }
            for (int e = tid; e < NX * NX; e += NT) Pn[e] = S.He[(NU + e / NX) * LANES + NU + e % NX];
        }
        for (int i = N2 - 1; i >= 0; i--) {
            double *W = blk(i);
            load_block(i, W, true);
            for (int e = tid; e < NX; e += NT) { vpi[e] = W[D.o_pi + e]; vbt[e] = W[D.o_bt + e]; }
            if (pend) { // rows need the old iterate and both steps
                expand_rows(delo, vw, vcr);
                expand_rows(dela, vdwa, nullptr);
                expand_rows(delf, vdw, nullptr);
                for (int e = tid; e < NX; e += NT) { vpi[e] = fma(a_prev, W[D.o_dpi + e], vpi[e]); W[D.o_pi + e] = vpi[e]; }
                TM::sync();
                // (the old values of the u rows are read from vw before it moves: keep a copy in vt)
                for (int e = tid; e < nzh; e += NT) vt[e] = vw[e];
                TM::sync();
                for (int e = tid; e < nzh; e += NT) { vw[e] = fma(a_prev, vdw[e], vw[e]); W[D.o_w + e] = vw[e]; }
            }
            TM::sync();
            expand_rows(del, vw, vcr);
            TM::sync();
            double rd = 0.0, rm = 0.0, mus = 0.0, bd = 0.0, rgs = 0.0;
            row_pass(i, W, true, [&](int e, int j, int q, Row &r, double *rw, double v, double wa, double wf, double &yr, double &yg, double &Gh) {
                (void)e;
                if (pend && r.act) {
                    double vo; // the row's value at the iterate the step was computed at
                    if (q < D.nbu) vo = vt[j * NU + D.uvar[q]];
                    else if (q < D.nbu + D.nbx) vo = delo[j * nxr + D.xvar[q - D.nbu]];
                    else vo = rw[6 * nrows] * delo[j * nxr + D.ipx] + rw[7 * nrows] * delo[j * nxr + D.ipy];
                    double g0_, g1_;
                    r.resid(vo); r.targets_pred(); r.reduce(g0_, g1_); r.expand(wa); r.template targets_corr<true>(sig_prev, so_prv); r.reduce(g0_, g1_);
                    r.expand(wf); r.apply(a_prev);
                    rw[0] = r.ll; rw[nrows] = r.lu; rw[2 * nrows] = r.tl; rw[3 * nrows] = r.tu;
                    if constexpr (SOFT) {
                        if (r.soft) { rw[8 * nrows] = r.sl; rw[9 * nrows] = r.su; rw[10 * nrows] = r.lsl; rw[11 * nrows] = r.lsu; rw[12 * nrows] = r.tsl; rw[13 * nrows] = r.tsu; }
                    }
                }
                r.resid(v); r.targets_pred(); r.reduce(Gh, yg);
                yr = -(r.ll - r.lu);
                if (r.act) {
                    rd = fmax(rd, fmax(fabs(r.rdl), fabs(r.rdu)));
                    rm = fmax(rm, fmax(r.ll * r.tl, r.lu * r.tu));
                    mus += r.ll * r.tl + r.lu * r.tu;
                    if (r.rdl != r.rdl || r.rdu != r.rdu || Gh != Gh) bd = fmax(bd, 2.0);
                    if constexpr (SOFT) {
                        if (r.soft) { // the slack pairs' residuals belong to the same four families (qp_ipm.hpp backward)
                            rgs = fmax(rgs, fmax(fabs(r.rsl), fabs(r.rsu)));
                            rd = fmax(rd, fmax(fabs(r.rdsl), fabs(r.rdsu)));
                            rm = fmax(rm, fmax(r.lsl * r.tsl, r.lsu * r.tsu));
                            mus += r.lsl * r.tsl + r.lsu * r.tsu;
                            if (r.rsl != r.rsl || r.rsu != r.rsu || r.rdsl != r.rdsl || r.rdsu != r.rdsu) bd = fmax(bd, 2.0);
                        }
                    }
                }
            });
            // r = g0 + H0 w + BA' pi_{i+1} - [0; pi_i];  rb = bt + BA w - x_{i+1}
            for (int c = tid; c < nzh; c += NT) {
                double a = vg0[c];
#pragma unroll 6
                for (int m = 0; m < nzh; m++) a = fma(Gm[c * nzh + m], vw[m], a);
                for (int s = 0; s < NX; s++) a = fma(BAm[s * nzh + c], vpin[s], a);
                if (i >= 1 && c >= nuh) a -= vpi[c - nuh];
                vr[c] = a;
            }
            for (int s = tid; s < NX; s += NT) {
                double a = vbt[s] - vxn[s];
                for (int c = 0; c < nzh; c++) a = fma(BAm[s * nzh + c], vw[c], a);
                vrb[s] = a;
            }
            TM::sync();
            rows_transposed(vr, yxr, yur);
            TM::sync();
            double rgl = 0.0, rbl = 0.0;
            for (int c = tid; c < nzh; c += NT) {
                W[D.o_rg + c] = vr[c];
                if (i >= 1 || c < nuh) rgl = fmax(rgl, fabs(vr[c]));
                if (vr[c] != vr[c]) bd = fmax(bd, 3.0);
                vgt[c] = vr[c];
            }
            for (int s = tid; s < NX; s += NT) { W[D.o_rb + s] = vrb[s]; rbl = fmax(rbl, fabs(vrb[s])); if (vrb[s] != vrb[s]) bd = fmax(bd, 4.0); }
            nm.rg = fmax(nm.rg, fmax(rgl, rgs)); nm.rb = fmax(nm.rb, rbl); nm.rd = fmax(nm.rd, rd); nm.rm = fmax(nm.rm, rm); nm.musum += mus;
            badf = fmax(badf, bd);
            TM::sync();
            rows_transposed(vgt, yxg, yug);
            // Ht = H0 + diag_u(wu) + sum_j SR_j' W_j SR_j   (lower triangle), then G = Ht + BA' P+ BA
            for (int e = tid; e < NX * nzh; e += NT) {
                const int s = e / nzh, c = e - s * nzh;
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < NX; m++) a = fma(Pn[s * NX + m], BAm[m * nzh + c], a);
                PBm[e] = a;
            }
            for (int s = tid; s < NX; s += NT) {
                double a = 0.0;
                for (int m = 0; m < NX; m++) a = fma(Pn[s * NX + m], vrb[m], a);
                vPb[s] = a;
                W[D.o_Pb + s] = a;
            }
            for (int e = tid; e < NX * NX; e += NT) W[D.o_P + e] = Pn[e];
            for (int e = tid; e < NX; e += NT) W[D.o_p + e] = vpv[e];
            TM::sync();
            const int ntri = nzh * (nzh + 1) / 2;
            for (int e = tid; e < ntri; e += NT) {
                const int a_ = tri[e] >> 8, c = tri[e] & 255;
                double acc = Gm[a_ * nzh + c];
                if (a_ == c && a_ < nuh) acc += wu[a_];
#pragma unroll 8
                for (int m = 0; m < Mb * nxr; m++) acc = fma(SRm[(long)m * nzh + a_] * wd[m], SRm[(long)m * nzh + c], acc);
                if (Kn > 0)
                    for (int j = 0; j < Mb; j++) {
                        const double *sx = SRm + (long)(j * nxr + D.ipx) * nzh, *sy = SRm + (long)(j * nxr + D.ipy) * nzh;
                        acc = fma(wxy[j], sx[a_] * sy[c] + sy[a_] * sx[c], acc);
                    }
#pragma unroll
                for (int s = 0; s < NX; s++) acc = fma(BAm[s * nzh + a_], PBm[s * nzh + c], acc);
                Gm[a_ * nzh + c] = acc;
            }
            // rq = gt + BA' (Pb + p_{i+1})
            for (int c = tid; c < nzh; c += NT) {
                double a = vgt[c];
                for (int s = 0; s < NX; s++) a = fma(BAm[s * nzh + c], vPb[s] + vpv[s], a);
                vrq[c] = a;
            }
            // eliminate the nuh input columns: [Luu; Lxu] stays in their place, the Schur complement P_i in the x block.  One barrier
            // per column: the trailing update uses the UNSCALED column (times 1 / pivot), which nothing writes during the step; the
            // columns are scaled to Cholesky form in one pass afterwards.
            for (int c = 0; c < nuh; c++) {
                TM::sync();
                const double piv = Gm[c * nzh + c];
                if (!(piv > 0.0)) badf = fmax(badf, 5.0);
                const double ipiv = 1.0 / piv;
                if (tid == 0) vdg[c] = 1.0 / sqrt(piv);
                const int nr = nzh - c - 1; // rows c+1 .. nzh-1, lower triangle of the trailing block
                for (int e = tid; e < nr * (nr + 1) / 2; e += NT) {
                    const int rr = c + 1 + (tri[e] >> 8), c2 = c + 1 + (tri[e] & 255);
                    Gm[rr * nzh + c2] = fma(-Gm[rr * nzh + c] * ipiv, Gm[c2 * nzh + c], Gm[rr * nzh + c2]);
                }
            }
            TM::sync();
            for (int e = tid; e < nzh * nuh; e += NT) {
                const int r = e / nuh, c = e - r * nuh;
                if (r >= c) Gm[r * nzh + c] *= vdg[c]; // (diagonal: piv / sqrt(piv))
            }
            solve_forward();
            for (int e = tid; e < nzh * nuh; e += NT) { const int r = e / nuh, c = e - r * nuh; W[D.o_Luu + e] = Gm[r * nzh + c]; }
            for (int c = tid; c < nuh; c += NT) { W[D.o_lus + c] = vrq[c]; W[D.o_dg + c] = vdg[c]; }
            // hand over to block i - 1
            for (int e = tid; e < NX * NX; e += NT) {
                const int s = e / NX, m = e - s * NX;
                Pn[e] = (m <= s) ? Gm[(nuh + s) * nzh + nuh + m] : Gm[(nuh + m) * nzh + nuh + s];
            }
            for (int s = tid; s < NX; s += NT) { vpv[s] = vrq[nuh + s]; vxn[s] = vw[nuh + s]; vpin[s] = vpi[s]; }
            TM::sync();
        }
        nm.rg = TM::rmax(nm.rg, red); nm.rb = TM::rmax(nm.rb, red); nm.rd = TM::rmax(nm.rd, red); nm.rm = TM::rmax(nm.rm, red);
        nm.musum = TM::rsum(nm.musum, red);
        // the initial-state residual dx0 - w_0[x] belongs to rb
        double e0m = 0.0;
        {
            const double *W0 = blk(0);
            for (int s = tid; s < NX; s += NT) {
                const double e0 = (P.x0[(long)b * NX + s] - P.x[((long)b * (N + 1)) * NX + s]) - W0[D.o_w + nuh + s];
                e0m = fmax(e0m, fabs(e0));
                if (e0 != e0) badf = fmax(badf, 6.0);
            }
        }
        nm.rb = fmax(nm.rb, TM::rmax(e0m, red));
        nm.bad = TM::rmax(badf, red) > 0.5; // (badf: 1 terminal residual, 2 row, 3 stationarity, 4 dynamics, 5 pivot, 6 initial state)
        return nm;
    }

    // ------------------------------------------------------------------ backward sweep on the stored factors (corrector rhs)
    USV_CDEV void backward_rhs(double sigmu)
    {
        {
            double *W = blk(N2);
            TM::sync();
            for (int e = tid; e < NX; e += NT) vpv[e] = W[D.o_rg + e];
        }
        for (int i = N2 - 1; i >= 0; i--) {
            double *W = blk(i);
            load_block(i, W, false);
            for (int e = tid; e < nzh * nuh; e += NT) { const int r = e / nuh, c = e - r * nuh; Gm[r * nzh + c] = W[D.o_Luu + e]; }
            for (int e = tid; e < NX; e += NT) { vPb[e] = W[D.o_Pb + e]; W[D.o_p + e] = vpv[e]; }
            for (int c = tid; c < nuh; c += NT) vdg[c] = W[D.o_dg + c];
            for (int c = tid; c < nzh; c += NT) vgt[c] = W[D.o_rg + c];
            expand_rows(del, vw, vcr);
            expand_rows(dela, vdwa, nullptr);
            TM::sync();
            row_pass(i, W, true, [&](int, int, int, Row &r, double *, double v, double wa, double, double &yr, double &yg, double &Gh) {
                double g0_, g1_;
                r.resid(v); r.targets_pred(); r.reduce(g0_, g1_); r.expand(wa); r.template targets_corr<true>(sigmu, so_cur); r.reduce(Gh, yg);
                yr = 0.0;
            });
            rows_transposed(vgt, yxg, yug);
            TM::sync();
            for (int c = tid; c < nzh; c += NT) {
                double a = vgt[c];
                for (int s = 0; s < NX; s++) a = fma(BAm[s * nzh + c], vPb[s] + vpv[s], a);
                vrq[c] = a;
            }
            solve_forward();
            for (int c = tid; c < nuh; c += NT) W[D.o_lus + c] = vrq[c];
            for (int s = tid; s < NX; s += NT) vpv[s] = vrq[nuh + s];
            TM::sync();
        }
    }

    // ------------------------------------------------------------------ forward sweep: step and step length
    // corr = false: affine step into dwa, returns alpha_aff and the sums S1, S2 of mu_aff;  true: final step into dw, dpi.
    USV_CDEV void forward(bool corr, double sigmu, double &alpha, double &S1, double &S2)
    {
        double qmax = 1.0, s1 = 0.0, s2 = 0.0;
        TM::sync();
        {
            const double *W0 = blk(0);
            for (int s = tid; s < NX; s += NT) vdx[s] = (P.x0[(long)b * NX + s] - P.x[((long)b * (N + 1)) * NX + s]) - W0[D.o_w + nuh + s];
        }
        for (int i = 0; i < N2; i++) {
            double *W = blk(i);
            load_block(i, W, false);
            for (int e = tid; e < nzh * nuh; e += NT) { const int r = e / nuh, c = e - r * nuh; Gm[r * nzh + c] = W[D.o_Luu + e]; }
            for (int c = tid; c < nuh; c += NT) { vlus[c] = W[D.o_lus + c]; vdg[c] = W[D.o_dg + c]; }
            for (int s = tid; s < NX; s += NT) vrb[s] = W[D.o_rb + s];
            TM::sync();
            // t = lus + Lxu' dx;  du = -Luu^-T t
            for (int c = tid; c < nuh; c += NT) {
                double a = vlus[c];
                for (int s = 0; s < NX; s++) a = fma(Gm[(nuh + s) * nzh + c], vdx[s], a);
                vt[c] = a;
            }
            solve_backward();
            double *dst = corr ? vdw : vdwa;
            for (int c = tid; c < nzh; c += NT) {
                const double v = (c < nuh) ? -vt[c] : vdx[c - nuh];
                dst[c] = v;
                W[(corr ? D.o_dw : D.o_dwa) + c] = v;
            }
            TM::sync();
            for (int s = tid; s < NX; s += NT) {
                double a = vrb[s];
                for (int c = 0; c < nzh; c++) a = fma(BAm[s * nzh + c], dst[c], a);
                vdxn[s] = a;
            }
            expand_rows(del, vw, vcr);
            expand_rows(dela, vdwa, nullptr);
            if (corr) expand_rows(delf, vdw, nullptr);
            TM::sync();
            if (corr) { // dpi_{i+1} = p_{i+1} + P_{i+1} dx_{i+1}
                double *Wn = blk(i + 1);
                for (int s = tid; s < NX; s += NT) {
                    double a = W[D.o_p + s];
                    for (int m = 0; m < NX; m++) a = fma(W[D.o_P + s * NX + m], vdxn[m], a);
                    Wn[D.o_dpi + s] = a;
                }
            }
            row_pass(i, W, false, [&](int, int, int, Row &r, double *, double v, double wa, double wf, double &, double &, double &) {
                double g0_, g1_;
                r.resid(v); r.targets_pred(); r.reduce(g0_, g1_); r.expand(wa);
                if (corr) { r.template targets_corr<true>(sigmu, so_cur); r.reduce(g0_, g1_); r.expand(wf); }
                qmax = r.blocking(qmax);
                if (r.act) { // (the sums of mu(alpha): of the affine step for sigma, of the corrected step for the conditional test)
                    s1 += r.ll * r.dtl + r.tl * r.dll + r.lu * r.dtu + r.tu * r.dlu;
                    s2 += r.dll * r.dtl + r.dlu * r.dtu;
                    if constexpr (SOFT) {
                        if (r.soft) {
                            s1 += r.lsl * r.dtsl + r.tsl * r.dlsl + r.lsu * r.dtsu + r.tsu * r.dlsu;
                            s2 += r.dlsl * r.dtsl + r.dlsu * r.dtsu;
                        }
                    }
                }
            });
            for (int s = tid; s < NX; s += NT) vdx[s] = vdxn[s];
            TM::sync();
        }
        {
            double *W = blk(N2);
            for (int s = tid; s < NX; s += NT) W[(corr ? D.o_dw : D.o_dwa) + s] = vdx[s];
        }
        qmax = TM::rmax(qmax, red);
        alpha = 1.0 / qmax;
        S1 = TM::rsum(s1, red); S2 = TM::rsum(s2, red);
    }

    // ------------------------------------------------------------------ expansion + RTI step + outputs
    USV_CDEV void finish(int status, int iters, const Norms &nm)
    {
        const bool ok = (status == 0 || status == 1);
        double tmin = 1e300;
        if (status != 4) {
            for (int i = 0; i < N2; i++) {
                double *W = blk(i);
                TM::sync();
                for (int e = tid; e < nzh; e += NT) vw[e] = W[D.o_w + e];
                for (int e = tid; e < NX; e += NT) { vdx[e] = W[D.o_w + nuh + e]; vpin[e] = blk(i + 1)[D.o_pi + e]; }
                for (int e = tid; e < nrows; e += NT) {
                    const int j = e / R, q = e - j * R;
                    if (!row_active(i, j, q)) continue;
                    const double *rw = W + D.o_row + e;
                    if (q >= D.nbu + D.nbx) tmin = fmin(tmin, rw[2 * nrows]);
                    const int kq = k0(i) + j;
                    if (P.lam_out) { // acados' row order [bu.., bx.., h..], lower | upper (qp_ipm.hpp export_rows)
                        const int nrow = S.nbu + S.nbx + Kn;
                        const int pos = q < D.nbu ? S.box_pos[D.uvar[q]] : (q < D.nbu + D.nbx ? S.box_pos[NU + D.xr[D.xvar[q - D.nbu]]] : S.nbu + S.nbx + (q - D.nbu - D.nbx));
                        double *L = P.lam_out + ((long)b * (N + 1) + kq) * P.nlam, *T = P.t_out + ((long)b * (N + 1) + kq) * P.nlam;
                        L[pos] = rw[0]; L[nrow + pos] = rw[nrows]; T[pos] = rw[2 * nrows]; T[nrow + pos] = rw[3 * nrows];
                        if constexpr (SOFT) {
                            if (q >= D.nbu + D.nbx) { // slack rows [sh..]: lower-slack bound | upper-slack bound
                                const int o = q - D.nbu - D.nbx, ns0 = 2 * nrow;
                                L[ns0 + o] = rw[10 * nrows]; L[ns0 + Kn + o] = rw[11 * nrows]; T[ns0 + o] = rw[12 * nrows]; T[ns0 + Kn + o] = rw[13 * nrows];
                            }
                        }
                    }
                    if constexpr (SOFT) {
                        if (q >= D.nbu + D.nbx && P.sl) {
                            const int o = q - D.nbu - D.nbx;
                            P.sl[((long)b * N + kq) * Kn + o] = rw[8 * nrows];
                            P.su[((long)b * N + kq) * Kn + o] = rw[9 * nrows];
                        }
                    }
                }
                TM::sync();
                // primal: the intermediate states by the original dynamics (d_part_cond_qp_expand_sol)
                const int mb = mbi(i);
                for (int j = 0; j < mb; j++) {
                    const int k = k0(i) + j;
                    load_stage(k);
                    if constexpr (SOFT) {
                        if (k == 0 && P.sl) { // soft rows of stage 0 (qp_ipm.hpp finish()): x_0 = x0 fixes their value, the slacks minimise their own penalty
                            const double e0x = P.x0[(long)b * NX + M::IPX] - vzb[NU + M::IPX], e0y = P.x0[(long)b * NX + M::IPY] - vzb[NU + M::IPY];
                            for (int o = tid; o < Kn; o += NT) {
                                const double *pk = P.p + ((long)b * (N + 1)) * 2 * Kn;
                                const double lhv = P.lh[((long)b * N) * Kn + o];
                                double d, ux, uy;
                                obs_dist(vzb[NU + M::IPX] - pk[2 * o], vzb[NU + M::IPY] - pk[2 * o + 1], d, ux, uy);
                                const double v0 = ux * e0x + uy * e0y;
                                double a = fmax(S.bsl[o], lhv - d - v0), qq = fmax(S.bsu[o], d + v0 - S.uh[o]);
                                if (S.Zl[o] > 0.0) a = fmax(a, -S.zl[o] / S.Zl[o]);
                                if (S.Zu[o] > 0.0) qq = fmax(qq, -S.zu[o] / S.Zu[o]);
                                P.sl[(long)b * N * Kn + o] = a;
                                P.su[(long)b * N * Kn + o] = qq;
                            }
                        }
                    }
                    for (int e = tid; e < NZ; e += NT) vdz[j * NZ + e] = (e < NU) ? vw[j * NU + e] : vdx[e - NU];
                    TM::sync();
                    for (int s = tid; s < NX; s += NT) {
                        double a = vq[s];
                        for (int c = 0; c < NZ; c++) a = fma(BAk[s * NZ + c], vdz[j * NZ + c], a);
                        vdxn[s] = a;
                    }
                    TM::sync();
                    for (int s = tid; s < NX; s += NT) vdx[s] = vdxn[s];
                    TM::sync();
                }
                // dynamics multipliers: pi_{k0+Mb} = pi of the next condensed stage; inside the block the adjoint recursion
                // pi_k = (Hc z_k + g_k - C_k'(ll - lu))_x + A_k' pi_{k+1}
                if (P.pi) {
                    for (int j = mb - 1; j >= 0; j--) {
                        const int k = k0(i) + j;
                        TM::sync();
                        for (int e = tid; e < NX; e += NT) P.pi[((long)b * N + k) * NX + e] = vpin[e]; // pi_{k+1}
                        if (j == 0) break;
                        load_stage(k);
                        for (int s = tid; s < NX; s += NT) {
                            double a = vgk[NU + s];
                            if (S.hdiag) a = fma(S.HcD[NU + s], vdz[j * NZ + NU + s], a);
                            else for (int c = 0; c < NZ; c++) a = fma(S.Hc[(NU + s) * LANES + c], vdz[j * NZ + c], a);
                            for (int m = 0; m < NX; m++) a = fma(BAk[m * NZ + NU + s], vpin[m], a);
                            // rows of this stage on state s
                            const double *rw = W + D.o_row + (long)j * R;
                            for (int q = D.nbu; q < D.nbu + D.nbx; q++)
                                if (D.xr[D.xvar[q - D.nbu]] == s) a -= rw[q] - rw[nrows + q];
                            if (Kn > 0 && (s == M::IPX || s == M::IPY))
                                for (int o = 0; o < Kn; o++) {
                                    const int q = D.nbu + D.nbx + o;
                                    a -= rw[(s == M::IPX ? 6 : 7) * nrows + q] * (rw[q] - rw[nrows + q]);
                                }
                            vtmp[s] = a;
                        }
                        TM::sync();
                        for (int s = tid; s < NX; s += NT) vpin[s] = vtmp[s];
                    }
                }
                // the RTI step (after the multiplier pass, which linearises at the OLD iterate)
                TM::sync();
                if (ok)
                    for (int e = tid; e < mb * NZ; e += NT) {
                        const int j = e / NZ, c = e - j * NZ, k = k0(i) + j;
                        if (c < NU) P.u[((long)b * N + k) * NU + c] += vdz[e];
                        else P.x[((long)b * (N + 1) + k) * NX + (c - NU)] += vdz[e];
                    }
            }
            if (ok) {
                const double *W = blk(N2);
                for (int e = tid; e < NX; e += NT) P.x[((long)b * (N + 1) + N) * NX + e] += W[D.o_w + e];
            }
        }
        tmin = -TM::rmax(-tmin, red);
        if (tid == 0) {
            if (P.obs_tmin) P.obs_tmin[b] = tmin;
            if (!ok && P.fail_count) atomic_one(P.fail_count);
            P.status[b] = ok ? 0 : 4;
            P.qp_iter[b] = iters;
            P.qp_status[b] = status;
            if (status != 4) { P.res[b * 4 + 0] = nm.rg; P.res[b * 4 + 1] = nm.rb; P.res[b * 4 + 2] = nm.rd; P.res[b * 4 + 3] = nm.rm; }
        }
        TM::sync();
    }
    USV_CDEV static void atomic_one(int *p) { lanes::count_one(p); }

    // ------------------------------------------------------------------ one instance
    USV_CDEV void solve(long group)
    {
        g = group;
        b = P.perm ? (long)P.perm[g] : g;
        const bool bad0 = condense();
        so_cur = 1.0; so_prv = 1.0;
        int status = bad0 ? 4 : 1, it = 0;
        Norms nm{0.0, 0.0, 0.0, 0.0, 0.0, false};
        bool pend = false;
        double a_prev = 0.0, sig_prev = 0.0;
        const double nc = (double)S.nc;
        while (!bad0) {
            nm = backward_factor(pend, a_prev, sig_prev);
            if (nm.bad || nm.rg != nm.rg || nm.rb != nm.rb) { status = 3; break; }
            if (nm.rg <= S.tol_stat && nm.rb <= S.tol_eq && nm.rd <= S.tol_ineq && nm.rm <= S.tol_comp) { status = 0; break; }
            if (it >= S.iter_max) { status = 1; break; }
            const double mu = nc > 0.0 ? nm.musum / nc : 0.0;
            double a_aff = 1.0, S1 = 0.0, S2 = 0.0, a = 1.0, d1, d2;
            forward(false, 0.0, a_aff, S1, S2);
            double sigmu = 0.0, mu_aff = 0.0;
            if (nc > 0.0) {
                mu_aff = (nm.musum + a_aff * S1 + a_aff * a_aff * S2) / nc;
                const double sg = mu_aff / mu;
                sigmu = sg * sg * sg * mu;
            }
            so_cur = 1.0;
            backward_rhs(sigmu);
            forward(true, sigmu, a, d1, d2);
            if (S.cpc && nc > 0.0) { // HPIPM's conditional predictor-corrector (QpIpm::solve): refused -> the centring-only step
                const double mu_pc = (nm.musum + a * d1 + a * a * d2) / nc;
                if (mu_pc > S.cpc_factor * mu_aff) {
                    so_cur = 0.0;
                    backward_rhs(sigmu);
                    forward(true, sigmu, a, d1, d2);
                }
            }
            if (a < S.alpha_min) { status = 2; break; }
            a_prev = a * ((1.0 - a) * 0.99 + a * 0.9999999);
            sig_prev = sigmu;
            so_prv = so_cur;
            pend = true;
            it++;
        }
        finish(status, it, nm);
    }
};

} // namespace usv
