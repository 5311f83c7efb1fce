// models.hpp — the three in-scope USV models as device functions.
//
// Each model provides f(x,u) together with the directional derivative Jx(x)·s + Ju(x,u)·su for ONE
// sensitivity column (s, su) (the lane's own column of [dx+/du | dx+/dx]; su is the unit vector of
// the lane's control or zero).  Models generated from a symbolic definition (codegen.py) have the
// same interface.  Formulas follow the reference's
// CasADi definitions (paths relative to /root/reference/catkin_ws/src/nmpc_ca/scripts/):
//   M0 usv_model              usv_acados/usv_model.py:61-122
//   M1 usv_model_guidance_ca1 usv_guidance_ca1/usv_model.py:61-140
//   M2 usv_model_pf_ca        usv_pf_ca/usv_model.py:61-168
// CasADi differentiates fabs() to sign(), if_else() branches to their own derivatives and
// sqrt(u²+v²) to a 0/0 at the origin; the same conventions are used here.
#pragma once
#include "lanes.hpp"
#include <cmath>

namespace usv {

USV_DEV double sgn(double a) { return (a > 0.0 ? 1.0 : 0.0) - (a < 0.0 ? 1.0 : 0.0); }

// 3-DOF surface-vessel block shared by M0 and M2 (usv_acados/usv_model.py:61-77,110-122).
// In: u,v,r,Tp,Ts and the matching entries of the column s; out: (udot,vdot,rdot) and J·s.
struct Dof3 {
    static constexpr double m = 30.0, Iz = 4.1, Bw = 0.41;
    static constexpr double Xud = -2.25, Yvd = -23.13, Yrd = -1.31, Nvd = -16.41, Nrd = -2.79;
    static constexpr double Yvv = -99.99, Yvr = -5.49, Nrv = -8.8, Nrr = -3.49;

    USV_DEV static void eval(double c, double u, double v, double r, double Tp, double Ts,
                             double su, double sv, double sr, double sTp, double sTs,
                             double *f, double *js)
    {
        const double CY = 0.5 * (-40.0 * 1000.0) *
                          (1.1 + 0.0045 * (1.01 / 0.09) - 0.1 * (0.27 / 0.09) + 0.016 * ((0.27 / 0.09) * (0.27 / 0.09)));
        const bool fast = u > 1.25;
        const double Xu = fast ? 64.55 : -25.0;
        const double Xuu = fast ? -70.92 : 0.0;
        const double au = fabs(u), av = fabs(v), ar = fabs(r);
        // sqrt(u^2 + v^2) and its reciprocal from one rsq estimate + Newton steps (lanes::frsqrt) instead of an IEEE square root
        // and an IEEE division: ~70 instructions less per evaluation, twenty evaluations per interval
        const double q2 = u * u + v * v;
        const double isp = lanes::frsqrt(q2); // inf / NaN at u = v = 0, as the CasADi expression's derivative
        const double sp = q2 > 0.0 ? q2 * isp : 0.0;
        const double Nr = -0.52 * sp;
        const double idu = 1.0 / (m - Xud), idv = 1.0 / (m - Yvd), idr = 1.0 / (Iz - Nrd);
        const double Tu = Tp + c * Ts;
        const double Tr = (Tp - c * Ts) * (Bw / 2.0);
        const double a = Yrd + Nvd;
        f[0] = (Tu - (-m + 2.0 * Yvd) * v - a * r * r - (-Xu * u - Xuu * au * u)) * idu;
        f[1] = (-(m - Xud) * u * r - (-CY * av - Yvv * av - Yvr * ar) * v) * idv;
        f[2] = (Tr - (-2.0 * Yvd * u * v - a * r * u + Xud * u * r) - (-Nr * r - Nrv * av * r - Nrr * ar * r)) * idr;
        // rows of the 3x5 Jacobian applied to (su,sv,sr,sTp,sTs)
        js[0] = ((Xu + 2.0 * Xuu * au) * su + (m - 2.0 * Yvd) * sv - 2.0 * a * r * sr + sTp + c * sTs) * idu;
        js[1] = (-(m - Xud) * r * su + (2.0 * (CY + Yvv) * av + Yvr * ar) * sv +
                 (-(m - Xud) * u + Yvr * sgn(r) * v) * sr) * idv;
        {
            const double dNu = -0.52 * u * isp, dNv = -0.52 * v * isp;
            js[2] = ((2.0 * Yvd * v + a * r - Xud * r + dNu * r) * su +
                     (2.0 * Yvd * u + dNv * r + Nrv * sgn(v) * r) * sv +
                     (a * u - Xud * u + Nr + Nrv * av + 2.0 * Nrr * ar) * sr +
                     (Bw / 2.0) * sTp - c * (Bw / 2.0) * sTs) * idr;
        }
    }
};

struct ModelM0 {
    static constexpr int ID = 0, NX = 5, NU = 2, IPX = 0, IPY = 0; // no obstacles (K = 0)
    // structural identities of the discrete map (see ModelM2): none for this model
    static constexpr unsigned OUT_UNIT = 0u, IN_UNIT = 0u;
    // discrete sensitivity pattern (MatPack, params.hpp), z = (U0, U1 | u, v, r, Tport, Tstbd): the 3-DOF rows see
    // everything; a thrust row is its own state plus dt * its rate
    static constexpr unsigned SENS[NX] = {0x7fu, 0x7fu, 0x7fu, 1u << 0, 1u << 1};
    static constexpr unsigned DIAG_ONE = (1u << 3) | (1u << 4);
    USV_DEV static void fjvp(const double *x, const double *U, const double *s, const double *su, double *f, double *js)
    {
        Dof3::eval(0.78, x[0], x[1], x[2], x[3], x[4], s[0], s[1], s[2], s[3], s[4], f, js);
        f[3] = U[0];
        f[4] = U[1];
        js[3] = su[0];
        js[4] = su[1];
    }
};

struct ModelM1 {
    static constexpr int ID = 1, NX = 8, NU = 1, IPX = 5, IPY = 6;
    // outputs u, v have f = 0; inputs ye, xned, yned appear in no right-hand side
    static constexpr unsigned OUT_UNIT = (1u << 0) | (1u << 1);
    static constexpr unsigned IN_UNIT = (1u << (NU + 2)) | (1u << (NU + 5)) | (1u << (NU + 6));
    // discrete sensitivity pattern, z = (U | u, v, ye, chie, psied, xned, yned, psi) = bits 0 | 1..8:
    //   chie' and psi' read (u, v, chie, psied), psied' = U, so both reach U through psied; ye' reads (u, v, chie);
    //   the positions read (u, v, psi).  Only chie feeds itself.
    static constexpr unsigned CH = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 4) | (1u << 5); // U, u, v, chie, psied
    static constexpr unsigned SENS[NX] = {0u, 0u, CH, CH, 1u << 0, CH | (1u << 8), CH | (1u << 8), CH};
    static constexpr unsigned DIAG_ONE = 0xffu & ~(1u << 3);
    // x = (u, v, ye, chie, psied, xned, yned, psi), T1 = 1
    USV_DEV static void fjvp(const double *x, const double *U, const double *s, const double *su, double *f, double *js)
    {
        const double u = x[0], v = x[1], chie = x[3], psied = x[4], psi = x[7];
        const double ue = u + 0.001;
        const double iden = 1.0 / (ue * ue + v * v);
        const double beta = atan2(v, ue);
        const double psie = chie - beta;
        double sp, cp, sq, cq;
        sincos(psie, &sp, &cp);
        sincos(psi, &sq, &cq);
        // directional derivative of psie along s
        const double dpsie = s[3] - (-v * iden * s[0] + ue * iden * s[1]);
        f[0] = 0.0;
        f[1] = 0.0;
        f[2] = u * sp + v * cp;
        f[3] = psied - psie;
        f[4] = U[0];
        f[5] = u * cq - v * sq;
        f[6] = u * sq + v * cq;
        f[7] = psied - psie;
        js[0] = 0.0;
        js[1] = 0.0;
        js[2] = sp * s[0] + cp * s[1] + (u * cp - v * sp) * dpsie;
        js[3] = s[4] - dpsie;
        js[4] = su[0];
        js[5] = cq * s[0] - sq * s[1] + (-u * sq - v * cq) * s[7];
        js[6] = sq * s[0] + cq * s[1] + (u * cq - v * sq) * s[7];
        js[7] = s[4] - dpsie;
    }
};

struct ModelM2 {
    static constexpr int ID = 2, NX = 14, NU = 2, IPX = 10, IPY = 11;
    // Structural identities of x+ = Phi(x,u), exact for any explicit RK scheme:
    //  OUT_UNIT bit j : f_j == 0, so x+_j = x_j and row j of [A B] is the unit vector e_j
    //                   (x1, y1, ak) -> plane j of [B A]' is never stored or loaded;
    //  IN_UNIT bit c  : variable c of [u;x] appears in no right-hand side, so column c of [B A] is
    //                   the unit vector (sinpsi, cospsi, ye, x1, y1, nedx, nedy) -> plane c of the
    //                   [B A] rows is never stored or loaded.
    // Pinned against the oracle's dense sensitivities in tests/test_oracle_models.py.
    static constexpr unsigned OUT_UNIT = (1u << 7) | (1u << 8) | (1u << 9);
    static constexpr unsigned IN_UNIT = (1u << (NU + 1)) | (1u << (NU + 2)) | (1u << (NU + 6)) | (1u << (NU + 7)) |
                                        (1u << (NU + 8)) | (1u << (NU + 10)) | (1u << (NU + 11));
    // discrete sensitivity pattern, z = (U0, U1 | psi, sinpsi, cospsi, u, v, r, ye, x1, y1, ak, nedx, nedy, Tport, Tstbd)
    // = bits 0, 1 | 2..15: the 3-DOF core (u, v, r) is driven by the thrusts and those by their rates; psi' = r;
    // sinpsi', cospsi', the positions and ye' read (psi, u, v[, r]), ye' also ak.  Only u, v, r feed themselves.
    static constexpr unsigned CORE = (1u << 0) | (1u << 1) | (1u << 5) | (1u << 6) | (1u << 7) | (1u << 14) | (1u << 15);
    static constexpr unsigned SENS[NX] = {CORE, CORE | (1u << 2), CORE | (1u << 2), CORE, CORE, CORE,
                                          CORE | (1u << 2) | (1u << 11), 0u, 0u, 0u, CORE | (1u << 2), CORE | (1u << 2),
                                          1u << 0, 1u << 1};
    static constexpr unsigned DIAG_ONE = 0x3fffu & ~((1u << 3) | (1u << 4) | (1u << 5));
    // x = (psi, sinpsi, cospsi, u, v, r, ye, x1, y1, ak, nedx, nedy, Tport, Tstbd), c = 1
    //
    // Transcendentals: the reference writes beta = atan2(v, u + .001), chi = psi + beta and uses sin / cos of psi, chi
    // and ak.  ak never changes inside a shooting interval (its right-hand side is zero), so its sine and cosine are
    // prepared once per interval (Pre) instead of at each of the 4 x steps stage points; and since
    // cos(beta) = ue / rho, sin(beta) = v / rho with rho = sqrt(ue^2 + v^2) in every quadrant, the angle-sum formulas
    // give sin / cos(chi) from sin / cos(psi) without the atan2 and without a second sincos (agreement with the
    // literal expressions: a few ulp, tests/test_ref_vectors.py).
    // psi does move inside an interval, but by no more than dt * |r| (a few hundredths of a radian: r is a bounded state), so its
    // sine and cosine at the 4 x steps stage points follow from those at the interval's start by the angle-sum formulas with a
    // short series in the increment (|d| <= 0.125: truncation below 1e-19; beyond that the library call) - one
    // sincos per interval instead of twenty.
    struct Pre { double sa, ca, psi0, sp0, cp0; };
    USV_DEV static Pre prepare(const double *x)
    {
        Pre p;
        sincos(x[9], &p.sa, &p.ca);
        p.psi0 = x[0];
        sincos(x[0], &p.sp0, &p.cp0);
        return p;
    }
    USV_DEV static void sincos_near(const Pre &pre, double psi, double &sp, double &cp)
    {
        const double d = psi - pre.psi0, d2 = d * d;
        if (fabs(d) > 0.125) { // (never on the reference's OCPs: |r| <= 1 rad/s, dt <= 0.05 s)
            sincos(psi, &sp, &cp);
            return;
        }
        // sin d = d (1 - d2/6 (1 - d2/20 (1 - d2/42 (1 - d2/72 (1 - d2/110))))),  cos d = 1 - d2/2 (1 - d2/12 (1 - d2/30 (1 - d2/56 (1 - d2/90))))
        const double sd = d * fma(-d2 * (1.0 / 6.0), fma(-d2 * (1.0 / 20.0), fma(-d2 * (1.0 / 42.0), fma(-d2 * (1.0 / 72.0), fma(-d2, 1.0 / 110.0, 1.0), 1.0), 1.0), 1.0), 1.0);
        const double cd = fma(-d2 * 0.5, fma(-d2 * (1.0 / 12.0), fma(-d2 * (1.0 / 30.0), fma(-d2 * (1.0 / 56.0), fma(-d2, 1.0 / 90.0, 1.0), 1.0), 1.0), 1.0), 1.0);
        sp = fma(pre.sp0, cd, pre.cp0 * sd);
        cp = fma(pre.cp0, cd, -pre.sp0 * sd);
    }
    USV_DEV static void fjvp(const double *x, const double *U, const double *s, const double *su, double *f, double *js)
    {
        fjvp_pre(prepare(x), x, U, s, su, f, js);
    }
    USV_DEV static void fjvp_pre(const Pre &pre, const double *x, const double *U, const double *s, const double *su, double *f, double *js)
    {
        const double psi = x[0], u = x[3], v = x[4], r = x[5];
        const double ue = u + .001;
        const double r2 = ue * ue + v * v;
        const double irho = lanes::frsqrt(r2);
        const double iden = irho * irho;
        const double cb = ue * irho, sb = v * irho;
        double sp, cp;
        sincos_near(pre, psi, sp, cp);
        const double sc = sp * cb + cp * sb, cc = cp * cb - sp * sb;
        const double sa = pre.sa, ca = pre.ca;
        const double dchi = s[0] + (-v * iden) * s[3] + (ue * iden) * s[4];
        double f3[3], j3[3];
        Dof3::eval(1.0, u, v, r, x[12], x[13], s[3], s[4], s[5], s[12], s[13], f3, j3);
        const double vx = u * cp - v * sp; // NED velocity components
        const double vy = u * sp + v * cp;
        f[0] = r;
        f[1] = cc * r;
        f[2] = -sc * r;
        f[3] = f3[0];
        f[4] = f3[1];
        f[5] = f3[2];
        f[6] = -vx * sa + vy * ca;
        f[7] = 0.0;
        f[8] = 0.0;
        f[9] = 0.0;
        f[10] = vx;
        f[11] = vy;
        f[12] = U[0];
        f[13] = U[1] / 1.0;
        const double dvx = cp * s[3] - sp * s[4] - vy * s[0];
        const double dvy = sp * s[3] + cp * s[4] + vx * s[0];
        js[0] = s[5];
        js[1] = -sc * r * dchi + cc * s[5];
        js[2] = -cc * r * dchi - sc * s[5];
        js[3] = j3[0];
        js[4] = j3[1];
        js[5] = j3[2];
        js[6] = -dvx * sa + dvy * ca + (-vx * ca - vy * sa) * s[9];
        js[7] = 0.0;
        js[8] = 0.0;
        js[9] = 0.0;
        js[10] = dvx;
        js[11] = dvy;
        js[12] = su[0];
        js[13] = su[1] / 1.0;
    }
};

} // namespace usv
