/*
 * usv_oracle.c — CPU restatement of the SQP-RTI hot path.  TEST INFRASTRUCTURE ONLY; see
 * usv_oracle.h for scope, the "parity unpinned" statement and the adopted conventions.
 * Every function cites the reference definition it follows (paths relative to
 * /root/reference/catkin_ws/src/nmpc_ca/scripts/).
 */
#include "usv_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NXM USV_NX_MAX
#define NUM USV_NU_MAX
#define NZM USV_NZ_MAX
#define KM USV_K_MAX
#define NYM USV_NY_MAX

/* ------------------------------------------------------------------------------------------
 * 1. Models
 * ---------------------------------------------------------------------------------------- */

/* 3-DOF surface-vessel coefficients: usv_acados/usv_model.py:61-77 (= usv_pf_ca/usv_model.py:61-77) */
static const double M_ = 30.0, IZ_ = 4.1, BW_ = 0.41;
static const double XUD = -2.25, YVD = -23.13, YRD = -1.31, NVD = -16.41, NRD = -2.79;
static const double YVV = -99.99, YVR = -5.49, NRV = -8.8, NRR = -3.49;

static double sgn(double a) { return (a > 0.0) - (a < 0.0); }

static double yv_coeff(void)
{ /* Yv = CY*|v| : usv_acados/usv_model.py:112 */
    return 0.5 * (-40.0 * 1000.0) *
           (1.1 + 0.0045 * (1.01 / 0.09) - 0.1 * (0.27 / 0.09) + 0.016 * ((0.27 / 0.09) * (0.27 / 0.09)));
}

/* (udot, vdot, rdot) of the 3-DOF block and its 3x5 Jacobian wrt (u,v,r,Tport,Tstbd).
 * usv_acados/usv_model.py:110-122, usv_pf_ca/usv_model.py:137-151. J may be NULL. */
static void dof3(double c, double u, double v, double r, double Tp, double Ts, double *f, double *J)
{
    const double Xu = (u > 1.25) ? 64.55 : -25.0;
    const double Xuu = (u > 1.25) ? -70.92 : 0.0;
    const double CY = yv_coeff();
    const double Yv = CY * fabs(v);
    const double s = sqrt(u * u + v * v);
    const double Nr = -0.52 * s;
    const double Tu = Tp + c * Ts;
    const double Tr = (Tp - c * Ts) * BW_ / 2.0;
    const double du = M_ - XUD, dv = M_ - YVD, dr = IZ_ - NRD;
    f[0] = (Tu - (-M_ + 2.0 * YVD) * v - (YRD + NVD) * r * r - (-Xu * u - Xuu * fabs(u) * u)) / du;
    f[1] = (-(M_ - XUD) * u * r - (-Yv - YVV * fabs(v) - YVR * fabs(r)) * v) / dv;
    f[2] = (Tr - (-2.0 * YVD * u * v - (YRD + NVD) * r * u + XUD * u * r) -
            (-Nr * r - NRV * fabs(v) * r - NRR * fabs(r) * r)) / dr;
    if (!J) return;
    /* columns: 0 u, 1 v, 2 r, 3 Tport, 4 Tstbd */
    J[0 * 5 + 0] = (Xu + 2.0 * Xuu * fabs(u)) / du;
    J[0 * 5 + 1] = -(-M_ + 2.0 * YVD) / du;
    J[0 * 5 + 2] = -2.0 * (YRD + NVD) * r / du;
    J[0 * 5 + 3] = 1.0 / du;
    J[0 * 5 + 4] = c / du;
    J[1 * 5 + 0] = -(M_ - XUD) * r / dv;
    J[1 * 5 + 1] = (2.0 * (CY + YVV) * fabs(v) + YVR * fabs(r)) / dv;
    J[1 * 5 + 2] = (-(M_ - XUD) * u + YVR * sgn(r) * v) / dv;
    J[1 * 5 + 3] = 0.0;
    J[1 * 5 + 4] = 0.0;
    {
        const double dNr_du = -0.52 * u / s, dNr_dv = -0.52 * v / s; /* NaN at u=v=0, as CasADi */
        J[2 * 5 + 0] = (2.0 * YVD * v + (YRD + NVD) * r - XUD * r + dNr_du * r) / dr;
        J[2 * 5 + 1] = (2.0 * YVD * u + dNr_dv * r + NRV * sgn(v) * r) / dr;
        J[2 * 5 + 2] = ((YRD + NVD) * u - XUD * u + Nr + NRV * fabs(v) + 2.0 * NRR * fabs(r)) / dr;
        J[2 * 5 + 3] = (BW_ / 2.0) / dr;
        J[2 * 5 + 4] = (-c * BW_ / 2.0) / dr;
    }
}

static usv_fjvp_fn g_gen_fjvp = 0;
static int g_gen_nx = 0, g_gen_nu = 0, g_gen_ipx = -1, g_gen_ipy = -1;

void usv_oracle_register_generated(usv_fjvp_fn fn, int nx, int nu, int ipx, int ipy)
{
    g_gen_fjvp = fn; g_gen_nx = nx; g_gen_nu = nu; g_gen_ipx = ipx; g_gen_ipy = ipy;
}

int usv_model_dims(int model, int *nx, int *nu)
{
    switch (model) {
    case USV_MGEN: if (!g_gen_fjvp) return -1; *nx = g_gen_nx; *nu = g_gen_nu; return 0;
    case USV_M0: *nx = 5; *nu = 2; return 0;   /* usv_acados/usv_model.py:81-91 */
    case USV_M1: *nx = 8; *nu = 1; return 0;   /* usv_guidance_ca1/usv_model.py:65-77 */
    case USV_M2: *nx = 14; *nu = 2; return 0;  /* usv_pf_ca/usv_model.py:81-100 */
    }
    return -1;
}

void usv_model_pos_idx(int model, int *ipx, int *ipy)
{
    if (model == USV_M1) { *ipx = 5; *ipy = 6; }        /* xned, yned */
    else if (model == USV_M2) { *ipx = 10; *ipy = 11; } /* nedx, nedy */
    else if (model == USV_MGEN) { *ipx = g_gen_ipx; *ipy = g_gen_ipy; }
    else { *ipx = -1; *ipy = -1; }
}

void usv_model_f(int model, const double *x, const double *U, double *f)
{
    if (model == USV_MGEN) {
        double s[NXM] = {0}, su[NUM] = {0}, js[NXM];
        g_gen_fjvp(x, U, s, su, f, js);
        return;
    }
    if (model == USV_M0) {
        /* usv_acados/usv_model.py:116-122, c = 0.78 (:77) */
        dof3(0.78, x[0], x[1], x[2], x[3], x[4], f, NULL);
        f[3] = U[0];
        f[4] = U[1];
    } else if (model == USV_M1) {
        /* usv_guidance_ca1/usv_model.py:117-128, T1 = 1 (:61) */
        const double u = x[0], v = x[1], chie = x[3], psied = x[4], psi = x[7];
        const double beta = atan2(v, u + 0.001);
        const double psie = chie - beta;
        f[0] = 0.0;
        f[1] = 0.0;
        f[2] = u * sin(psie) + v * cos(psie);
        f[3] = (psied - psie) / 1.0;
        f[4] = U[0];
        f[5] = u * cos(psi) - v * sin(psi);
        f[6] = u * sin(psi) + v * cos(psi);
        f[7] = (psied - psie) / 1.0;
    } else {
        /* usv_pf_ca/usv_model.py:137-160, c = 1.0 (:77) */
        const double c = 1.0;
        const double psi = x[0], u = x[3], v = x[4], r = x[5], ak = x[9];
        const double beta = atan2(v, u + .001);
        const double chi = psi + beta;
        double f3[3];
        dof3(c, u, v, r, x[12], x[13], f3, NULL);
        f[0] = r;
        f[1] = cos(chi) * r;
        f[2] = -sin(chi) * r;
        f[3] = f3[0];
        f[4] = f3[1];
        f[5] = f3[2];
        f[6] = -(u * cos(psi) - v * sin(psi)) * sin(ak) + (u * sin(psi) + v * cos(psi)) * cos(ak);
        f[7] = 0.0;
        f[8] = 0.0;
        f[9] = 0.0;
        f[10] = u * cos(psi) - v * sin(psi);
        f[11] = u * sin(psi) + v * cos(psi);
        f[12] = U[0];
        f[13] = U[1] / c;
    }
}

void usv_model_jac(int model, const double *x, const double *U, double *Jx, double *Ju)
{
    int nx, nu, i;
    (void)U;
    usv_model_dims(model, &nx, &nu);
    for (i = 0; i < nx * nx; i++) Jx[i] = 0.0;
    for (i = 0; i < nx * nu; i++) Ju[i] = 0.0;
    if (model == USV_MGEN) { /* dense Jacobians column by column from the generated tangent code */
        int c, r;
        for (c = 0; c < nx + nu; c++) {
            double s[NXM] = {0}, su[NUM] = {0}, f[NXM], js[NXM];
            if (c < nu) su[c] = 1.0; else s[c - nu] = 1.0;
            g_gen_fjvp(x, U, s, su, f, js);
            for (r = 0; r < nx; r++) {
                if (c < nu) Ju[r * nu + c] = js[r]; else Jx[r * nx + (c - nu)] = js[r];
            }
        }
        return;
    }
    if (model == USV_M0) {
        double f3[3], J3[15];
        int a, b;
        dof3(0.78, x[0], x[1], x[2], x[3], x[4], f3, J3);
        for (a = 0; a < 3; a++)
            for (b = 0; b < 5; b++) Jx[a * 5 + b] = J3[a * 5 + b];
        Ju[3 * 2 + 0] = 1.0;
        Ju[4 * 2 + 1] = 1.0;
    } else if (model == USV_M1) {
        const double u = x[0], v = x[1], chie = x[3], psi = x[7];
        const double ue = u + 0.001, den = ue * ue + v * v;
        const double beta = atan2(v, ue);
        const double bu = -v / den, bv = ue / den; /* d beta / du, dv */
        const double psie = chie - beta;
        const double sp = sin(psie), cp = cos(psie);
        /* d psie/du = -bu, d psie/dv = -bv, d psie/dchie = 1 */
        const double g = u * cp - v * sp; /* d f2 / d psie */
        Jx[2 * 8 + 0] = sp + g * (-bu);
        Jx[2 * 8 + 1] = cp + g * (-bv);
        Jx[2 * 8 + 3] = g;
        Jx[3 * 8 + 0] = bu;
        Jx[3 * 8 + 1] = bv;
        Jx[3 * 8 + 3] = -1.0;
        Jx[3 * 8 + 4] = 1.0;
        Ju[4 * 1 + 0] = 1.0;
        Jx[5 * 8 + 0] = cos(psi);
        Jx[5 * 8 + 1] = -sin(psi);
        Jx[5 * 8 + 7] = -u * sin(psi) - v * cos(psi);
        Jx[6 * 8 + 0] = sin(psi);
        Jx[6 * 8 + 1] = cos(psi);
        Jx[6 * 8 + 7] = u * cos(psi) - v * sin(psi);
        Jx[7 * 8 + 0] = bu;
        Jx[7 * 8 + 1] = bv;
        Jx[7 * 8 + 3] = -1.0;
        Jx[7 * 8 + 4] = 1.0;
    } else {
        const double c = 1.0;
        const double psi = x[0], u = x[3], v = x[4], r = x[5], ak = x[9];
        const double ue = u + .001, den = ue * ue + v * v;
        const double beta = atan2(v, ue);
        const double bu = -v / den, bv = ue / den;
        const double chi = psi + beta, sc = sin(chi), cc = cos(chi);
        const double sp = sin(psi), cp = cos(psi), sa = sin(ak), ca = cos(ak);
        double f3[3], J3[15];
        int a;
        const int col3[5] = {3, 4, 5, 12, 13};
        dof3(c, u, v, r, x[12], x[13], f3, J3);
        Jx[0 * 14 + 5] = 1.0;
        Jx[1 * 14 + 0] = -sc * r;
        Jx[1 * 14 + 3] = -sc * r * bu;
        Jx[1 * 14 + 4] = -sc * r * bv;
        Jx[1 * 14 + 5] = cc;
        Jx[2 * 14 + 0] = -cc * r;
        Jx[2 * 14 + 3] = -cc * r * bu;
        Jx[2 * 14 + 4] = -cc * r * bv;
        Jx[2 * 14 + 5] = -sc;
        for (a = 0; a < 3; a++) {
            int b;
            for (b = 0; b < 5; b++) Jx[(3 + a) * 14 + col3[b]] = J3[a * 5 + b];
        }
        Jx[6 * 14 + 0] = -(-u * sp - v * cp) * sa + (u * cp - v * sp) * ca;
        Jx[6 * 14 + 3] = -cp * sa + sp * ca;
        Jx[6 * 14 + 4] = sp * sa + cp * ca;
        Jx[6 * 14 + 9] = -(u * cp - v * sp) * ca - (u * sp + v * cp) * sa;
        Jx[10 * 14 + 0] = -u * sp - v * cp;
        Jx[10 * 14 + 3] = cp;
        Jx[10 * 14 + 4] = -sp;
        Jx[11 * 14 + 0] = u * cp - v * sp;
        Jx[11 * 14 + 3] = sp;
        Jx[11 * 14 + 4] = cp;
        Ju[12 * 2 + 0] = 1.0;
        Ju[13 * 2 + 1] = 1.0 / c;
    }
}

void usv_model_h(int model, int K, const double *x, const double *p, double *h, double *Cxy)
{
    /* distance_i = sqrt((px-ox_i)^2 + (py-oy_i)^2): usv_guidance_ca1/usv_model.py:133-140,
     * usv_pf_ca/usv_model.py:165-168; generalised to K obstacles. */
    int ipx, ipy, i;
    usv_model_pos_idx(model, &ipx, &ipy);
    for (i = 0; i < K; i++) {
        const double dx = x[ipx] - p[2 * i], dy = x[ipy] - p[2 * i + 1];
        const double d = sqrt(dx * dx + dy * dy);
        h[i] = d;
        if (Cxy) {
            Cxy[2 * i] = dx / d;
            Cxy[2 * i + 1] = dy / d;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * 2. ERK4 + forward VDE (acados sim_erk, 4 stages, 1 step; integrator_type = "ERK":
 *    usv_guidance_ca1/acados_settings.py:194). State (x, Sx, Su) with S(0) = [I 0],
 *    dSx/dt = Jx Sx, dSu/dt = Jx Su + Ju.
 * ---------------------------------------------------------------------------------------- */
static void vde(int model, int nx, int nu, const double *x, const double *S, const double *u,
                double *fx, double *fS)
{
    double Jx[NXM * NXM], Ju[NXM * NUM];
    const int nz = nx + nu;
    int i, j, k;
    usv_model_f(model, x, u, fx);
    usv_model_jac(model, x, u, Jx, Ju);
    /* S is nx x (nx+nu), columns [Sx | Su] */
    for (i = 0; i < nx; i++)
        for (j = 0; j < nz; j++) {
            double acc = (j >= nx) ? Ju[i * nu + (j - nx)] : 0.0;
            for (k = 0; k < nx; k++) acc += Jx[i * nx + k] * S[k * nz + j];
            fS[i * nz + j] = acc;
        }
}

/* one RK4 step of the augmented system, in place on (x, S) */
static void rk4_step(int model, int nx, int nu, double h, double *x, double *S, const double *u)
{
    const int nz = nx + nu, ns = nx * nz;
    int i;
    double xs[NXM], Ss[NXM * NZM];
    double k1x[NXM], k2x[NXM], k3x[NXM], k4x[NXM];
    double k1S[NXM * NZM], k2S[NXM * NZM], k3S[NXM * NZM], k4S[NXM * NZM];
    vde(model, nx, nu, x, S, u, k1x, k1S);
    for (i = 0; i < nx; i++) xs[i] = x[i] + 0.5 * h * k1x[i];
    for (i = 0; i < ns; i++) Ss[i] = S[i] + 0.5 * h * k1S[i];
    vde(model, nx, nu, xs, Ss, u, k2x, k2S);
    for (i = 0; i < nx; i++) xs[i] = x[i] + 0.5 * h * k2x[i];
    for (i = 0; i < ns; i++) Ss[i] = S[i] + 0.5 * h * k2S[i];
    vde(model, nx, nu, xs, Ss, u, k3x, k3S);
    for (i = 0; i < nx; i++) xs[i] = x[i] + h * k3x[i];
    for (i = 0; i < ns; i++) Ss[i] = S[i] + h * k3S[i];
    vde(model, nx, nu, xs, Ss, u, k4x, k4S);
    for (i = 0; i < nx; i++) x[i] += h / 6.0 * (k1x[i] + 2.0 * k2x[i] + 2.0 * k3x[i] + k4x[i]);
    for (i = 0; i < ns; i++) S[i] += h / 6.0 * (k1S[i] + 2.0 * k2S[i] + 2.0 * k3S[i] + k4S[i]);
}

/* `steps` RK4 steps of size dt/steps over one shooting interval (acados sim_method_num_steps) */
void usv_erk_sens(int model, double dt, int steps, const double *x, const double *u, double *xn,
                  double *A, double *B)
{
    int nx, nu, nz, i, j, n;
    double S[NXM * NZM];
    usv_model_dims(model, &nx, &nu);
    nz = nx + nu;
    if (steps < 1) steps = 1;
    for (i = 0; i < nx * nz; i++) S[i] = 0.0;
    for (i = 0; i < nx; i++) { S[i * nz + i] = 1.0; xn[i] = x[i]; }
    for (n = 0; n < steps; n++) rk4_step(model, nx, nu, dt / steps, xn, S, u);
    for (i = 0; i < nx; i++)
        for (j = 0; j < nz; j++) {
            if (j < nx) A[i * nx + j] = S[i * nz + j];
            else B[i * nu + (j - nx)] = S[i * nz + j];
        }
}

void usv_rk4_sens(int model, double dt, const double *x, const double *u, double *xn, double *A,
                  double *B)
{
    usv_erk_sens(model, dt, 1, x, u, xn, A, B);
}

/* ------------------------------------------------------------------------------------------
 * 3. OCP definitions
 * ---------------------------------------------------------------------------------------- */
/* The QP solver's argument profile: what acados' ocp_qp_hpipm_opts_initialize_default leaves in d_ocp_qp_ipm_arg - HPIPM's
 * d_ocp_qp_ipm_arg_set_default(mode) followed by acados' overwrites - AS RECALLED (neither tree is under /root/reference: usv_oracle.h,
 * "PARITY UNPINNED"; DESIGN.md section 2 has the field-by-field table).  The reference selects the solver and touches none of its knobs
 * (scripts/usv_pf_ca/acados_settings.py:172,183-186; scripts/usv_guidance_ca1/acados_settings.py:190,198-204).
 *                     SPEED   BALANCE  ROBUST   acados' overwrite      R04 (this restatement up to round 5)
 *   mu0               10      10       100      1                      10
 *   alpha_min         1e-12   1e-12    1e-12    1e-8                   1e-12
 *   res_g/b/d/m_max   1e-8 each                 1e-6, 1e-8, 1e-8, 1e-8 the same
 *   iter_max          15      30       100      50                     50
 *   cond_pred_corr    1       1        1        -                      0
 *   itref_corr_max    0       2        4        -                      0
 * Returns 0, or -1 for an unknown mode. */
int usv_opts_profile(usv_opts *o, int mode)
{
    if (mode < USV_HPIPM_BALANCE || mode > USV_HPIPM_R04) return -1;
    o->qp_iter_max = 50;
    o->thr0 = 0.1;
    o->tol_stat = 1e-6;
    o->tol_eq = 1e-8;
    o->tol_ineq = 1e-8;
    o->tol_comp = 1e-8;
    o->riccati = USV_RICCATI_SQRT;
    o->cpc_factor = 2.0;
    o->mu0 = mode == USV_HPIPM_R04 ? 10.0 : 1.0;
    o->alpha_min = mode == USV_HPIPM_R04 ? 1e-12 : 1e-8;
    o->cond_pred_corr = mode == USV_HPIPM_R04 ? 0 : 1;
    o->itref_corr_max = mode == USV_HPIPM_BALANCE ? 2 : (mode == USV_HPIPM_ROBUST ? 4 : 0);
    return 0;
}

void usv_opts_defaults(usv_opts *o)
{
    usv_opts_profile(o, USV_HPIPM_BALANCE);
}

int usv_spec_defaults(usv_spec *s, int model, int N, double Tf, int K)
{
    int nx, nu, i;
    memset(s, 0, sizeof(*s));
    if (usv_model_dims(model, &nx, &nu)) return -1;
    if (model == USV_M0) K = 0;
    if (K < 0 || K > KM || N < 1) return -1;
    if (model == USV_MGEN && g_gen_ipx < 0 && K > 0) return -1;
    s->model = model;
    s->N = N;
    s->dt = Tf / N;
    s->K = K;
    s->nx = nx;
    s->nu = nu;
    s->ny = nx + nu;
    s->ny_e = nx;
    usv_opts_defaults(&s->opts);
    s->sim_steps = 1;
    s->nlp_max_iter = 100;
    s->nlp_tol[0] = s->nlp_tol[1] = s->nlp_tol[2] = s->nlp_tol[3] = 1e-6;
    /* Vx = [I;0], Vx_e = I : e.g. usv_guidance_ca1/acados_settings.py:92-103 */
    for (i = 0; i < nx; i++) {
        s->Vx[i * nx + i] = 1.0;
        s->Vx_e[i * nx + i] = 1.0;
    }
    if (model == USV_MGEN) return 0; /* dimensions only: weights / bounds come from the caller's OCP */
    if (model == USV_M0) {
        /* usv_acados/acados_settings.py:75-81,96-98,115-120; usv_model.py:129-139 */
        const double Q[5] = {1e3, 1e-3, 1e3, 1e-1, 1e-1}, R[2] = {1e-2, 1e-2};
        const double Qe[5] = {5e3, 5e-3, 5e3, 5e-1, 5e-1};
        const double lbx[5] = {-1.5, -1.5, -1.0, -30.0, -30.0}, ubx[5] = {1.5, 1.5, 1.0, 35.0, 35.0};
        for (i = 0; i < 5; i++) s->W[i * 7 + i] = Q[i];
        for (i = 0; i < 2; i++) s->W[(5 + i) * 7 + 5 + i] = R[i];
        for (i = 0; i < 5; i++) s->W_e[i * 5 + i] = Qe[i];
        s->Vu[5 * 2 + 0] = 1.0;
        s->Vu[6 * 2 + 1] = 1.0;
        s->nbx = 5;
        for (i = 0; i < 5; i++) { s->idxbx[i] = i; s->lbx[i] = lbx[i]; s->ubx[i] = ubx[i]; }
        s->nbu = 2;
        for (i = 0; i < 2; i++) { s->idxbu[i] = i; s->lbu[i] = -30.0; s->ubu[i] = 30.0; }
    } else if (model == USV_M1) {
        /* usv_guidance_ca1/acados_settings.py:75-80,96-97,105-108,118-120,138-178 */
        const double Q[8] = {0, 0, 0.05, 0.01, 0, 0, 0, 0}, Qe[8] = {0, 0, 0.1, 0.05, 0, 0, 0, 0};
        for (i = 0; i < 8; i++) s->W[i * 9 + i] = Q[i];
        s->W[8 * 9 + 8] = 0.2;
        for (i = 0; i < 8; i++) s->W_e[i * 8 + i] = Qe[i];
        s->Vu[8 * 1 + 0] = 1.0;
        s->nbx = 0;
        s->nbu = 1;
        s->idxbu[0] = 0; s->lbu[0] = -0.5; s->ubu[0] = 0.5;
        s->soft = 1;
        for (i = 0; i < K; i++) {
            s->uh[i] = 1000000.0;
            s->lsh[i] = -0.2; s->ush[i] = 0.0;
            s->zl[i] = 1.0; s->zu[i] = 1.0; s->Zl[i] = 0.0; s->Zu[i] = 0.0;
        }
    } else {
        /* usv_pf_ca/acados_settings.py:93-99,115-117 (Vu rows 8,9: reproduced, not "fixed"),
         * 134-139,151-158; usv_model.py:171-190 */
        const double Q[14] = {0, 0.3, 0.3, 80.0, 0, 0, 0.8, 0, 0, 0, 0, 0, 0.0001, 0.0001};
        const double Qe[14] = {0, 0.5, 0.5, 100.0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0.0005, 0.0005};
        const int idx[5] = {3, 4, 5, 12, 13};
        const double lbx[5] = {-2.0, -2.0, -10.0, -30.0, -30.0}, ubx[5] = {2.0, 2.0, 10.0, 36.5, 36.5};
        for (i = 0; i < 14; i++) s->W[i * 16 + i] = Q[i];
        for (i = 0; i < 14; i++) s->W_e[i * 14 + i] = Qe[i];
        s->Vu[8 * 2 + 0] = 1.0;
        s->Vu[9 * 2 + 1] = 1.0;
        s->nbx = 5;
        for (i = 0; i < 5; i++) { s->idxbx[i] = idx[i]; s->lbx[i] = lbx[i]; s->ubx[i] = ubx[i]; }
        s->nbu = 2;
        for (i = 0; i < 2; i++) { s->idxbu[i] = i; s->lbu[i] = -30.0; s->ubu[i] = 30.0; }
        s->soft = 0;
        for (i = 0; i < K; i++) s->uh[i] = 1000000.0;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * 4. QP container + linearisation (acados preparation phase: sim_erk, ocp_nlp_cost_ls,
 *    ocp_nlp_constraints_bgh)
 * ---------------------------------------------------------------------------------------- */
static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

usv_qp *usv_qp_alloc(const usv_spec *s)
{
    usv_qp *q = (usv_qp *)calloc(1, sizeof(usv_qp));
    const int N = s->N, nx = s->nx, nu = s->nu, nz = nx + nu, K = s->K;
    int i;
    q->N = N; q->nx = nx; q->nu = nu; q->nz = nz; q->K = K;
    q->nbu = s->nbu; q->nbx = s->nbx; q->soft = s->soft;
    for (i = 0; i < s->nbu; i++) q->idxbu[i] = s->idxbu[i];
    for (i = 0; i < s->nbx; i++) q->idxbx[i] = s->idxbx[i];
    usv_model_pos_idx(s->model, &q->ipx, &q->ipy);
    q->A = dalloc((size_t)N * nx * nx);
    q->B = dalloc((size_t)N * nx * nu);
    q->b = dalloc((size_t)N * nx);
    q->H = dalloc((size_t)(N + 1) * nz * nz);
    q->g = dalloc((size_t)(N + 1) * nz);
    q->dx0 = dalloc(nx);
    q->lbu = dalloc((size_t)N * (s->nbu ? s->nbu : 1));
    q->ubu = dalloc((size_t)N * (s->nbu ? s->nbu : 1));
    q->lbx = dalloc((size_t)(N + 1) * (s->nbx ? s->nbx : 1));
    q->ubx = dalloc((size_t)(N + 1) * (s->nbx ? s->nbx : 1));
    q->Cxy = dalloc((size_t)(N + 1) * (K ? K : 1) * 2);
    q->lg = dalloc((size_t)(N + 1) * (K ? K : 1));
    q->ug = dalloc((size_t)(N + 1) * (K ? K : 1));
    q->zl = dalloc(K + s->nbx); q->zu = dalloc(K + s->nbx); q->Zl = dalloc(K + s->nbx); q->Zu = dalloc(K + s->nbx);
    q->lsl = dalloc(K + s->nbx); q->lsu = dalloc(K + s->nbx);
    for (i = 0; i < s->nbx; i++) q->sbx[i] = s->sbx[i];
    return q;
}

void usv_qp_free(usv_qp *q)
{
    if (!q) return;
    free(q->scratch);
    free(q->A); free(q->B); free(q->b); free(q->H); free(q->g); free(q->dx0);
    free(q->lbu); free(q->ubu); free(q->lbx); free(q->ubx); free(q->Cxy); free(q->lg); free(q->ug);
    free(q->zl); free(q->zu); free(q->Zl); free(q->Zu); free(q->lsl); free(q->lsu);
    free(q);
}

usv_qp_sol *usv_qp_sol_alloc(const usv_qp *q)
{
    usv_qp_sol *s = (usv_qp_sol *)calloc(1, sizeof(usv_qp_sol));
    const int N = q->N, K = q->K ? q->K : 1, nbu = q->nbu ? q->nbu : 1, nbx = q->nbx ? q->nbx : 1;
    s->dz = dalloc((size_t)(N + 1) * q->nz);
    s->pi = dalloc((size_t)(N + 1) * q->nx);
    s->lam_bu = dalloc((size_t)N * 2 * nbu); s->t_bu = dalloc((size_t)N * 2 * nbu);
    s->lam_bx = dalloc((size_t)(N + 1) * 2 * nbx); s->t_bx = dalloc((size_t)(N + 1) * 2 * nbx);
    s->lam_g = dalloc((size_t)(N + 1) * 2 * K); s->t_g = dalloc((size_t)(N + 1) * 2 * K);
    s->sl = dalloc((size_t)(N + 1) * K); s->su = dalloc((size_t)(N + 1) * K);
    s->lam_s = dalloc((size_t)(N + 1) * 2 * K); s->t_s = dalloc((size_t)(N + 1) * 2 * K);
    s->sl_bx = dalloc((size_t)(N + 1) * nbx); s->su_bx = dalloc((size_t)(N + 1) * nbx);
    s->lam_sbx = dalloc((size_t)(N + 1) * 2 * nbx); s->t_sbx = dalloc((size_t)(N + 1) * 2 * nbx);
    return s;
}

void usv_qp_sol_free(usv_qp_sol *s)
{
    if (!s) return;
    free(s->dz); free(s->pi); free(s->lam_bu); free(s->t_bu); free(s->lam_bx); free(s->t_bx);
    free(s->lam_g); free(s->t_g); free(s->sl); free(s->su); free(s->lam_s); free(s->t_s);
    free(s->sl_bx); free(s->su_bx); free(s->lam_sbx); free(s->t_sbx);
    free(s);
}

/* GN Hessian and gradient of 0.5*scale*||Vx x + Vu u - yref||_W^2 in [u;x] ordering
 * (acados ocp_nlp_cost_ls; cost_type LINEAR_LS: usv_guidance_ca1/acados_settings.py:83-84). */
static void ls_cost(int nx, int nu, int ny, double scale, const double *W, const double *Vx,
                    const double *Vu, const double *x, const double *u, const double *yref,
                    double *H, double *g)
{
    const int nz = nx + nu;
    double V[NYM * NZM], res[NYM], WV[NYM * NZM], Wr[NYM];
    int i, j, k;
    for (i = 0; i < ny; i++) {
        for (j = 0; j < nu; j++) V[i * nz + j] = Vu ? Vu[i * nu + j] : 0.0;
        for (j = 0; j < nx; j++) V[i * nz + nu + j] = Vx[i * nx + j];
    }
    for (i = 0; i < ny; i++) {
        double r = -yref[i];
        for (j = 0; j < nu; j++) r += V[i * nz + j] * (u ? u[j] : 0.0);
        for (j = 0; j < nx; j++) r += V[i * nz + nu + j] * x[j];
        res[i] = r;
    }
    for (i = 0; i < ny; i++) {
        double a = 0.0;
        for (k = 0; k < ny; k++) a += W[i * ny + k] * res[k];
        Wr[i] = a;
        for (j = 0; j < nz; j++) {
            double acc = 0.0;
            for (k = 0; k < ny; k++) acc += W[i * ny + k] * V[k * nz + j];
            WV[i * nz + j] = acc;
        }
    }
    for (i = 0; i < nz; i++) {
        double a = 0.0;
        for (k = 0; k < ny; k++) a += V[k * nz + i] * Wr[k];
        g[i] = scale * a;
        for (j = 0; j < nz; j++) {
            double acc = 0.0;
            for (k = 0; k < ny; k++) acc += V[k * nz + i] * WV[k * nz + j];
            H[i * nz + j] = scale * acc;
        }
    }
}

void usv_linearize(const usv_spec *s, const double *x, const double *u, const double *x0,
                   const double *yref, const double *yref_e, const double *p, const double *lh,
                   usv_qp *q)
{
    const int N = s->N, nx = s->nx, nu = s->nu, nz = nx + nu, K = s->K, ny = s->ny;
    int k, i;
    for (i = 0; i < nx; i++) q->dx0[i] = x0[i] - x[i];
    for (k = 0; k < N; k++) {
        double xn[NXM];
        usv_erk_sens(s->model, s->dt, s->sim_steps, x + k * nx, u + k * nu, xn, q->A + (size_t)k * nx * nx,
                     q->B + (size_t)k * nx * nu);
        for (i = 0; i < nx; i++) q->b[k * nx + i] = xn[i] - x[(k + 1) * nx + i];
        ls_cost(nx, nu, ny, s->dt, s->W, s->Vx, s->Vu, x + k * nx, u + k * nu, yref + k * ny,
                q->H + (size_t)k * nz * nz, q->g + k * nz);
        for (i = 0; i < s->nbu; i++) {
            q->lbu[k * s->nbu + i] = s->lbu[i] - u[k * nu + s->idxbu[i]];
            q->ubu[k * s->nbu + i] = s->ubu[i] - u[k * nu + s->idxbu[i]];
        }
        for (i = 0; i < s->nbx; i++) {
            q->lbx[k * s->nbx + i] = s->lbx[i] - x[k * nx + s->idxbx[i]];
            q->ubx[k * s->nbx + i] = s->ubx[i] - x[k * nx + s->idxbx[i]];
        }
        if (K > 0) {
            double h[KM];
            usv_model_h(s->model, K, x + k * nx, p + (size_t)k * 2 * K, h, q->Cxy + (size_t)k * 2 * K);
            for (i = 0; i < K; i++) {
                q->lg[k * K + i] = lh[k * K + i] - h[i];
                q->ug[k * K + i] = s->uh[i] - h[i];
            }
        }
    }
    { /* terminal: H_N has zero u rows/cols */
        double He[NXM * NXM], ge[NXM];
        int j;
        ls_cost(nx, 0, s->ny_e, 1.0, s->W_e, s->Vx_e, NULL, x + N * nx, NULL, yref_e, He, ge);
        for (i = 0; i < nz * nz; i++) q->H[(size_t)N * nz * nz + i] = 0.0;
        for (i = 0; i < nz; i++) q->g[N * nz + i] = 0.0;
        for (i = 0; i < nx; i++) {
            q->g[N * nz + nu + i] = ge[i];
            for (j = 0; j < nx; j++) q->H[(size_t)N * nz * nz + (nu + i) * nz + nu + j] = He[i * nx + j];
        }
    }
    for (i = 0; i < K; i++) { /* slack cost scaled like the stage cost (dt) */
        q->zl[i] = s->dt * s->zl[i]; q->zu[i] = s->dt * s->zu[i];
        q->Zl[i] = s->dt * s->Zl[i]; q->Zu[i] = s->dt * s->Zu[i];
        q->lsl[i] = s->lsh[i]; q->lsu[i] = s->ush[i];
    }
    for (i = 0; i < s->nbx; i++) { /* soft state bounds: the same scaling */
        q->zl[K + i] = s->dt * s->zl_bx[i]; q->zu[K + i] = s->dt * s->zu_bx[i];
        q->Zl[K + i] = s->dt * s->Zl_bx[i]; q->Zu[K + i] = s->dt * s->Zu_bx[i];
        q->lsl[K + i] = s->lsbx[i]; q->lsu[K + i] = s->usbx[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * 5. OCP-QP interior point (HPIPM d_ocp_qp_ipm_solve restated): Mehrotra predictor-corrector,
 *    inequality rows eliminated into the stage Hessians, backward Riccati + forward sweep.
 * ---------------------------------------------------------------------------------------- */
#define MAXR (NUM + NXM + KM)

typedef struct row_t { /* one two-sided inequality row  dl <= c'z (+sl) ,  c'z (-su) <= du */
    int kind;          /* 0 box (unit vector at j0), 1 obstacle row (cx at j0, cy at j1) */
    int j0, j1;
    double cx, cy;
    int soft, ks;      /* ks: index into soft data */
    double dl, du;
    double ll, lu, tl, tu;          /* multipliers / slacks of the two sides */
    double sl, su, lsl, lsu, tsl, tsu; /* soft: slack values, their bound multipliers + slacks */
    double rdl, rdu, rsl, rsu, rdsl, rdsu; /* residuals */
    double ml, mu_, msl, msu;       /* complementarity targets (m-hat) */
    double dll, dlu, dtl, dtu, dsl, dsu, dlsl, dlsu, dtsl, dtsu; /* step */
    double Gl, Gu, Dl, Du, rhol, rhou; /* eliminations */
} row_t;

typedef struct stage_t {
    int m;
    row_t r[MAXR];
    double z[NZM], pi[NXM];
    double rg[NZM], rb[NXM];
    double Ht[NZM * NZM], gt[NZM];
    double Luu[NUM * NUM], Lxu[NXM * NUM], P[NXM * NXM], Pb[NXM];
    double p[NXM], lu[NUM];
    double dz[NZM], dpi[NXM];
} stage_t;

static double row_dot(const row_t *r, const double *z)
{
    return r->kind == 0 ? z[r->j0] : r->cx * z[r->j0] + r->cy * z[r->j1];
}

static void row_axpy(const row_t *r, double a, double *g)
{
    if (r->kind == 0) g[r->j0] += a;
    else { g[r->j0] += a * r->cx; g[r->j1] += a * r->cy; }
}

static void row_rank1(const row_t *r, double a, double *H, int nz)
{
    if (r->kind == 0) H[r->j0 * nz + r->j0] += a;
    else {
        H[r->j0 * nz + r->j0] += a * r->cx * r->cx;
        H[r->j0 * nz + r->j1] += a * r->cx * r->cy;
        H[r->j1 * nz + r->j0] += a * r->cy * r->cx;
        H[r->j1 * nz + r->j1] += a * r->cy * r->cy;
    }
}

/* lower Cholesky of the leading n x n block of a (row-major, leading dim ld); 0 ok.
 * The first `nstrict` pivots must be positive; a non-positive later pivot zeroes its column
 * (BLASFEO dpotrf_l semantics: inverse pivot := 0), which is what lets HPIPM run on the
 * positive-SEMIdefinite state blocks these OCPs have (zero weights, states without dynamics). */
static int chol_lower(double *a, int n, int ld, int nstrict)
{
    int i, j, k;
    for (j = 0; j < n; j++) {
        double d = a[j * ld + j];
        for (k = 0; k < j; k++) d -= a[j * ld + k] * a[j * ld + k];
        if (d != d) return 1;
        if (!(d > 0.0)) {
            if (j < nstrict) return 1;
            a[j * ld + j] = 0.0;
            for (i = j + 1; i < n; i++) a[i * ld + j] = 0.0;
            continue;
        }
        d = sqrt(d);
        a[j * ld + j] = d;
        for (i = j + 1; i < n; i++) {
            double v = a[i * ld + j];
            for (k = 0; k < j; k++) v -= a[i * ld + k] * a[j * ld + k];
            a[i * ld + j] = v / d;
        }
    }
    return 0;
}

typedef struct ipm_ws {
    const usv_qp *q;
    stage_t *st;
    double res[4], mu;
    int nc; /* number of (lambda,t) pairs */
} ipm_ws;

static void build_rows(ipm_ws *w)
{
    const usv_qp *q = w->q;
    const int N = q->N, nu = q->nu, K = q->K;
    int k, i;
    w->nc = 0;
    for (k = 0; k <= N; k++) {
        stage_t *s = &w->st[k];
        int m = 0;
        memset(s, 0, sizeof(*s));
        if (k < N)
            for (i = 0; i < q->nbu; i++) {
                row_t *r = &s->r[m++];
                r->kind = 0; r->j0 = q->idxbu[i];
                r->dl = q->lbu[k * q->nbu + i]; r->du = q->ubu[k * q->nbu + i];
            }
        if (k >= 1 && k < N) {
            for (i = 0; i < q->nbx; i++) {
                row_t *r = &s->r[m++];
                r->kind = 0; r->j0 = nu + q->idxbx[i];
                r->dl = q->lbx[k * q->nbx + i]; r->du = q->ubx[k * q->nbx + i];
                r->soft = q->sbx[i]; r->ks = K + i;
            }
            for (i = 0; i < K; i++) {
                row_t *r = &s->r[m++];
                r->kind = 1; r->j0 = nu + q->ipx; r->j1 = nu + q->ipy;
                r->cx = q->Cxy[(size_t)k * 2 * K + 2 * i]; r->cy = q->Cxy[(size_t)k * 2 * K + 2 * i + 1];
                r->dl = q->lg[k * K + i]; r->du = q->ug[k * K + i];
                r->soft = q->soft; r->ks = i;
            }
        }
        s->m = m;
        for (i = 0; i < m; i++) w->nc += s->r[i].soft ? 4 : 2;
    }
}

static void init_cold(ipm_ws *w, const usv_opts *o)
{
    const usv_qp *q = w->q;
    int k, i;
    for (k = 0; k <= q->N; k++) {
        stage_t *s = &w->st[k];
        for (i = 0; i < s->m; i++) {
            row_t *r = &s->r[i];
            const double v = row_dot(r, s->z); /* z = 0 */
            r->sl = 0.0; r->su = 0.0;
            r->tl = fmax(v + r->sl - r->dl, o->thr0);
            r->tu = fmax(r->du - v + r->su, o->thr0);
            r->ll = o->mu0 / r->tl;
            r->lu = o->mu0 / r->tu;
            if (r->soft) {
                r->tsl = fmax(r->sl - q->lsl[r->ks], o->thr0);
                r->tsu = fmax(r->su - q->lsu[r->ks], o->thr0);
                r->lsl = o->mu0 / r->tsl;
                r->lsu = o->mu0 / r->tsu;
            }
        }
    }
}

static void residuals(ipm_ws *w)
{
    const usv_qp *q = w->q;
    const int N = q->N, nx = q->nx, nu = q->nu, nz = q->nz;
    double rg = 0, rb = 0, rd = 0, rm = 0, mu = 0;
    int k, i, j;
    for (i = 0; i < nx; i++) { /* x0 equality */
        const double e = q->dx0[i] - w->st[0].z[nu + i];
        w->st[0].dz[nu + i] = e; /* reused as the forward-sweep start by the caller */
        rb = fmax(rb, fabs(e));
    }
    for (k = 0; k <= N; k++) {
        stage_t *s = &w->st[k];
        const double *H = q->H + (size_t)k * nz * nz;
        for (i = 0; i < nz; i++) {
            double a = q->g[k * nz + i];
            for (j = 0; j < nz; j++) a += H[i * nz + j] * s->z[j];
            s->rg[i] = a;
        }
        if (k < N) {
            const double *A = q->A + (size_t)k * nx * nx, *B = q->B + (size_t)k * nx * nu;
            const stage_t *sn = &w->st[k + 1];
            for (i = 0; i < nx; i++) {
                double a = q->b[k * nx + i] - sn->z[nu + i];
                for (j = 0; j < nx; j++) a += A[i * nx + j] * s->z[nu + j];
                for (j = 0; j < nu; j++) a += B[i * nu + j] * s->z[j];
                s->rb[i] = a;
                rb = fmax(rb, fabs(a));
            }
            for (j = 0; j < nu; j++) {
                double a = 0;
                for (i = 0; i < nx; i++) a += B[i * nu + j] * sn->pi[i];
                s->rg[j] += a;
            }
            for (j = 0; j < nx; j++) {
                double a = 0;
                for (i = 0; i < nx; i++) a += A[i * nx + j] * sn->pi[i];
                s->rg[nu + j] += a;
            }
        }
        if (k >= 1)
            for (i = 0; i < nx; i++) s->rg[nu + i] -= s->pi[i];
        for (i = 0; i < s->m; i++) {
            row_t *r = &s->r[i];
            const double v = row_dot(r, s->z);
            row_axpy(r, -(r->ll - r->lu), s->rg);
            r->rdl = v + r->sl - r->dl - r->tl;
            r->rdu = r->du - v + r->su - r->tu;
            rd = fmax(rd, fmax(fabs(r->rdl), fabs(r->rdu)));
            rm = fmax(rm, fmax(r->ll * r->tl, r->lu * r->tu));
            mu += r->ll * r->tl + r->lu * r->tu;
            if (r->soft) {
                r->rsl = q->Zl[r->ks] * r->sl + q->zl[r->ks] - r->ll - r->lsl;
                r->rsu = q->Zu[r->ks] * r->su + q->zu[r->ks] - r->lu - r->lsu;
                r->rdsl = r->sl - q->lsl[r->ks] - r->tsl;
                r->rdsu = r->su - q->lsu[r->ks] - r->tsu;
                rg = fmax(rg, fmax(fabs(r->rsl), fabs(r->rsu)));
                rd = fmax(rd, fmax(fabs(r->rdsl), fabs(r->rdsu)));
                rm = fmax(rm, fmax(r->lsl * r->tsl, r->lsu * r->tsu));
                mu += r->lsl * r->tsl + r->lsu * r->tsu;
            }
        }
        for (i = (k == N ? nu : 0); i < nz; i++) {
            if (k == 0 && i >= nu) continue; /* x_0 is not a variable */
            rg = fmax(rg, fabs(s->rg[i]));
        }
    }
    w->res[0] = rg; w->res[1] = rb; w->res[2] = rd; w->res[3] = rm;
    w->mu = w->nc ? mu / w->nc : 0.0;
}

/* eliminate the inequality rows into Ht (if `fact`) and gt */
static void reduce_rows(ipm_ws *w, int fact)
{
    const usv_qp *q = w->q;
    const int N = q->N, nz = q->nz;
    int k, i;
    for (k = 0; k <= N; k++) {
        stage_t *s = &w->st[k];
        if (fact) memcpy(s->Ht, q->H + (size_t)k * nz * nz, sizeof(double) * nz * nz);
        for (i = 0; i < nz; i++) s->gt[i] = s->rg[i];
        for (i = 0; i < s->m; i++) {
            row_t *r = &s->r[i];
            double Ghl, Ghu, gl, gu;
            r->Gl = r->ll / r->tl;
            r->Gu = r->lu / r->tu;
            if (r->soft) {
                const double Gsl = r->lsl / r->tsl, Gsu = r->lsu / r->tsu;
                r->Dl = q->Zl[r->ks] + r->Gl + Gsl;
                r->Du = q->Zu[r->ks] + r->Gu + Gsu;
                r->rhol = -r->rsl - r->ml / r->tl - r->Gl * r->rdl - r->msl / r->tsl - Gsl * r->rdsl;
                r->rhou = -r->rsu - r->mu_ / r->tu - r->Gu * r->rdu - r->msu / r->tsu - Gsu * r->rdsu;
                Ghl = r->Gl * (1.0 - r->Gl / r->Dl);
                Ghu = r->Gu * (1.0 - r->Gu / r->Du);
                gl = r->ml / r->tl + r->Gl * r->rdl + r->Gl * r->rhol / r->Dl;
                gu = r->mu_ / r->tu + r->Gu * r->rdu + r->Gu * r->rhou / r->Du;
            } else {
                Ghl = r->Gl; Ghu = r->Gu;
                gl = r->ml / r->tl + r->Gl * r->rdl;
                gu = r->mu_ / r->tu + r->Gu * r->rdu;
            }
            if (fact) row_rank1(r, Ghl + Ghu, s->Ht, nz);
            row_axpy(r, gl - gu, s->gt);
        }
    }
}

/* backward Riccati factorisation. SQRT: HPIPM d_ocp_qp_fact_solve_kkt_step structure
 * (W = [B A]' L_{k+1}; G = Ht + W W'; potrf); CLASSIC: G = Ht + [B A]' P_{k+1} [B A]. */
static int riccati_factor(ipm_ws *w, int mode)
{
    const usv_qp *q = w->q;
    const int N = q->N, nx = q->nx, nu = q->nu, nz = q->nz;
    double Lxx[NXM * NXM]; /* Cholesky factor of P_{k+1} (sqrt mode) */
    int k, i, j, l;
    {
        stage_t *s = &w->st[N];
        for (i = 0; i < nx; i++)
            for (j = 0; j < nx; j++) s->P[i * nx + j] = s->Ht[(nu + i) * nz + nu + j];
        if (mode == USV_RICCATI_SQRT) {
            memcpy(Lxx, s->P, sizeof(double) * nx * nx);
            if (chol_lower(Lxx, nx, nx, 0)) return 1;
            for (i = 0; i < nx; i++)
                for (j = i + 1; j < nx; j++) Lxx[i * nx + j] = 0.0;
        }
    }
    for (k = N - 1; k >= 0; k--) {
        stage_t *s = &w->st[k];
        const stage_t *sn = &w->st[k + 1];
        const double *A = q->A + (size_t)k * nx * nx, *B = q->B + (size_t)k * nx * nu;
        double BA[NXM * NZM]; /* [B A] : nx x nz */
        double G[NZM * NZM];
        for (i = 0; i < nx; i++) {
            for (j = 0; j < nu; j++) BA[i * nz + j] = B[i * nu + j];
            for (j = 0; j < nx; j++) BA[i * nz + nu + j] = A[i * nx + j];
        }
        memcpy(G, s->Ht, sizeof(double) * nz * nz);
        if (mode == USV_RICCATI_SQRT) {
            double Wm[NZM * NXM]; /* W = [B A]' Lxx : nz x nx */
            for (i = 0; i < nz; i++)
                for (j = 0; j < nx; j++) {
                    double a = 0;
                    for (l = j; l < nx; l++) a += BA[l * nz + i] * Lxx[l * nx + j];
                    Wm[i * nx + j] = a;
                }
            for (i = 0; i < nz; i++)
                for (j = 0; j <= i; j++) {
                    double a = 0;
                    for (l = 0; l < nx; l++) a += Wm[i * nx + l] * Wm[j * nx + l];
                    G[i * nz + j] += a;
                    if (j != i) G[j * nz + i] = G[i * nz + j];
                }
        } else {
            double T[NZM * NXM]; /* T = [B A]' P : nz x nx */
            for (i = 0; i < nz; i++)
                for (j = 0; j < nx; j++) {
                    double a = 0;
                    for (l = 0; l < nx; l++) a += BA[l * nz + i] * sn->P[l * nx + j];
                    T[i * nx + j] = a;
                }
            for (i = 0; i < nz; i++)
                for (j = 0; j < nz; j++) {
                    double a = 0;
                    for (l = 0; l < nx; l++) a += T[i * nx + l] * BA[l * nz + j];
                    G[i * nz + j] += a;
                }
        }
        /* P_{k+1} b_k for the vector recursion (b = dynamics residual, fixed per factorisation) */
        for (i = 0; i < nx; i++) {
            double a = 0;
            for (j = 0; j < nx; j++) a += sn->P[i * nx + j] * s->rb[j];
            s->Pb[i] = a;
        }
        if (mode == USV_RICCATI_SQRT) {
            if (chol_lower(G, nz, nz, nu)) return 1;
            for (i = 0; i < nu; i++)
                for (j = 0; j < nu; j++) s->Luu[i * nu + j] = (j <= i) ? G[i * nz + j] : 0.0;
            for (i = 0; i < nx; i++)
                for (j = 0; j < nu; j++) s->Lxu[i * nu + j] = G[(nu + i) * nz + j];
            for (i = 0; i < nx; i++)
                for (j = 0; j < nx; j++) Lxx[i * nx + j] = (j <= i) ? G[(nu + i) * nz + nu + j] : 0.0;
            for (i = 0; i < nx; i++)
                for (j = 0; j <= i; j++) {
                    double a = 0;
                    for (l = 0; l <= j; l++) a += Lxx[i * nx + l] * Lxx[j * nx + l];
                    s->P[i * nx + j] = a;
                    s->P[j * nx + i] = a;
                }
        } else {
            for (i = 0; i < nu; i++)
                for (j = 0; j < nu; j++) s->Luu[i * nu + j] = G[i * nz + j];
            if (chol_lower(s->Luu, nu, nu, nu)) return 1;
            for (i = 0; i < nu; i++)
                for (j = i + 1; j < nu; j++) s->Luu[i * nu + j] = 0.0;
            for (i = 0; i < nx; i++) /* Lxu = G_xu Luu^-T */
                for (j = 0; j < nu; j++) {
                    double a = G[(nu + i) * nz + j];
                    for (l = 0; l < j; l++) a -= s->Lxu[i * nu + l] * s->Luu[j * nu + l];
                    s->Lxu[i * nu + j] = a / s->Luu[j * nu + j];
                }
            for (i = 0; i < nx; i++)
                for (j = 0; j < nx; j++) {
                    double a = G[(nu + i) * nz + nu + j];
                    for (l = 0; l < nu; l++) a -= s->Lxu[i * nu + l] * s->Lxu[j * nu + l];
                    s->P[i * nx + j] = a;
                }
        }
    }
    return 0;
}

/* backward vector recursion + forward sweep with the current gt / rb; dz[0].x preset */
static void riccati_solve(ipm_ws *w)
{
    const usv_qp *q = w->q;
    const int N = q->N, nx = q->nx, nu = q->nu, nz = q->nz;
    int k, i, j;
    for (i = 0; i < nx; i++) w->st[N].p[i] = w->st[N].gt[nu + i];
    for (k = N - 1; k >= 0; k--) {
        stage_t *s = &w->st[k];
        const stage_t *sn = &w->st[k + 1];
        const double *A = q->A + (size_t)k * nx * nx, *B = q->B + (size_t)k * nx * nu;
        double h[NXM], rq[NZM];
        for (i = 0; i < nx; i++) h[i] = s->Pb[i] + sn->p[i];
        for (j = 0; j < nu; j++) {
            double a = s->gt[j];
            for (i = 0; i < nx; i++) a += B[i * nu + j] * h[i];
            rq[j] = a;
        }
        for (j = 0; j < nx; j++) {
            double a = s->gt[nu + j];
            for (i = 0; i < nx; i++) a += A[i * nx + j] * h[i];
            rq[nu + j] = a;
        }
        for (i = 0; i < nu; i++) { /* lu = Luu^-1 ru */
            double a = rq[i];
            for (j = 0; j < i; j++) a -= s->Luu[i * nu + j] * s->lu[j];
            s->lu[i] = a / s->Luu[i * nu + i];
        }
        for (i = 0; i < nx; i++) {
            double a = rq[nu + i];
            for (j = 0; j < nu; j++) a -= s->Lxu[i * nu + j] * s->lu[j];
            s->p[i] = a;
        }
    }
    for (k = 0; k < N; k++) {
        stage_t *s = &w->st[k];
        stage_t *sn = &w->st[k + 1];
        const double *A = q->A + (size_t)k * nx * nx, *B = q->B + (size_t)k * nx * nu;
        double t[NUM];
        for (j = 0; j < nu; j++) {
            double a = s->lu[j];
            for (i = 0; i < nx; i++) a += s->Lxu[i * nu + j] * s->dz[nu + i];
            t[j] = a;
        }
        for (i = nu - 1; i >= 0; i--) { /* du = -Luu^-T t */
            double a = t[i];
            for (j = i + 1; j < nu; j++) a -= s->Luu[j * nu + i] * t[j];
            t[i] = a / s->Luu[i * nu + i];
        }
        for (j = 0; j < nu; j++) s->dz[j] = -t[j];
        for (i = 0; i < nx; i++) {
            double a = s->rb[i];
            for (j = 0; j < nx; j++) a += A[i * nx + j] * s->dz[nu + j];
            for (j = 0; j < nu; j++) a += B[i * nu + j] * s->dz[j];
            sn->dz[nu + i] = a;
        }
        for (i = 0; i < nx; i++) {
            double a = sn->p[i];
            for (j = 0; j < nx; j++) a += sn->P[i * nx + j] * sn->dz[nu + j];
            sn->dpi[i] = a;
        }
    }
    for (j = 0; j < nu; j++) w->st[N].dz[j] = 0.0;
    (void)nz;
}

static void expand_rows(ipm_ws *w)
{
    const usv_qp *q = w->q;
    int k, i;
    for (k = 0; k <= q->N; k++) {
        stage_t *s = &w->st[k];
        for (i = 0; i < s->m; i++) {
            row_t *r = &s->r[i];
            const double wv = row_dot(r, s->dz);
            if (r->soft) {
                r->dsl = (r->rhol - r->Gl * wv) / r->Dl;
                r->dsu = (r->rhou + r->Gu * wv) / r->Du;
                r->dtsl = r->dsl + r->rdsl;
                r->dtsu = r->dsu + r->rdsu;
                r->dlsl = -(r->msl + r->lsl * r->dtsl) / r->tsl;
                r->dlsu = -(r->msu + r->lsu * r->dtsu) / r->tsu;
            } else {
                r->dsl = 0.0; r->dsu = 0.0;
            }
            r->dtl = wv + r->dsl + r->rdl;
            r->dtu = -wv + r->dsu + r->rdu;
            r->dll = -(r->ml + r->ll * r->dtl) / r->tl;
            r->dlu = -(r->mu_ + r->lu * r->dtu) / r->tu;
        }
    }
}

static double ratio(double v, double dv, double a) { return (dv < 0.0 && -v / dv < a) ? -v / dv : a; }

static double step_length(ipm_ws *w)
{
    double a = 1.0;
    int k, i;
    for (k = 0; k <= w->q->N; k++) {
        stage_t *s = &w->st[k];
        for (i = 0; i < s->m; i++) {
            row_t *r = &s->r[i];
            a = ratio(r->ll, r->dll, a); a = ratio(r->lu, r->dlu, a);
            a = ratio(r->tl, r->dtl, a); a = ratio(r->tu, r->dtu, a);
            if (r->soft) {
                a = ratio(r->lsl, r->dlsl, a); a = ratio(r->lsu, r->dlsu, a);
                a = ratio(r->tsl, r->dtsl, a); a = ratio(r->tsu, r->dtsu, a);
            }
        }
    }
    return a;
}

/* ---- HPIPM options beyond the plain Mehrotra iteration (usv_opts: off by default) -------------------------------------------------
 * Restated from HPIPM's published d_ocp_qp_ipm_solve (cond_pred_corr, itref_corr_max) as recalled - the acados tree is absent here. */

/* duality measure after a step of length a along the current direction */
static double mu_after(const ipm_ws *w, double a)
{
    double m = 0.0;
    int k, i;
    for (k = 0; k <= w->q->N; k++)
        for (i = 0; i < w->st[k].m; i++) {
            const row_t *r = &w->st[k].r[i];
            m += (r->ll + a * r->dll) * (r->tl + a * r->dtl) + (r->lu + a * r->dlu) * (r->tu + a * r->dtu);
            if (r->soft)
                m += (r->lsl + a * r->dlsl) * (r->tsl + a * r->dtsl) + (r->lsu + a * r->dlsu) * (r->tsu + a * r->dtsu);
        }
    return w->nc ? m / w->nc : 0.0;
}

/* One round of iterative refinement of the step held in (dz, dpi, row steps): the residual of every equation of the Newton system at
 * that step becomes the right-hand side of a second solve on the same factorisation, whose solution is added (HPIPM
 * d_ocp_qp_res_compute_lin + d_ocp_qp_solve_kkt_step).  Returns 0 without solving when the residual norms (stat, eq, ineq, comp)
 * are all below tol[] already. */
static int refine_step(ipm_ws *w, const double *tol)
{
    const usv_qp *q = w->q;
    const int N = q->N, nx = q->nx, nu = q->nu, nz = q->nz;
    stage_t *keep = (stage_t *)malloc(sizeof(stage_t) * ((size_t)N + 1));
    double nrm[4] = {0.0, 0.0, 0.0, 0.0};
    int k, i, j;
    memcpy(keep, w->st, sizeof(stage_t) * ((size_t)N + 1));
    for (k = 0; k <= N; k++) { /* residuals of the linear system at the current step, written over the rhs fields */
        stage_t *s = &w->st[k];
        const stage_t *o = &keep[k];
        const double *H = q->H + (size_t)k * nz * nz;
        for (i = 0; i < nz; i++) {
            double a = o->rg[i];
            for (j = 0; j < nz; j++) a += H[i * nz + j] * o->dz[j];
            s->rg[i] = a;
        }
        if (k < N) {
            const double *A = q->A + (size_t)k * nx * nx, *B = q->B + (size_t)k * nx * nu;
            const stage_t *on = &keep[k + 1];
            for (i = 0; i < nx; i++) {
                double a = o->rb[i] - on->dz[nu + i];
                for (j = 0; j < nx; j++) a += A[i * nx + j] * o->dz[nu + j];
                for (j = 0; j < nu; j++) a += B[i * nu + j] * o->dz[j];
                s->rb[i] = a;
                nrm[1] = fmax(nrm[1], fabs(a));
            }
            for (j = 0; j < nu; j++) for (i = 0; i < nx; i++) s->rg[j] += B[i * nu + j] * on->dpi[i];
            for (j = 0; j < nx; j++) for (i = 0; i < nx; i++) s->rg[nu + j] += A[i * nx + j] * on->dpi[i];
        }
        if (k >= 1) for (i = 0; i < nx; i++) s->rg[nu + i] -= o->dpi[i];
        for (i = 0; i < s->m; i++) {
            row_t *r = &s->r[i];
            const row_t *c = &o->r[i];
            const double wv = row_dot(c, o->dz);
            row_axpy(c, -(c->dll - c->dlu), s->rg);
            r->rdl = c->rdl + wv + c->dsl - c->dtl;
            r->rdu = c->rdu - wv + c->dsu - c->dtu;
            r->ml = c->ml + c->ll * c->dtl + c->tl * c->dll;
            r->mu_ = c->mu_ + c->lu * c->dtu + c->tu * c->dlu;
            nrm[2] = fmax(nrm[2], fmax(fabs(r->rdl), fabs(r->rdu)));
            nrm[3] = fmax(nrm[3], fmax(fabs(r->ml), fabs(r->mu_)));
            if (c->soft) {
                r->rsl = c->rsl + q->Zl[c->ks] * c->dsl - c->dll - c->dlsl;
                r->rsu = c->rsu + q->Zu[c->ks] * c->dsu - c->dlu - c->dlsu;
                r->rdsl = c->rdsl + c->dsl - c->dtsl;
                r->rdsu = c->rdsu + c->dsu - c->dtsu;
                r->msl = c->msl + c->lsl * c->dtsl + c->tsl * c->dlsl;
                r->msu = c->msu + c->lsu * c->dtsu + c->tsu * c->dlsu;
                nrm[0] = fmax(nrm[0], fmax(fabs(r->rsl), fabs(r->rsu)));
                nrm[2] = fmax(nrm[2], fmax(fabs(r->rdsl), fabs(r->rdsu)));
                nrm[3] = fmax(nrm[3], fmax(fabs(r->msl), fabs(r->msu)));
            }
        }
        for (i = (k == N ? nu : 0); i < nz; i++) {
            if (k == 0 && i >= nu) continue;
            nrm[0] = fmax(nrm[0], fabs(s->rg[i]));
        }
    }
    if (nrm[0] <= tol[0] && nrm[1] <= tol[1] && nrm[2] <= tol[2] && nrm[3] <= tol[3]) {
        memcpy(w->st, keep, sizeof(stage_t) * ((size_t)N + 1));
        free(keep);
        return 0;
    }
    for (i = 0; i < nx; i++) w->st[0].dz[nu + i] = 0.0; /* dx_0 of the correction: the x0 equation holds exactly */
    for (k = 0; k < N; k++) /* P_{k+1} b_k for the correction's b_k (riccati_factor made it for the step's) */
        for (i = 0; i < nx; i++) {
            double a = 0.0;
            for (j = 0; j < nx; j++) a += w->st[k + 1].P[i * nx + j] * w->st[k].rb[j];
            w->st[k].Pb[i] = a;
        }
    reduce_rows(w, 0);
    riccati_solve(w);
    expand_rows(w);
    for (k = 0; k <= N; k++) { /* step := step + correction, right-hand sides back */
        stage_t *s = &w->st[k];
        stage_t *o = &keep[k];
        for (i = 0; i < nz; i++) o->dz[i] += s->dz[i];
        for (i = 0; i < nx; i++) o->dpi[i] += s->dpi[i];
        for (i = 0; i < s->m; i++) {
            row_t *c = &o->r[i];
            const row_t *r = &s->r[i];
            c->dll += r->dll; c->dlu += r->dlu; c->dtl += r->dtl; c->dtu += r->dtu;
            c->dsl += r->dsl; c->dsu += r->dsu; c->dlsl += r->dlsl; c->dlsu += r->dlsu; c->dtsl += r->dtsl; c->dtsu += r->dtsu;
        }
    }
    memcpy(w->st, keep, sizeof(stage_t) * ((size_t)N + 1));
    free(keep);
    return 1;
}

int usv_qp_solve(const usv_qp *q, const usv_opts *o, usv_qp_sol *sol)
{
    const int N = q->N, nx = q->nx, nu = q->nu, nz = q->nz, K = q->K;
    ipm_ws w;
    int k, i, it, status = 1;
    w.q = q;
    if (!q->scratch) ((usv_qp *)q)->scratch = calloc((size_t)N + 1, sizeof(stage_t)); /* build_rows clears it */
    w.st = (stage_t *)q->scratch;
    build_rows(&w);
    init_cold(&w, o);
    residuals(&w);
    sol->cpc_fallbacks = 0;
    for (it = 0; it < o->qp_iter_max; it++) {
        double a_aff, a, mu_aff = 0.0, sigma;
        if (w.res[0] != w.res[0] || w.res[1] != w.res[1] || w.res[2] != w.res[2] || w.res[3] != w.res[3]) {
            status = 3;
            break;
        }
        if (w.res[0] <= o->tol_stat && w.res[1] <= o->tol_eq && w.res[2] <= o->tol_ineq &&
            w.res[3] <= o->tol_comp) {
            status = 0;
            break;
        }
        /* predictor */
        for (k = 0; k <= N; k++)
            for (i = 0; i < w.st[k].m; i++) {
                row_t *r = &w.st[k].r[i];
                r->ml = r->ll * r->tl; r->mu_ = r->lu * r->tu;
                r->msl = r->lsl * r->tsl; r->msu = r->lsu * r->tsu;
            }
        reduce_rows(&w, 1);
        if (riccati_factor(&w, o->riccati)) { status = 3; break; }
        riccati_solve(&w);
        expand_rows(&w);
        a_aff = step_length(&w);
        for (k = 0; k <= N; k++)
            for (i = 0; i < w.st[k].m; i++) {
                row_t *r = &w.st[k].r[i];
                mu_aff += (r->ll + a_aff * r->dll) * (r->tl + a_aff * r->dtl) +
                          (r->lu + a_aff * r->dlu) * (r->tu + a_aff * r->dtu);
                if (r->soft)
                    mu_aff += (r->lsl + a_aff * r->dlsl) * (r->tsl + a_aff * r->dtsl) +
                              (r->lsu + a_aff * r->dlsu) * (r->tsu + a_aff * r->dtsu);
            }
        if (w.nc) {
            mu_aff /= w.nc;
            sigma = mu_aff / w.mu;
            sigma = sigma * sigma * sigma;
            /* corrector */
            for (k = 0; k <= N; k++)
                for (i = 0; i < w.st[k].m; i++) {
                    row_t *r = &w.st[k].r[i];
                    r->ml = r->ll * r->tl + r->dll * r->dtl - sigma * w.mu;
                    r->mu_ = r->lu * r->tu + r->dlu * r->dtu - sigma * w.mu;
                    if (r->soft) {
                        r->msl = r->lsl * r->tsl + r->dlsl * r->dtsl - sigma * w.mu;
                        r->msu = r->lsu * r->tsu + r->dlsu * r->dtsu - sigma * w.mu;
                    }
                }
            reduce_rows(&w, 0);
            riccati_solve(&w);
            expand_rows(&w);
            if (o->cond_pred_corr) {
                /* the corrected step must not leave the central path further than the predictor promised: otherwise centring only */
                const double a_pc = step_length(&w);
                if (mu_after(&w, a_pc) > o->cpc_factor * mu_aff) {
                    for (k = 0; k <= N; k++)
                        for (i = 0; i < w.st[k].m; i++) {
                            row_t *r = &w.st[k].r[i];
                            r->ml = r->ll * r->tl - sigma * w.mu;
                            r->mu_ = r->lu * r->tu - sigma * w.mu;
                            if (r->soft) {
                                r->msl = r->lsl * r->tsl - sigma * w.mu;
                                r->msu = r->lsu * r->tsu - sigma * w.mu;
                            }
                        }
                    reduce_rows(&w, 0);
                    riccati_solve(&w);
                    expand_rows(&w);
                    sol->cpc_fallbacks++;
                }
            }
            /* iterative refinement of the direction that will be taken (HPIPM refines after the conditional block, as recalled) */
            if (o->itref_corr_max > 0) {
                const double tol[4] = {o->tol_stat, o->tol_eq, o->tol_ineq, o->tol_comp};
                int rr;
                for (rr = 0; rr < o->itref_corr_max; rr++)
                    if (!refine_step(&w, tol)) break;
            }
        }
        a = step_length(&w);
        if (a < o->alpha_min) { status = 2; break; }
        a = a * ((1.0 - a) * 0.99 + a * 0.9999999);
        for (k = 0; k <= N; k++) {
            stage_t *s = &w.st[k];
            for (i = 0; i < nz; i++) s->z[i] += a * s->dz[i];
            if (k >= 1)
                for (i = 0; i < nx; i++) s->pi[i] += a * s->dpi[i];
            for (i = 0; i < s->m; i++) {
                row_t *r = &s->r[i];
                r->ll += a * r->dll; r->lu += a * r->dlu; r->tl += a * r->dtl; r->tu += a * r->dtu;
                if (r->soft) {
                    r->sl += a * r->dsl; r->su += a * r->dsu;
                    r->lsl += a * r->dlsl; r->lsu += a * r->dlsu;
                    r->tsl += a * r->dtsl; r->tsu += a * r->dtsu;
                }
            }
        }
        residuals(&w);
    }
    if (it == o->qp_iter_max && status == 1) {
        /* final check after the last update */
        if (w.res[0] <= o->tol_stat && w.res[1] <= o->tol_eq && w.res[2] <= o->tol_ineq &&
            w.res[3] <= o->tol_comp)
            status = 0;
    }
    /* export */
    for (k = 0; k <= N; k++) {
        stage_t *s = &w.st[k];
        int m = 0;
        for (i = 0; i < nz; i++) sol->dz[k * nz + i] = s->z[i];
        for (i = 0; i < nx; i++) sol->pi[k * nx + i] = s->pi[i];
        if (k < N)
            for (i = 0; i < q->nbu; i++, m++) {
                sol->lam_bu[k * 2 * q->nbu + i] = s->r[m].ll; sol->lam_bu[k * 2 * q->nbu + q->nbu + i] = s->r[m].lu;
                sol->t_bu[k * 2 * q->nbu + i] = s->r[m].tl; sol->t_bu[k * 2 * q->nbu + q->nbu + i] = s->r[m].tu;
            }
        if (k >= 1 && k < N) {
            for (i = 0; i < q->nbx; i++, m++) {
                sol->lam_bx[k * 2 * q->nbx + i] = s->r[m].ll; sol->lam_bx[k * 2 * q->nbx + q->nbx + i] = s->r[m].lu;
                sol->t_bx[k * 2 * q->nbx + i] = s->r[m].tl; sol->t_bx[k * 2 * q->nbx + q->nbx + i] = s->r[m].tu;
                sol->sl_bx[k * q->nbx + i] = s->r[m].sl; sol->su_bx[k * q->nbx + i] = s->r[m].su;
                sol->lam_sbx[k * 2 * q->nbx + i] = s->r[m].lsl; sol->lam_sbx[k * 2 * q->nbx + q->nbx + i] = s->r[m].lsu;
                sol->t_sbx[k * 2 * q->nbx + i] = s->r[m].tsl; sol->t_sbx[k * 2 * q->nbx + q->nbx + i] = s->r[m].tsu;
            }
            for (i = 0; i < K; i++, m++) {
                sol->lam_g[k * 2 * K + i] = s->r[m].ll; sol->lam_g[k * 2 * K + K + i] = s->r[m].lu;
                sol->t_g[k * 2 * K + i] = s->r[m].tl; sol->t_g[k * 2 * K + K + i] = s->r[m].tu;
                sol->sl[k * K + i] = s->r[m].sl; sol->su[k * K + i] = s->r[m].su;
                sol->lam_s[k * 2 * K + i] = s->r[m].lsl; sol->lam_s[k * 2 * K + K + i] = s->r[m].lsu;
                sol->t_s[k * 2 * K + i] = s->r[m].tsl; sol->t_s[k * 2 * K + K + i] = s->r[m].tsu;
            }
        }
    }
    sol->iter = it;
    sol->status = status;
    for (i = 0; i < 4; i++) sol->res[i] = w.res[i];
    (void)nu;
    return status;
}

/* ------------------------------------------------------------------------------------------
 * 5b. Pieces of the full SQP (acados ocp_nlp_sqp + ocp_nlp_res_compute restated).
 * ---------------------------------------------------------------------------------------- */
static void load_rows(ipm_ws *w, const usv_qp_sol *sol)
{
    const usv_qp *q = w->q;
    const int N = q->N, K = q->K;
    int k, i;
    for (k = 0; k <= N; k++) {
        stage_t *s = &w->st[k];
        int m = 0;
        if (k < N)
            for (i = 0; i < q->nbu; i++, m++) {
                s->r[m].ll = sol->lam_bu[k * 2 * q->nbu + i]; s->r[m].lu = sol->lam_bu[k * 2 * q->nbu + q->nbu + i];
                s->r[m].tl = sol->t_bu[k * 2 * q->nbu + i]; s->r[m].tu = sol->t_bu[k * 2 * q->nbu + q->nbu + i];
            }
        if (k >= 1 && k < N) {
            for (i = 0; i < q->nbx; i++, m++) {
                s->r[m].ll = sol->lam_bx[k * 2 * q->nbx + i]; s->r[m].lu = sol->lam_bx[k * 2 * q->nbx + q->nbx + i];
                s->r[m].tl = sol->t_bx[k * 2 * q->nbx + i]; s->r[m].tu = sol->t_bx[k * 2 * q->nbx + q->nbx + i];
                s->r[m].sl = sol->sl_bx[k * q->nbx + i]; s->r[m].su = sol->su_bx[k * q->nbx + i];
                s->r[m].lsl = sol->lam_sbx[k * 2 * q->nbx + i]; s->r[m].lsu = sol->lam_sbx[k * 2 * q->nbx + q->nbx + i];
                s->r[m].tsl = sol->t_sbx[k * 2 * q->nbx + i]; s->r[m].tsu = sol->t_sbx[k * 2 * q->nbx + q->nbx + i];
            }
            for (i = 0; i < K; i++, m++) {
                s->r[m].ll = sol->lam_g[k * 2 * K + i]; s->r[m].lu = sol->lam_g[k * 2 * K + K + i];
                s->r[m].tl = sol->t_g[k * 2 * K + i]; s->r[m].tu = sol->t_g[k * 2 * K + K + i];
                s->r[m].sl = sol->sl[k * K + i]; s->r[m].su = sol->su[k * K + i];
                s->r[m].lsl = sol->lam_s[k * 2 * K + i]; s->r[m].lsu = sol->lam_s[k * 2 * K + K + i];
                s->r[m].tsl = sol->t_s[k * 2 * K + i]; s->r[m].tsu = sol->t_s[k * 2 * K + K + i];
            }
        }
    }
}

/* Dynamics multipliers by the adjoint recursion pi_k = (H z + g - C'(ll - lu))_x + A_k' pi_{k+1} at the QP
 * solution held in `sol` (what the device kernel carries instead of pi += alpha dpi; the two agree to the QP's
 * stationarity tolerance).  Overwrites sol->pi. */
void usv_qp_adjoint_pi(const usv_qp *q, usv_qp_sol *sol)
{
    const int N = q->N, nx = q->nx, nu = q->nu, nz = q->nz;
    ipm_ws w;
    int k, i, j;
    w.q = q;
    w.st = (stage_t *)calloc((size_t)N + 1, sizeof(stage_t));
    build_rows(&w);
    load_rows(&w, sol);
    for (k = N; k >= 1; k--) {
        stage_t *s = &w.st[k];
        const double *H = q->H + (size_t)k * nz * nz;
        const double *z = sol->dz + (size_t)k * nz;
        double t[NZM];
        for (i = 0; i < nz; i++) {
            double a = q->g[k * nz + i];
            for (j = 0; j < nz; j++) a += H[i * nz + j] * z[j];
            t[i] = a;
        }
        if (k < N) {
            const double *A = q->A + (size_t)k * nx * nx;
            for (j = 0; j < nx; j++)
                for (i = 0; i < nx; i++) t[nu + j] += A[i * nx + j] * sol->pi[(k + 1) * nx + i];
        }
        for (i = 0; i < s->m; i++) row_axpy(&s->r[i], -(s->r[i].ll - s->r[i].lu), t);
        for (i = 0; i < nx; i++) sol->pi[k * nx + i] = t[nu + i];
    }
    free(w.st);
}

/* NLP residuals (inf-norms: stationarity, dynamics, inequality, complementarity) of the iterate the QP `q`
 * was linearised at, with the multipliers / slacks of `sol` (the previous QP; NULL = all zero, first iteration). */
void usv_nlp_residuals(const usv_qp *q, const usv_qp_sol *sol, double *res)
{
    const int N = q->N, nx = q->nx;
    ipm_ws w;
    int k, i;
    w.q = q;
    w.st = (stage_t *)calloc((size_t)N + 1, sizeof(stage_t));
    build_rows(&w);
    if (sol) {
        load_rows(&w, sol);
        for (k = 1; k <= N; k++)
            for (i = 0; i < nx; i++) w.st[k].pi[i] = sol->pi[k * nx + i];
    }
    residuals(&w); /* z = 0: the QP residual at the origin IS the NLP residual of the linearisation point */
    for (i = 0; i < 4; i++) res[i] = w.res[i];
    free(w.st);
}

/* Full SQP on one instance: linearise, test the NLP residuals, solve the QP, full step; acados status
 * (0 converged, 2 max iter, 4 QP failure).  info[8] = {sqp_iter, last qp_status, res_stat, res_eq, res_ineq,
 * res_comp, total qp iterations, 0}. */
int usv_sqp(const usv_spec *s, double *x, double *u, const double *x0, const double *yref,
            const double *yref_e, const double *p, const double *lh, double *info)
{
    const int N = s->N, nx = s->nx, nu = s->nu, nz = nx + nu;
    usv_qp *q = usv_qp_alloc(s);
    usv_qp_sol *sol = usv_qp_sol_alloc(q);
    int k, i, it, status = 2, qs = 0, have = 0, qpit = 0;
    double res[4] = {0, 0, 0, 0};
    const int max_iter = s->nlp_max_iter > 0 ? s->nlp_max_iter : 100;
    for (it = 0; it < max_iter; it++) {
        usv_linearize(s, x, u, x0, yref, yref_e, p, lh, q);
        usv_nlp_residuals(q, have ? sol : NULL, res);
        if (res[0] <= s->nlp_tol[0] && res[1] <= s->nlp_tol[1] && res[2] <= s->nlp_tol[2] && res[3] <= s->nlp_tol[3]) {
            status = 0;
            break;
        }
        qs = usv_qp_solve(q, &s->opts, sol);
        qpit += sol->iter;
        if (!(qs == 0 || qs == 1)) { status = 4; break; }
        usv_qp_adjoint_pi(q, sol);
        have = 1;
        for (k = 0; k <= N; k++) {
            for (i = 0; i < nx; i++) x[k * nx + i] += sol->dz[k * nz + nu + i];
            if (k < N)
                for (i = 0; i < nu; i++) u[k * nu + i] += sol->dz[k * nz + i];
        }
    }
    if (info) {
        info[0] = it; info[1] = qs;
        for (i = 0; i < 4; i++) info[2 + i] = res[i];
        info[6] = qpit; info[7] = 0;
    }
    usv_qp_sol_free(sol);
    usv_qp_free(q);
    return status;
}

int usv_sqp_batch(const usv_spec *s, int B, double *x, double *u, const double *x0,
                  const double *yref, const double *yref_e, const double *p, const double *lh,
                  int *status, int *sqp_iter, double *res)
{
    const int N = s->N, nx = s->nx, nu = s->nu, K = s->K;
    int b, worst = 0;
    for (b = 0; b < B; b++) {
        double info[8];
        const int st = usv_sqp(s, x + (size_t)b * (N + 1) * nx, u + (size_t)b * N * nu, x0 + (size_t)b * nx,
                               yref + (size_t)b * N * s->ny, yref_e + (size_t)b * s->ny_e,
                               p + (size_t)b * (N + 1) * 2 * K, lh + (size_t)b * N * K, info);
        if (status) status[b] = st;
        if (sqp_iter) sqp_iter[b] = (int)info[0];
        if (res) { res[b * 4 + 0] = info[2]; res[b * 4 + 1] = info[3]; res[b * 4 + 2] = info[4]; res[b * 4 + 3] = info[5]; }
        if (st > worst) worst = st;
    }
    return worst;
}

/* ------------------------------------------------------------------------------------------
 * 6. SQP-RTI iteration (acados ocp_nlp_sqp_rti: preparation + feedback + full step).
 *    Caller protocol it serves: usv_guidance_ca1/main.py:111-175.
 * ---------------------------------------------------------------------------------------- */
static int rti_core(const usv_spec *s, usv_qp *q, usv_qp_sol *sol, double *x, double *u, const double *x0,
                    const double *yref, const double *yref_e, const double *p, const double *lh, double *sl,
                    double *su, double *pi, double *info)
{
    const int N = s->N, nx = s->nx, nu = s->nu, nz = nx + nu, K = s->K;
    int k, i, qs, status;
    usv_linearize(s, x, u, x0, yref, yref_e, p, lh, q);
    /* Stage-0 obstacle rows.  acados applies the nh rows at stages 0..N-1; at stage 0 they depend on no free variable
     * (D = 0 and x_0 is pinned to x0), so they are not rows of the QP solved here - but a HARD row that x0 violates
     * makes acados' QP infeasible: status 4 and an untouched iterate, reported here without iterating (qp_status 4). */
    if (K > 0 && !q->soft) {
        for (i = 0; i < K; i++) {
            const double v0 = q->Cxy[2 * i] * q->dx0[q->ipx] + q->Cxy[2 * i + 1] * q->dx0[q->ipy];
            if (q->lg[i] - v0 > s->opts.tol_ineq || v0 - q->ug[i] > s->opts.tol_ineq) {
                if (info) { info[0] = 0; info[1] = 4; for (k = 0; k < 4; k++) info[2 + k] = 0.0; info[6] = 0; info[7] = 0; }
                return 4;
            }
        }
    }
    qs = usv_qp_solve(q, &s->opts, sol);
    status = (qs == 0 || qs == 1) ? 0 : 4; /* max-iter tolerated in RTI */
    if (status == 0) {
        for (k = 0; k <= N; k++) {
            for (i = 0; i < nx; i++) x[k * nx + i] += sol->dz[k * nz + nu + i];
            if (k < N)
                for (i = 0; i < nu; i++) u[k * nu + i] += sol->dz[k * nz + i];
        }
    }
    for (k = 0; k < N; k++) {
        for (i = 0; i < K; i++) {
            if (sl) sl[k * K + i] = sol->sl[k * K + i];
            if (su) su[k * K + i] = sol->su[k * K + i];
        }
        if (k == 0 && q->soft) /* soft stage-0 rows: the slacks are constants, the minimisers of their own penalty */
            for (i = 0; i < K; i++) {
                const double v0 = q->Cxy[2 * i] * q->dx0[q->ipx] + q->Cxy[2 * i + 1] * q->dx0[q->ipy];
                double a = fmax(q->lsl[i], q->lg[i] - v0), c = fmax(q->lsu[i], v0 - q->ug[i]);
                if (q->Zl[i] > 0.0) a = fmax(a, -q->zl[i] / q->Zl[i]);
                if (q->Zu[i] > 0.0) c = fmax(c, -q->zu[i] / q->Zu[i]);
                if (sl) sl[i] = a;
                if (su) su[i] = c;
            }
        if (pi)
            for (i = 0; i < nx; i++) pi[k * nx + i] = sol->pi[(k + 1) * nx + i];
    }
    if (info) {
        info[0] = sol->iter; info[1] = qs;
        for (i = 0; i < 4; i++) info[2 + i] = sol->res[i];
        info[6] = 0; info[7] = 0;
    }
    return status;
}

int usv_rti(const usv_spec *s, double *x, double *u, const double *x0, const double *yref,
            const double *yref_e, const double *p, const double *lh, double *sl, double *su,
            double *pi, double *info)
{
    usv_qp *q = usv_qp_alloc(s);
    usv_qp_sol *sol = usv_qp_sol_alloc(q);
    const int status = rti_core(s, q, sol, x, u, x0, yref, yref_e, p, lh, sl, su, pi, info);
    usv_qp_sol_free(sol);
    usv_qp_free(q);
    return status;
}

int usv_rti_batch(const usv_spec *s, int B, double *x, double *u, const double *x0,
                  const double *yref, const double *yref_e, const double *p, const double *lh,
                  int *status, int *qp_iter)
{
    const int N = s->N, nx = s->nx, nu = s->nu, K = s->K;
    int b, worst = 0;
    for (b = 0; b < B; b++) {
        double info[8];
        const int st = usv_rti(s, x + (size_t)b * (N + 1) * nx, u + (size_t)b * N * nu, x0 + (size_t)b * nx,
                               yref + (size_t)b * N * s->ny, yref_e + (size_t)b * s->ny_e,
                               p + (size_t)b * (N + 1) * 2 * K, lh + (size_t)b * N * K, NULL, NULL, NULL, info);
        if (status) status[b] = st;
        if (qp_iter) qp_iter[b] = (int)info[0];
        if (st > worst) worst = st;
    }
    return worst;
}

/* All-core variant of the batch driver (one instance per thread, OpenMP dynamic schedule): the CPU
 * baseline bench.py times next to the GPU run.  nthreads <= 0: the OpenMP default. */
int usv_rti_batch_mt(const usv_spec *s, int B, double *x, double *u, const double *x0,
                     const double *yref, const double *yref_e, const double *p, const double *lh,
                     int *status, int *qp_iter, int nthreads)
{
    const int N = s->N, nx = s->nx, nu = s->nu, K = s->K;
    int b, worst = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel reduction(max : worst)
    {
        /* one QP + solution + IPM work space per thread, reused for all its instances: allocating them per
         * instance (1-2 MB each) serialises a many-core machine in the kernel's address-space lock */
        usv_qp *q = usv_qp_alloc(s);
        usv_qp_sol *sol = usv_qp_sol_alloc(q);
#pragma omp for schedule(dynamic, 4)
        for (b = 0; b < B; b++) {
            double info[8];
            const int st = rti_core(s, q, sol, x + (size_t)b * (N + 1) * nx, u + (size_t)b * N * nu, x0 + (size_t)b * nx,
                                    yref + (size_t)b * N * s->ny, yref_e + (size_t)b * s->ny_e,
                                    p + (size_t)b * (N + 1) * 2 * K, lh + (size_t)b * N * K, NULL, NULL, NULL, info);
            if (status) status[b] = st;
            if (qp_iter) qp_iter[b] = (int)info[0];
            if (st > worst) worst = st;
        }
        usv_qp_sol_free(sol);
        usv_qp_free(q);
    }
    return worst;
}
