/*
 * usv_oracle.h — CPU restatement (plain C, FP64) of the reference's SQP-RTI hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path (the HIP library under
 * mpc_collisionavoidance_amd/csrc, the Python host layer) may include, link or call this code;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker.
 *
 * PARITY UNPINNED: the arithmetic of the reference's path lives in the un-vendored `acados`
 * submodule (/root/reference/.gitmodules:1-3 — acados + BLASFEO + HPIPM, commit pin not
 * recoverable, API evidence brackets it to ~2020) and in git-ignored CasADi-generated C
 * (/root/reference/.gitignore:41-44).  The reference holds no tests and no golden vectors.  This
 * file therefore restates the *published* algorithm (acados SQP-RTI: ERK4 + forward VDE,
 * linear-least-squares Gauss-Newton, BGH constraints, HPIPM primal-dual IPM on a Riccati
 * recursion) for the OCPs that the reference defines exactly:
 *
 *   M0 `usv_model`              catkin_ws/src/nmpc_ca/scripts/usv_acados/usv_model.py:61-154
 *                               …/usv_acados/acados_settings.py:64-156
 *   M1 `usv_model_guidance_ca1` …/usv_guidance_ca1/usv_model.py:61-184, acados_settings.py:64-194
 *   M2 `usv_model_pf_ca`        …/usv_pf_ca/usv_model.py:61-213, acados_settings.py:64-176
 *
 * and it is pinned by independent checkers instead (sympy Jacobians, finite differences of the
 * RK4 map, a dense KKT check and an independent dense QP solve — tests/test_oracle_*.py).
 *
 * Conventions adopted where the reference is silent (acados/HPIPM defaults, from the published
 * algorithm; every one of them is a named field of usv_opts so it can be flipped):
 *   - ERK: classic RK4, 1 step per shooting interval, forward sensitivities by the VDE.
 *   - cost scaling: stage weight dt*W, terminal weight W_e (no `unscale`:
 *     usv_guidance_ca1/acados_settings.py:85-90 has it commented out).
 *   - x0 is imposed by elimination (lbx_0 = ubx_0 = x0: usv_guidance_ca1/main.py:111-112).
 *   - input bounds on stages 0..N-1, state bounds and h on stages 1..N-1, nothing at N
 *     (con_h_expr_e / lbx_e never set).  Stage-0 h rows depend on no free variable (D = 0).
 *   - soft h (M1): lh - sl <= h <= uh + su, sl >= lsh, su >= ush, stage cost dt*(zl*sl + zu*su +
 *     Zl/2 sl^2 + Zu/2 su^2); slacks are cold-started inside every QP.
 *   - QP: Mehrotra predictor-corrector IPM with HPIPM's conditional predictor-corrector, cold start
 *     (mu0 = 1, thr0 = 0.1), square-root backward Riccati with two rounds of iterative refinement,
 *     step-length damping alpha*((1-alpha)*0.99 + alpha*0.9999999), exit on inf-norm residuals
 *     (stat 1e-6, eq/ineq/comp 1e-8), iter_max 50, alpha_min 1e-8: HPIPM's BALANCE mode with
 *     acados' overwrites (usv_opts_profile; the other modes and the pre-round-6 defaults by name).
 *   - RTI: one linearisation + one QP + full step; QP max-iter is tolerated (status 0),
 *     NaN / min-step give status 4.
 */
#ifndef USV_ORACLE_H
#define USV_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define USV_NX_MAX 14
#define USV_NU_MAX 2
#define USV_NZ_MAX 16
#define USV_K_MAX 32
#define USV_NY_MAX 16

enum { USV_M0 = 0, USV_M1 = 1, USV_M2 = 2, USV_MGEN = 3 };

/* Hook for a model generated from a symbolic definition (mpc_collisionavoidance_amd/codegen.py,
 * emit_oracle_c): fjvp(x, U, s, su, f, js) = f and Jx s + Ju su.  Model id USV_MGEN then refers to it;
 * usv_spec_defaults(USV_MGEN) only fills the dimensions, the caller supplies weights and bounds. */
typedef void (*usv_fjvp_fn)(const double *, const double *, const double *, const double *, double *, double *);
void usv_oracle_register_generated(usv_fjvp_fn fn, int nx, int nu, int ipx, int ipy);
enum { USV_RICCATI_SQRT = 0, USV_RICCATI_CLASSIC = 1 };
/* QP solver profiles (usv_opts_profile; the device library has the same names: include/usvmpc.h USVMPC_HPIPM_*): HPIPM's modes with acados'
 * overwrites, and R04 = this restatement's defaults up to round 5 (HPIPM's SPEED values without acados' overwrites, no conditional
 * predictor-corrector, no refinement) */
enum { USV_HPIPM_BALANCE = 0, USV_HPIPM_SPEED = 1, USV_HPIPM_ROBUST = 2, USV_HPIPM_R04 = 3 };

/* (values in the comments: the default profile, USV_HPIPM_BALANCE) */
typedef struct usv_opts {
    int qp_iter_max;      /* 50 */
    double mu0;           /* 1 */
    double thr0;          /* 0.1 */
    double tol_stat;      /* 1e-6 */
    double tol_eq;        /* 1e-8 */
    double tol_ineq;      /* 1e-8 */
    double tol_comp;      /* 1e-8 */
    double alpha_min;     /* 1e-8 */
    int riccati;          /* USV_RICCATI_SQRT */
    int cond_pred_corr;   /* 1: conditional predictor-corrector - when the corrected step leaves the duality measure above
                           * cpc_factor x the predictor's mu_aff, the step is replaced by the centring-only one (no second-order term) */
    double cpc_factor;    /* 2.0 */
    int itref_corr_max;   /* 2.  > 0: that many rounds of iterative refinement of the corrector's KKT solve (residual of the linear
                           * system with the step just computed, solved again on the same factorisation, added), each skipped once the
                           * residual is below the exit tolerances */
} usv_opts;

/* One OCP definition, shared by every instance of a batch. Dense row-major matrices. */
typedef struct usv_spec {
    int model;            /* USV_M0 | USV_M1 | USV_M2 */
    int N;                /* shooting intervals */
    double dt;            /* Tf / N */
    int K;                /* circular obstacles: nh = K, np = 2K (M0: 0) */
    int nx, nu, ny, ny_e; /* filled by usv_spec_defaults */
    double W[USV_NY_MAX * USV_NY_MAX];
    double W_e[USV_NX_MAX * USV_NX_MAX];
    double Vx[USV_NY_MAX * USV_NX_MAX];
    double Vu[USV_NY_MAX * USV_NU_MAX];
    double Vx_e[USV_NX_MAX * USV_NX_MAX];
    int nbu; int idxbu[USV_NU_MAX]; double lbu[USV_NU_MAX], ubu[USV_NU_MAX];
    int nbx; int idxbx[USV_NX_MAX]; double lbx[USV_NX_MAX], ubx[USV_NX_MAX];
    double uh[USV_K_MAX];
    int soft;             /* 1: every h row is soft (idxsh = 0..K-1) */
    double lsh[USV_K_MAX], ush[USV_K_MAX];
    double zl[USV_K_MAX], zu[USV_K_MAX], Zl[USV_K_MAX], Zu[USV_K_MAX];
    usv_opts opts;
    /* soft state bounds (acados idxsbx / lsbx / usbx; S/race_cars/acados_settings_dev.py:107-127): flag per entry
     * of the bx list, lower bounds of its two slacks, and its slack penalties (acados orders the slack penalty
     * vectors zl.. as [sbx.., sh..]; they are kept apart here) */
    int sbx[USV_NX_MAX];
    double lsbx[USV_NX_MAX], usbx[USV_NX_MAX];
    double zl_bx[USV_NX_MAX], zu_bx[USV_NX_MAX], Zl_bx[USV_NX_MAX], Zu_bx[USV_NX_MAX];
    int sim_steps;        /* RK4 steps per shooting interval (sim_method_num_steps); 0 = 1 */
    int nlp_max_iter;     /* full SQP only (nlp_solver_max_iter); 0 = 100 */
    double nlp_tol[4];    /* full SQP only: stat, eq, ineq, comp (acados default 1e-6 each) */
} usv_spec;

/* Fill `s` with the reference's OCP definition for `model` (weights, selectors, bounds, soft
 * setup exactly as the cited acados_settings.py), generalised to K obstacles. */
int usv_spec_defaults(usv_spec *s, int model, int N, double Tf, int K);
void usv_opts_defaults(usv_opts *o);           /* = usv_opts_profile(o, USV_HPIPM_BALANCE) */
int usv_opts_profile(usv_opts *o, int mode);    /* USV_HPIPM_*; -1 for an unknown mode */

/* ---- model functions (pinned by sympy / known answers in tests) ---- */
int usv_model_dims(int model, int *nx, int *nu);
void usv_model_f(int model, const double *x, const double *u, double *f);
/* Jx: nx*nx row-major (d f_i / d x_j), Ju: nx*nu */
void usv_model_jac(int model, const double *x, const double *u, double *Jx, double *Ju);
/* position states used by the obstacle distance h_i = sqrt((px-ox_i)^2+(py-oy_i)^2) */
void usv_model_pos_idx(int model, int *ipx, int *ipy);
/* h[K], Cxy[K*2] = dh_i/d(px,py) */
void usv_model_h(int model, int K, const double *x, const double *p, double *h, double *Cxy);
/* one RK4 step with forward sensitivities: xn[nx], A[nx*nx], B[nx*nu] (row-major) */
void usv_rk4_sens(int model, double dt, const double *x, const double *u,
                  double *xn, double *A, double *B);

/* the same with `steps` RK4 steps of size dt/steps */
void usv_erk_sens(int model, double dt, int steps, const double *x, const double *u,
                  double *xn, double *A, double *B);

/* ---- dense OCP-QP of one RTI iteration (uniform stage dims, [u;x] ordering) ---- */
typedef struct usv_qp {
    int N, nx, nu, nz, K, nbu, nbx, soft;
    int idxbu[USV_NU_MAX], idxbx[USV_NX_MAX];
    int ipx, ipy;
    double *A, *B, *b;      /* [N][nx*nx], [N][nx*nu], [N][nx] */
    double *H, *g;          /* [N+1][nz*nz], [N+1][nz] (terminal: u rows/cols zero) */
    double *dx0;            /* [nx]  x0 - xbar_0 */
    double *lbu, *ubu;      /* [N][nbu]   relative to ubar */
    double *lbx, *ubx;      /* [N+1][nbx] relative to xbar; active for 1 <= k <= N-1 */
    double *Cxy;            /* [N+1][K*2] */
    double *lg, *ug;        /* [N+1][K]   relative to hbar; active for 1 <= k <= N-1 */
    double *zl, *zu, *Zl, *Zu, *lsl, *lsu; /* [K + nbx] soft data (h rows, then bx rows), already scaled by dt */
    int sbx[USV_NX_MAX];    /* which bx rows are soft */
    void *scratch;          /* IPM work space, allocated by the first usv_qp_solve on this QP and reused */
} usv_qp;

typedef struct usv_qp_sol {
    double *dz;             /* [N+1][nz] */
    double *pi;             /* [N+1][nx] (pi[0] unused) */
    double *lam_bu, *t_bu;  /* [N][2*nbu]  (lower | upper) */
    double *lam_bx, *t_bx;  /* [N+1][2*nbx] */
    double *lam_g, *t_g;    /* [N+1][2*K] */
    double *sl, *su;        /* [N+1][K] */
    double *lam_s, *t_s;    /* [N+1][2*K]  slack bound multipliers (sl | su) */
    double *sl_bx, *su_bx;  /* [N+1][nbx]  slacks of the soft state bounds */
    double *lam_sbx, *t_sbx; /* [N+1][2*nbx] */
    int iter, status;       /* status: 0 ok, 1 max iter, 2 min step, 3 nan */
    double res[4];          /* inf-norms: stat, eq, ineq, comp */
    int cpc_fallbacks;      /* usv_opts.cond_pred_corr: iterations whose corrected step was replaced by the centring-only one */
} usv_qp_sol;

usv_qp *usv_qp_alloc(const usv_spec *s);
void usv_qp_free(usv_qp *q);
usv_qp_sol *usv_qp_sol_alloc(const usv_qp *q);
void usv_qp_sol_free(usv_qp_sol *s);

/* Per-instance data, row-major: x[(N+1)*nx], u[N*nu], x0[nx], yref[N*ny], yref_e[ny_e],
 * p[(N+1)*2K], lh[N*K]. */
void usv_linearize(const usv_spec *s, const double *x, const double *u, const double *x0,
                   const double *yref, const double *yref_e, const double *p, const double *lh,
                   usv_qp *q);
int usv_qp_solve(const usv_qp *q, const usv_opts *o, usv_qp_sol *sol);

/* One SQP-RTI iteration in place on (x,u). Optional outputs may be NULL:
 * sl/su [N*K] (row k = stage k; stage 0 rows zero), pi [N*nx] (pi_1..pi_N), info[8] =
 * {qp_iter, qp_status, res_stat, res_eq, res_ineq, res_comp, 0, 0}. Returns acados-style status. */
int usv_rti(const usv_spec *s, double *x, double *u, const double *x0,
            const double *yref, const double *yref_e, const double *p, const double *lh,
            double *sl, double *su, double *pi, double *info);

/* ---- full SQP (nlp_solver_type = "SQP": S/race_cars/acados_settings_dev.py:157-164; commented knobs
 * S/usv_guidance_ca1/acados_settings.py:192-204) ---- */
void usv_qp_adjoint_pi(const usv_qp *q, usv_qp_sol *sol);
void usv_nlp_residuals(const usv_qp *q, const usv_qp_sol *sol, double *res);
int usv_sqp(const usv_spec *s, double *x, double *u, const double *x0,
            const double *yref, const double *yref_e, const double *p, const double *lh, double *info);
int usv_sqp_batch(const usv_spec *s, int B, double *x, double *u, const double *x0,
                  const double *yref, const double *yref_e, const double *p, const double *lh,
                  int *status, int *sqp_iter, double *res);

/* Batch driver used as the CPU baseline: instance b uses the b-th slice of every array. */
int usv_rti_batch(const usv_spec *s, int B, double *x, double *u, const double *x0,
                  const double *yref, const double *yref_e, const double *p, const double *lh,
                  int *status, int *qp_iter);

/* The same, instances distributed over nthreads OpenMP threads (<= 0: default). */
int usv_rti_batch_mt(const usv_spec *s, int B, double *x, double *u, const double *x0,
                     const double *yref, const double *yref_e, const double *p, const double *lh,
                     int *status, int *qp_iter, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
