/*
 * usvmpc.h — C ABI of the MI355X batched SQP-RTI solver (libusvmpc.so).
 *
 * Drop-in boundary: this is what a binding would call instead of the generated acados solver of
 * the reference.  Reference interfaces replaced (paths relative to /root/reference):
 *
 *   Python (acados_template.AcadosOcpSolver, used by the closed-loop scripts):
 *     AcadosOcpSolver(ocp, json_file=...)     catkin_ws/src/nmpc_ca/scripts/usv_guidance_ca1/acados_settings.py:207
 *                                             -> usvmpc_create
 *     solver.set(stage, "lbx"/"ubx", x0)      .../usv_guidance_ca1/main.py:111-112,174-175 -> usvmpc_set("x0")
 *     solver.set(stage, "yref", v)            .../usv_guidance_ca1/main.py:125,129         -> usvmpc_set("yref")
 *     solver.set(stage, "p", v)               .../usv_guidance_ca1/main.py:126,130         -> usvmpc_set("p")
 *     solver.constraints_set(stage, "lh", v)  .../usv_guidance_ca1/main.py:127             -> usvmpc_set("lh")
 *     solver.solve()                          .../usv_guidance_ca1/main.py:135             -> usvmpc_solve
 *     solver.get(stage, "x"|"u")              .../usv_guidance_ca1/main.py:147-148,169     -> usvmpc_get
 *   C (generated acados_solver_<model>.h + libacados, used by the ROS nodes):
 *     acados_create / acados_free             catkin_ws/src/nmpc_ca/src/nmpc_guidance_ca1.cpp:165,220 -> usvmpc_create / usvmpc_destroy
 *     ocp_nlp_constraints_model_set(lbx/ubx/lh) .../nmpc_guidance_ca1.cpp:515-516,571     -> usvmpc_set
 *     ocp_nlp_cost_model_set(yref)            .../nmpc_guidance_ca1.cpp:569,573            -> usvmpc_set
 *     acados_update_params(stage, p, np)      .../nmpc_guidance_ca1.cpp:570,574            -> usvmpc_set("p")
 *     acados_solve()                          .../nmpc_guidance_ca1.cpp:577                -> usvmpc_solve
 *     ocp_nlp_out_get(stage, "x"|"u")         .../nmpc_guidance_ca1.cpp:583,586            -> usvmpc_get
 *
 * One handle = one OCP definition x a batch of B independent instances on one HIP device.
 * All arrays are FP64, instance-major: element (b, stage, i) of a per-stage field of length n is
 * at [(b * n_stages + stage) * n + i].  The caller owns every host array; the library copies on
 * set and fills caller memory on get.  Return value 0 = ok; solver outcomes reuse acados'
 * status values (0 ok, 4 QP failure); negative values are API errors (usvmpc_last_error).
 * A handle is single-caller; distinct handles are independent.  Nothing here falls back to the
 * CPU: without a HIP device usvmpc_create fails.
 */
#ifndef USVMPC_H
#define USVMPC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define USVMPC_NX_MAX 14
#define USVMPC_NU_MAX 2
#define USVMPC_NY_MAX 16
#define USVMPC_K_MAX 32

enum { USVMPC_MODEL_USV = 0,             /* `usv_model`              nx 5  nu 2           */
       USVMPC_MODEL_GUIDANCE_CA1 = 1,    /* `usv_model_guidance_ca1` nx 8  nu 1, soft h   */
       USVMPC_MODEL_PF_CA = 2,           /* `usv_model_pf_ca`        nx 14 nu 2, hard h   */
       USVMPC_MODEL_GENERATED = 3 };     /* model compiled into THIS library from a symbolic definition
                                            (mpc_collisionavoidance_amd/codegen.py); only in libraries built for it */

/* The QP solver's argument profile.  The reference selects HPIPM through acados (qp_solver = "PARTIAL_CONDENSING_HPIPM":
 * catkin_ws/src/nmpc_ca/scripts/usv_pf_ca/acados_settings.py:172, usv_guidance_ca1/acados_settings.py:190) and leaves every knob at its default
 * (the tolerances are there, commented: :183-186 / :198-204), i.e. at what acados' ocp_qp_hpipm_opts_initialize_default produces: one of HPIPM's
 * modes (d_ocp_qp_ipm_arg_set_default) plus acados' own overwrites.  Neither source tree is in /root/reference; DESIGN.md section 2 lists every
 * field as recalled.  The modes differ, for this library, in nothing the kernels do (iterative refinement and the LQ fall-back are the checker's
 * business: oracle/usv_oracle.h); they are kept apart so that a binding can pass acados' `hpipm_mode` through:
 *   BALANCE / SPEED / ROBUST  cond_pred_corr = 1 (HPIPM, every mode), mu0 = 1, alpha_min = 1e-8, tolerances 1e-6 / 1e-8 / 1e-8 / 1e-8,
 *                             iter_max = 50 (acados' overwrites of the mode's values)
 *   R04                       this library's behaviour up to its round 5: cond_pred_corr = 0, mu0 = 10, alpha_min = 1e-12 (HPIPM's own
 *                             mode values without acados' overwrites) - kept reachable for comparison */
enum { USVMPC_HPIPM_BALANCE = 0, USVMPC_HPIPM_SPEED = 1, USVMPC_HPIPM_ROBUST = 2, USVMPC_HPIPM_R04 = 3 };

enum { USVMPC_E_ARG = -1, USVMPC_E_FIELD = -2, USVMPC_E_STAGE = -3, USVMPC_E_SIZE = -4,
       USVMPC_E_HIP = -5, USVMPC_E_NODEVICE = -6 };

/* The OCP definition (what AcadosOcp carries).  Dense row-major matrices with the model's
 * actual dimensions: W ny x ny, W_e nx x nx, Vx ny x nx, Vu ny x nu, Vx_e nx x nx. */
typedef struct usvmpc_desc {
    int model;
    int N;                 /* shooting intervals, >= 2 (the reference's OCPs: 20 .. 100); usvmpc_create refuses less */
    double Tf;
    int K;                 /* circular obstacles: nh = K, np = 2K */
    int batch;
    int device;            /* HIP device ordinal */
    double W[USVMPC_NY_MAX * USVMPC_NY_MAX];
    double W_e[USVMPC_NX_MAX * USVMPC_NX_MAX];
    double Vx[USVMPC_NY_MAX * USVMPC_NX_MAX];
    double Vu[USVMPC_NY_MAX * USVMPC_NU_MAX];
    double Vx_e[USVMPC_NX_MAX * USVMPC_NX_MAX];
    int nbu; int idxbu[USVMPC_NU_MAX]; double lbu[USVMPC_NU_MAX], ubu[USVMPC_NU_MAX];
    int nbx; int idxbx[USVMPC_NX_MAX]; double lbx[USVMPC_NX_MAX], ubx[USVMPC_NX_MAX];
    double uh[USVMPC_K_MAX];
    int soft;              /* 1: all h rows soft (idxsh = 0..K-1) */
    double lsh[USVMPC_K_MAX], ush[USVMPC_K_MAX];
    double zl[USVMPC_K_MAX], zu[USVMPC_K_MAX], Zl[USVMPC_K_MAX], Zu[USVMPC_K_MAX];
    int qp_iter_max;
    double mu0, thr0, tol_stat, tol_eq, tol_ineq, tol_comp, alpha_min;
    /* soft state bounds (acados idxsbx / lsbx / usbx, scripts/race_cars/acados_settings_dev.py:107-127), per entry of
     * the bx list: flag, lower bounds of the two slacks, slack penalties (acados keeps the penalties of all slacks in
     * one vector ordered [sbx.., sh..]; zl.. above are the sh part, these the sbx part) */
    int sbx[USVMPC_NX_MAX];
    double lsbx[USVMPC_NX_MAX], usbx[USVMPC_NX_MAX];
    double zl_bx[USVMPC_NX_MAX], zu_bx[USVMPC_NX_MAX], Zl_bx[USVMPC_NX_MAX], Zu_bx[USVMPC_NX_MAX];
    /* integrator: RK4 steps per shooting interval (acados sim_method_num_steps; 0 is read as 1) */
    int sim_num_steps;
    /* full SQP only (usvmpc_solve_sqp): nlp_solver_max_iter (0 is read as 100) and the exit tolerances on
     * the NLP residuals stat / eq / ineq / comp (0 is read as 1e-6, the acados default) */
    int nlp_max_iter;
    double nlp_tol_stat, nlp_tol_eq, nlp_tol_ineq, nlp_tol_comp;
    /* QP solver profile (USVMPC_HPIPM_*; usvmpc_hpipm_profile fills the fields it governs: mu0, alpha_min, the tolerances, qp_iter_max and the
     * two below) and HPIPM's conditional predictor-corrector: 1 = a corrected step that leaves the duality measure above cpc_factor (0 is read
     * as 2) x the predictor's mu_aff is redone with the centring-only step */
    int hpipm_mode;
    int cond_pred_corr;
    double cpc_factor;
} usvmpc_desc;

typedef struct usvmpc_handle usvmpc_handle;

/* nx, nu of a model id; returns 0 or USVMPC_E_ARG */
int usvmpc_model_dims(int model, int *nx, int *nu);
/* solver option defaults: thr0 0.1, one RK4 step per interval, the NLP tolerances, and the QP solver profile USVMPC_HPIPM_BALANCE
 * (iter_max 50, mu0 1, tolerances 1e-6/1e-8, alpha_min 1e-8, cond_pred_corr 1) */
void usvmpc_default_options(usvmpc_desc *d);
/* the fields of d that the QP solver profile `mode` governs (above); returns 0 or USVMPC_E_ARG */
int usvmpc_hpipm_profile(usvmpc_desc *d, int mode);

int usvmpc_create(const usvmpc_desc *d, usvmpc_handle **out);
int usvmpc_destroy(usvmpc_handle *h);

/* fields: "x0" (n = nx), "yref" (stage 0..N-1: n = ny; stage N: n = nx), "p" (stage 0..N,
 * n = 2K), "lh" (stage 0..N-1, n = K), "x" (stage 0..N, n = nx), "u" (stage 0..N-1, n = nu).
 * v holds [batch][n] for one stage, or, with stage = -1, [batch][n_stages][n] for all stages
 * ("yref" with stage -1 covers stages 0..N-1 only). */
int usvmpc_set(usvmpc_handle *h, const char *field, int stage, const double *v, size_t n);
/* fields: "x", "u", "pi" (stage 1..N), "sl", "su" (stage 0..N-1, n = K), "res" (stage ignored,
 * n = 4: QP residuals stat/eq/ineq/comp), "nlp_res" (the same for the NLP, full SQP only); same stage = -1
 * convention. */
int usvmpc_get(usvmpc_handle *h, const char *field, int stage, double *out, size_t n);
/* "lam" / "t" (stage 0..N, n = 2 (nrow + ns)): the inequality multipliers and slacks of the last QP, what acados'
 * ocp_nlp_out_get(.., stage, "lam" | "t") returns: rows of a stage in acados' order [bu.., bx.., h..] (nrow = nbu + nbx + K),
 * slack rows [sbx.., sh..] (ns = soft state bounds + K when the h rows are soft); the vector is
 * [lower (nrow) | upper (nrow) | lower-slack bound (ns) | upper-slack bound (ns)].  Rows a stage does not have (state bounds
 * and h rows at stage 0 - x0 is eliminated -, everything at stage N) read 0.  Gathered from the solver workspace by a kernel
 * of its own on the first such get after a solve. */
/* further get fields: "obs_tmin" (stage ignored, n = 1): the smallest lower-side slack t_l over the instance's obstacle rows
 * in the last QP (1e300 without rows) - below ~1e-3 the solution touches a keep-out circle, i.e. an obstacle row is active */
/* integer per-instance results: "status" (0 | 4; after usvmpc_solve_sqp 0 | 2 | 4), "qp_status" (0 ok,
 * 1 max iter, 2 min step, 3 nan, 4 x0 violates a hard obstacle row of stage 0 - the QP has no feasible point), "qp_iter", "sqp_iter" */
int usvmpc_get_int(usvmpc_handle *h, const char *field, int *out);

/* One SQP-RTI iteration for every instance; returns the worst status. status may be NULL. */
int usvmpc_solve(usvmpc_handle *h, int *status);
/* Full SQP (acados nlp_solver_type "SQP": the option the reference's settings files mention but leave
 * commented, scripts/usv_guidance_ca1/acados_settings.py:192-204; used by scripts/race_cars/
 * acados_settings_dev.py:157): per instance, linearise - test the NLP residuals - solve the QP - full step,
 * until the residuals are below the tolerances (status 0), nlp_max_iter QPs have been solved (2) or a QP
 * fails (4).  Converged instances are frozen while the rest of the batch continues.  The multipliers of
 * the last QP are kept between calls, so a call from a converged point returns after 0 iterations.  Afterwards
 * usvmpc_get_int "sqp_iter" and usvmpc_get "nlp_res" (n = 4) describe the run. */
int usvmpc_solve_sqp(usvmpc_handle *h, int *status);
/* Enqueue one RTI iteration on the handle's stream without synchronising or reading back */
int usvmpc_solve_async(usvmpc_handle *h);
int usvmpc_sync(usvmpc_handle *h);

/* device pointer of a field ("x","u","x0","yref","yref_e","p","lh","pi","sl","su","status",
 * "qp_iter","res") for zero-copy use from torch / RCCL */
int usvmpc_get_device_ptr(usvmpc_handle *h, const char *field, void **dptr);
/* HIP-event durations (ms) of the two kernels of the most recent solve / of the last n solves
 * (oldest first, n <= 64), measured on the stream the kernels were launched on */
int usvmpc_last_kernel_ms(usvmpc_handle *h, float *linearize_ms, float *qp_ms);
int usvmpc_kernel_ms(usvmpc_handle *h, int n, float *linearize_ms, float *qp_ms);
/* tick-to-tick times (ms) over the last n solves: ms[i] = start of solve i + 1 minus start of solve i on the handle's stream (n - 1 values,
 * oldest first, 2 <= n <= 64) - everything a closed-loop step enqueued in between (QP launch, hand-over, the caller's own kernels) included;
 * what bench.py reports the median of ("report median") */
int usvmpc_tick_ms(usvmpc_handle *h, int n, float *ms);
/* number of instances whose solve ended with status != 0, for each of the last n solves (oldest first, n <= 64); counted on
 * the device by the solve itself, so a closed loop can be audited without a read-back per tick */
int usvmpc_fail_counts(usvmpc_handle *h, int n, int *counts);
/* ... and the number whose QP did not converge to the IPM tolerances (qp_status != 0: iteration cap, step-length floor, NaN, x0 inside
 * a hard keep-out circle), per solve likewise: solves minus this = "solves" as SURVEY.md 8(d) counts them (IPM converged to the stated
 * tolerance).  RTI solves (usvmpc_solve / usvmpc_solve_async); counted by a small kernel behind the QP launch. */
int usvmpc_unconverged_counts(usvmpc_handle *h, int n, int *counts);
/* ... and their sum over every RTI solve since the handle was created (a device-side running sum: no limit of 64 solves) */
int usvmpc_unconverged_total(usvmpc_handle *h, long long *total);
/* option "handover_iter": how many instances each of the last n RTI launches handed over to its follow-up launch (oldest first, n <= 64) */
int usvmpc_handover_counts(usvmpc_handle *h, int n, int *counts);
/* ... and how long that follow-up launch (kernel usv_qp_resume) took, in ms, for each of the last n solves - part of usvmpc_kernel_ms' qp_ms;
 * 0 for a solve without one */
int usvmpc_followup_ms(usvmpc_handle *h, int n, float *ms);
/* option "handover_co": of those, how many the follow-up kernel running BESIDE the launch (usv_qp_resume_co) finished, and how many of its
 * waits for a list entry ran into the spin limit (left to the follow-up launch behind the main one; timeouts may be NULL) */
int usvmpc_handover_co_counts(usvmpc_handle *h, int n, int *finished, int *timeouts);
/* option "pipeline_linearize": how many linearisations made ahead of time (on the second stream, beside the previous tick's QP launch)
 * were used by the following solve / discarded because the caller wrote x, u or yref in between.  The lineariser only runs ahead after
 * two solves in a row without such a write, so a caller that sets yref every tick (the reference's protocol) discards none. */
int usvmpc_pipeline_stats(usvmpc_handle *h, long *used, long *discarded);
/* Which mapping the last solve (RTI, or the last launch of a full SQP) ran on: 0 = four instances per wavefront (one per 16-lane row: the throughput mapping), 1 = ONE
 * instance per wavefront (option "wide": the latency mapping north_star names - the four rows of the wave share out the stage-local
 * constraint-row work of four consecutive stages), 4 = one instance per workgroup of FOUR wavefronts (option "wide_waves": a whole CU
 * shares out the row work of 16 consecutive stages; default for soft-row OCPs in batches of at most one instance per CU).  The latency
 * mapping is taken by default for batches that leave SIMDs idle (RTI solves and the launches of a full SQP; every layout with a diagonal
 * Hessian - i.e. every OCP of the reference: up to 32 obstacle rows, soft state bounds); the planes live in the CU's LDS when the horizon
 * fits, else - and for a full SQP - in HBM.
 * WHICH mapping ran does not show in the results: all of them take every sum in the same order and contract multiply-adds the same way
 * (qp_ipm.hpp: #pragma clang fp contract(on)), so statuses, iteration counts, iterates and multipliers are the same BITS - an instance
 * solved in a handle of 1, of 512 or of 65 536 returns the same answer (tests/test_gpu_wide.py, test_gpu_closed_loop.py::
 * test_shards_that_land_on_the_other_mapping_equal_the_unsharded_batch).  The reference solves one instance per call:
 * nmpc_guidance_ca1.cpp:577,612, usv_pf_ca/main.py:142-186. */
int usvmpc_last_mapping(usvmpc_handle *h, int *mapping);
/* Closed-loop hand-over on the device: x0 <- x_1 (+ sigma * N(0,1) on the states selected by option
 * "disturbance_mask", default all), enqueued on the stream.
 * Replaces x0 = solver.get(1,"x"); solver.set(0,"lbx",x0); solver.set(0,"ubx",x0)
 * (catkin_ws/src/nmpc_ca/scripts/usv_guidance_ca1/main.py:169-175). */
int usvmpc_advance(usvmpc_handle *h, double sigma, unsigned long long seed);
/* Adopt a caller-owned HIP stream (e.g. torch's current stream) for all subsequent work */
int usvmpc_set_stream(usvmpc_handle *h, void *stream);
/* run-time options (scheduling / placement only: results are the same bits whatever they are set to - except "merge_box_rows", which
 * changes rounding, and "qp_cond_N", which selects another formulation of the same QP):
 *   "qp_cond_N" (default 0) - acados' qp_solver_cond_N (qp_solver = PARTIAL_CONDENSING_HPIPM:
 *       catkin_ws/src/nmpc_ca/scripts/usv_pf_ca/acados_settings.py:172; the reference never sets it, i.e. blocks of one stage = the
 *       default here).  A value N2 < N makes every RTI solve condense its QP to N2 dense stages first (HPIPM d_part_cond_qp:
 *       states of N / N2 consecutive stages - one more in the first N mod N2 blocks - eliminated through the dynamics;
 *       nx + ceil(N / N2) nu <= 64), solve THAT QP with the same
 *       interior-point method and expand the solution (x, u, pi) - a kernel of its own (csrc/cond_ipm.hpp: one instance per
 *       workgroup, the block's matrices in LDS).  Obstacle rows hard or soft; E_ARG for soft state bounds; "lam" / "t" only when
 *       their buffers exist before the solve (option "keep_multipliers"); usvmpc_solve_sqp keeps solving the uncondensed stages.  Same solution as the default path up to the IPM exit
 *       tolerances; which of the two is faster is measured in DESIGN.md section 6.  0 or N: off;
 *   "sort_by_difficulty" (default 1) - instances are handed to the wavefront rows in the order of their IPM iteration counts of
 *       the previous solve, hardest first (an instance that runs long must not start late); "sort_two_ticks" = 1 (default 0)
 *       orders by the larger of the last two solves' counts instead;
 *   "static_obstacles" (default 0) - every stage uses stage 0's p and lh (what the reference's callers set:
 *       scripts/usv_pf_ca/main.py puts one obstacle set on all stages), which the kernel then keeps in registers;
 *   "pack_box_rows" (default 1 when the rows fit) - box-row multipliers share the obstacle rows' planes;
 *   "dynamic_rows" (default 1) - an RTI solve is one persistent launch whose wavefront rows pull instances from a device
 *       queue as their QPs converge (0: every row keeps its first instance and idles until its wave is done);
 *   "lds_workspace" (default -1) - per-stage planes of the QP in the CU's LDS instead of HBM: -1 when one round of
 *       workgroups covers the batch (an instance's whole horizon must fit in 160 KB), 0 never, 1 whenever it fits.  The same
 *       bits as the HBM placement (a separately compiled instantiation of the same code, contracted the same way);
 *   "merge_box_rows" (default 1) - when every box row rides in an idle lane of the last obstacle chunk's planes the sweeps
 *       process them there (one row pass instead of two).  THE option that changes rounding: a row's share of the complementarity
 *       sums is added in another lane, so statuses and iteration counts are equal and iterates agree to rounding, not to the bit;
 *       a handle keeps one setting for its lifetime unless the caller changes it;
 *   "wide_waves" (default -1) - wavefronts per instance of the latency mapping: -1 four for OCPs with soft obstacle rows or two obstacle
 *       chunks (K > 16) while the batch is at most one instance per CU - two from N = 40 - (their row work is the larger share: 6 - 25 % per
 *       tick), one otherwise; 1; 4 (built for the packed row layouts and the one without obstacle rows; the others stay at one);
 *   "wide" (default -1) - the latency mapping, ONE instance per wavefront (usvmpc_last_mapping): -1 while the batch fits the device's
 *       SIMDs twice over, 0 never, 1 whenever the OCP's layout allows it; results do not change by a bit;
 *   "aux_in_lds" (default 1) - an RTI solve keeps the per-stage aux plane (dense box rows, linearisation point, r_g, l_u) in the
 *       wavefronts' LDS instead of streaming it, when the horizon fits without costing a resident wavefront (a separately
 *       compiled instantiation of the same arithmetic: the same bits);
 *   "pipeline_linearize" (default 1; RTI solves of handles with >= 16384 instances) - the lineariser of the NEXT tick is enqueued on a
 *       second stream behind the QP launch and runs in that launch's tail, instance by instance as results become final (the
 *       few it has to skip are redone in front of the next QP launch); any usvmpc_set, option change or device-pointer access in
 *       between makes the next solve linearise afresh.  The queue order of a tick is then made one tick earlier.  Scheduling
 *       only: results are bit-identical to the un-pipelined sequence;
 *   "host_mirror" (default: on for handles whose caller-visible arrays total <= 1 MiB, i.e. the single-instance drop-in faces) -
 *       usvmpc_set writes a pinned host mirror and the next solve uploads the dirty fields in one asynchronous copy instead
 *       (ordering: a set becomes visible on the device with the NEXT launch of the handle, not at the call - except once a device pointer
 *       has been handed out (usvmpc_get_device_ptr): from then on stage-wise sets are enqueued on the handle's stream at the call, so that a
 *       caller's own kernels on that stream see them in program order)
 *       of one synchronising copy per call (the reference issues 3N+4 setters per tick: scripts/usv_guidance_ca1/main.py:
 *       123-130, src/nmpc_guidance_ca1.cpp:567-574); x / u / status come back in one copy and usvmpc_get "x" / "u" is served
 *       from it.  0 switches it off for the handle (it cannot be switched on again);
 *   "hpipm_mode" (USVMPC_HPIPM_*; default: the descriptor's) - re-applies a QP solver profile to the handle: mu0, alpha_min and cond_pred_corr
 *       as the profile says (tolerances and iter_max are the same in every profile and keep the descriptor's values);
 *   "cond_pred_corr" (default: the descriptor's, 1 in every profile but R04), "cpc_factor" (default 2) - HPIPM's conditional predictor-corrector,
 *       an option of the QP solver the reference selects (qp_solver = PARTIAL_CONDENSING_HPIPM: catkin_ws/src/nmpc_ca/scripts/usv_pf_ca/
 *       acados_settings.py:172) that every mode acados can pick switches on (DESIGN.md section 2 lists every HPIPM argument, adopted or not): an
 *       IPM iteration whose corrected step leaves the duality measure above cpc_factor x the predictor's is redone with the centring-only step.
 *       THE option that changes results beyond rounding (another iteration path to the same tolerance).  The test is built into every kernel -
 *       every mapping, the follow-up launch of the hand-over, the launches of a full SQP, the partially condensed solve; switched off, the same
 *       kernels return the bits of kernels without it;
 *   "handover_iter" (default -1) - RTI launches on the throughput mapping: once every instance of the launch has been handed out, a row
 *       whose instance has passed this many IPM iterations leaves it to a follow-up launch on the latency mapping (kernel usv_qp_resume: one
 *       instance per wavefront, the suspended solve's planes copied into LDS first), which finishes it at half the time per iteration of a lone
 *       16-lane row - a launch ends with its 30 - 50 iteration instances on an otherwise idle device.  Scheduling only: the mappings return the
 *       same bits.  -1: past 20 iterations where the horizon's planes fit a CU's LDS AND the batch is at most three times what the device holds at
 *       once (re-measured under the default QP solver profile, whose solves are shorter: with "handover_co" -18 % per tick at 4 096 instances,
 *       -13 % at 8 192, -4 % at 16 384, nothing from 32 768 up), never otherwise; 0: never; n > 0: past n, whatever the batch.  Layouts: one
 *       obstacle chunk / no obstacle rows, packed box rows, no soft state bounds;
 *   "handover_lds" (default 1) - 0: the follow-up launch works over the planes in HBM whatever the horizon (measured: a loss);
 *   "handover_co" (default -1 = on where the follow-up works in LDS and the handle owns its stream; 0: off) - the follow-up kernel also runs
 *       BESIDE the draining launch (kernel usv_qp_resume_co on a stream of its own): its workgroups come onto the device as wavefronts of the
 *       main launch leave, wait - a bounded wait, "handover_co_spin" polls - for list entries to appear and finish those instances while the
 *       main launch is still draining; what they do not get to is done by the launch behind it (every entry is taken by exactly one of the
 *       two).  Scheduling only.  "handover_co_wgs": workgroups of that kernel (0 = one per CU: what finds room beside the main launch's workgroups at once);
 *   "disturbance_mask" (default all ones) - bit j set: usvmpc_advance adds its noise to state j (the reference's commented
 *       hooks disturb x0[3] and x0[5] only: catkin_ws/src/nmpc_ca/scripts/usv_pf_ca/main.py:181-183). */
int usvmpc_set_option(usvmpc_handle *h, const char *name, double value);
/* ---- Guidance front end (model usv_model_guidance_ca1 only): the arithmetic either side of the solver
 * call in the reference's ROS node, batched on the device (class NMPC in
 * catkin_ws/src/nmpc_ca/src/nmpc_guidance_ca1.cpp).  Per-instance state (waypoint index k, past_psied) lives in
 * the handle.  Arrays are host pointers, instance-major.
 *   reset   = main(), new waypoint list                        :616-632  (waypoints [B][2*npts], psi [B])
 *   prepare = velocityCallback / obstaclesCallback / body2NED / waypoint_manager / control (input part)
 *                                                               :223-230,252-376,441-574
 *             vel_uv [B][2], pose [B][3] = (nedx, nedy, psi), obstacles [B][lmax][3] = body (x, y, R),
 *             n_obstacles [B], lmax <= 64; writes x0 and the (stage-independent) p / lh of the solver and switches
 *             the solver to static obstacles; asynchronous on the handle's stream
 *   publish = control (output part)                             :583-600  desired heading / r / speed, ye
 * The "static_obstacles" option (usvmpc_set_option) makes every stage use stage 0's p and lh. */
int usvmpc_guidance_reset(usvmpc_handle *h, const double *waypoints, int npts, const double *psi);
int usvmpc_guidance_prepare(usvmpc_handle *h, const double *vel_uv, const double *pose, const double *obstacles,
                            const int *n_obstacles, int lmax);
/* The obstacle simulator's simulate() (catkin_ws/src/simulation/scripts/obstacle_sim_node.py:56-81,101-117):
 * world[batch][n_world][3] = (X, Y, R) in NED, pose[batch][3] = (ned_x, ned_y, yaw); every obstacle closer than
 * max_radius (the node: 100) is reported in the body frame, in list order, at most 64 per instance.  The lists
 * stay on the device as the input of the next usvmpc_guidance_prepare when that is called with
 * n_obstacles == NULL; obstacles [batch][64][3] / n_obstacles [batch] (both optional) receive a copy. */
int usvmpc_guidance_sense(usvmpc_handle *h, const double *pose, const double *world, int n_world, double max_radius,
                          double *obstacles, int *n_obstacles);
int usvmpc_guidance_publish(usvmpc_handle *h, double *heading, double *r_des, double *speed, double *ye, int *active);
int usvmpc_guidance_state(usvmpc_handle *h, int *wp_index, float *past_psied);
/* Profiling aid: stream `nplanes` workspace planes with the solver kernels' access instruction
 * (kernel usv_calib_stream) and report the exact byte counts, to calibrate HBM PMC counters.
 * Overwrites solver scratch; the next usvmpc_solve re-initialises it. */
int usvmpc_calibrate_traffic(usvmpc_handle *h, int nplanes, double *bytes_read, double *bytes_written);
/* Test entry points (no handle): the device transcription of the reference's model files evaluated on caller-supplied
 * points - f [n][nx] and the Jacobian J [n][nx][nu+nx] with respect to z = [u; x] exactly as the lineariser obtains them
 * (one tangent column per call of the model's fjvp), for the CasADi expressions of
 * catkin_ws/src/nmpc_ca/scripts/{usv_acados,usv_guidance_ca1,usv_pf_ca}/usv_model.py; and the obstacle row
 * h = sqrt((px-ox)^2 + (py-oy)^2) with its position gradient exactly as the QP kernel evaluates it:
 * pos [n][2], p [n][2K] -> h [n][K], grad [n][K][2]. */
int usvmpc_debug_model_eval(int model, int device, int n, const double *x, const double *u, double *f, double *J);
int usvmpc_debug_obstacle_eval(int device, int n, int K, const double *pos, const double *p, double *h, double *grad);
/* bytes of device memory held by the handle */
size_t usvmpc_device_bytes(usvmpc_handle *h);
const char *usvmpc_last_error(usvmpc_handle *h);

#ifdef __cplusplus
}
#endif
#endif
