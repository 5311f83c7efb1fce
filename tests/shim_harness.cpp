// Node-like caller for the acados shim (tests/test_shim.py): the solver-facing statements of
// NMPC::NMPC / NMPC::control in /root/reference/catkin_ws/src/nmpc_ca/src/nmpc_guidance_ca1.cpp
// (:44-52 globals, :165 create, :515-516, :567-574 setters, :577 solve, :583-586 getters, :220 free),
// run for a few closed-loop ticks; prints u0 and x1 of every tick.
#include "acados_c/ocp_nlp_interface.h"
#include "acados_solver_usv_model_guidance_ca1.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

extern "C" int usvmpc_shim_last_kernel_ms(float *linearize_ms, float *qp_ms);

ocp_nlp_in *nlp_in;
ocp_nlp_out *nlp_out;
ocp_nlp_solver *nlp_solver;
void *nlp_opts;
ocp_nlp_plan *nlp_solver_plan;
ocp_nlp_config *nlp_config;
ocp_nlp_dims *nlp_dims;

#define N 100
#define NX 8
#define NU 1
#define NY 9
#define NYN 8

int main(int argc, char **argv)
{
    const int ticks = argc > 1 ? std::atoi(argv[1]) : 3;
    if (acados_create()) { std::printf("acados_create failed\n"); return 1; }
    double x0[NX] = {0.7, 0.0, 4.0, -1.5707963267948966, -1.5707963267948966, 0.0, 0.0, 0.0};
    double yref[NY] = {0}, yref_e[NYN] = {0};
    double p_obs[16], r_obs[8];
    for (int i = 0; i < 8; i++) { p_obs[2 * i] = 100; p_obs[2 * i + 1] = 100; r_obs[i] = 0; }
    const double ob[4][2] = {{4, 4}, {4, 7}, {4, 12}, {4, 20}};
    for (int i = 0; i < 4; i++) { p_obs[2 * i] = ob[i][0]; p_obs[2 * i + 1] = ob[i][1]; r_obs[i] = 1.5; }
    double wall_us = 0.0, kern_us = 0.0; // per tick: setters + solve + getters, and the two kernels inside it (first tick left out)
    for (int t = 0; t < ticks; t++) {
        const auto t0 = std::chrono::steady_clock::now();
        ocp_nlp_constraints_model_set(nlp_config, nlp_dims, nlp_in, 0, "lbx", x0);
        ocp_nlp_constraints_model_set(nlp_config, nlp_dims, nlp_in, 0, "ubx", x0);
        for (int ii = 0; ii < N; ii++) {
            ocp_nlp_cost_model_set(nlp_config, nlp_dims, nlp_in, ii, "yref", yref);
            acados_update_params(ii, p_obs, 16);
            ocp_nlp_constraints_model_set(nlp_config, nlp_dims, nlp_in, ii, "lh", r_obs);
        }
        ocp_nlp_cost_model_set(nlp_config, nlp_dims, nlp_in, N, "yref", yref_e);
        acados_update_params(N, p_obs, 16);
        const int status = acados_solve();
        double u0[NU], x1[NX];
        ocp_nlp_out_get(nlp_config, nlp_dims, nlp_out, 0, "u", (void *)u0);
        ocp_nlp_out_get(nlp_config, nlp_dims, nlp_out, 1, "x", (void *)x1);
        const auto t1 = std::chrono::steady_clock::now();
        float lin = 0, qp = 0;
        usvmpc_shim_last_kernel_ms(&lin, &qp);
        if (t > 0) { wall_us += std::chrono::duration<double, std::micro>(t1 - t0).count(); kern_us += 1e3 * (lin + qp); }
        std::printf("tick %d status %d u0 %.17g x1", t, status, u0[0]);
        for (int i = 0; i < NX; i++) { std::printf(" %.17g", x1[i]); x0[i] = x1[i]; }
        std::printf("\n");
    }
    if (ticks > 1)
        std::printf("timing per tick over %d ticks: wall %.1f us, kernels %.1f us, outside the kernels %.1f us (%d setter calls, 2 getter calls)\n",
                    ticks - 1, wall_us / (ticks - 1), kern_us / (ticks - 1), (wall_us - kern_us) / (ticks - 1), 3 * N + 4);
    acados_free();
    return 0;
}
