// acados_shim.cpp — libacados_ocp_solver_usv_model_guidance_ca1.so on top of libusvmpc.so: the symbols the
// reference ROS node links against (see include/acados_solver_usv_model_guidance_ca1.h), one instance.
// The OCP definition is baked in exactly as acados bakes it into generated code, from
// /root/reference/catkin_ws/src/nmpc_ca/scripts/usv_guidance_ca1/acados_settings.py:64-194 with
// Tf = 5, N = 100 (scripts/usv_guidance_ca1/main.py:54-55, src/nmpc_guidance_ca1.cpp:64).
#include "include/acados_solver_usv_model_guidance_ca1.h"
#include "../../../include/usvmpc.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
constexpr int N = 100, NX = 8, NU = 1, NY = 9, K = 8;
usvmpc_handle *g_h = nullptr;
int g_token; // address used for the opaque nlp_* pointers

int fail(const char *what)
{
    std::fprintf(stderr, "acados shim: %s: %s\n", what, g_h ? usvmpc_last_error(g_h) : "no solver");
    return 1;
}
} // namespace

extern "C" {

int acados_create(void)
{
    if (g_h) return 0;
    usvmpc_desc d;
    std::memset(&d, 0, sizeof(d));
    d.model = USVMPC_MODEL_GUIDANCE_CA1;
    d.N = N; d.Tf = 5.0; d.K = K; d.batch = 1; d.device = 0;
    usvmpc_default_options(&d);
    const double Q[NX] = {0, 0, 0.05, 0.01, 0, 0, 0, 0}, Qe[NX] = {0, 0, 0.1, 0.05, 0, 0, 0, 0};
    for (int i = 0; i < NX; i++) {
        d.W[i * NY + i] = Q[i];
        d.W_e[i * NX + i] = Qe[i];
        d.Vx[i * NX + i] = 1.0;
        d.Vx_e[i * NX + i] = 1.0;
    }
    d.W[8 * NY + 8] = 0.2;
    d.Vu[8 * NU + 0] = 1.0;
    d.nbu = 1; d.idxbu[0] = 0; d.lbu[0] = -0.5; d.ubu[0] = 0.5;
    d.nbx = 0;
    d.soft = 1;
    for (int i = 0; i < K; i++) {
        d.uh[i] = 1000000.0; d.lsh[i] = -0.2; d.ush[i] = 0.0;
        d.zl[i] = 1.0; d.zu[i] = 1.0; d.Zl[i] = 0.0; d.Zu[i] = 0.0;
    }
    if (usvmpc_create(&d, &g_h)) { g_h = nullptr; return 1; }
    // acados_create(): parameter_values = 100, lh = 1.5 (acados_settings.py:126-137,185), x = x0 = 0, u = 0
    std::vector<double> p(2 * K, 100.0), lh(K, 1.5);
    for (int k = 0; k <= N; k++) usvmpc_set(g_h, "p", k, p.data(), 2 * K);
    for (int k = 0; k < N; k++) usvmpc_set(g_h, "lh", k, lh.data(), K);
    nlp_in = (ocp_nlp_in *)&g_token; nlp_out = (ocp_nlp_out *)&g_token; nlp_solver = (ocp_nlp_solver *)&g_token;
    nlp_opts = &g_token; nlp_solver_plan = (ocp_nlp_plan *)&g_token; nlp_config = (ocp_nlp_config *)&g_token;
    nlp_dims = (ocp_nlp_dims *)&g_token;
    return 0;
}

int acados_free(void)
{
    if (g_h) usvmpc_destroy(g_h);
    g_h = nullptr;
    return 0;
}

int acados_solve(void)
{
    if (!g_h) return fail("acados_solve");
    int st = 0;
    const int rc = usvmpc_solve(g_h, &st);
    return rc < 0 ? 1 : st;
}

int acados_update_params(int stage, double *value, int np_)
{
    if (!g_h || usvmpc_set(g_h, "p", stage, value, (size_t)np_) != 0) return fail("acados_update_params");
    return 0;
}

int ocp_nlp_constraints_model_set(ocp_nlp_config *, ocp_nlp_dims *, ocp_nlp_in *, int stage, const char *field, void *value)
{
    const std::string f(field ? field : "");
    int rc;
    if (f == "lbx" || f == "ubx") {
        if (stage != 0) { std::fprintf(stderr, "acados shim: %s is the x0 embedding (stage 0 only)\n", f.c_str()); return 1; }
        rc = usvmpc_set(g_h, "x0", 0, (const double *)value, NX); // the node always writes lbx = ubx = x0 (:515-516)
    } else if (f == "lh") {
        rc = usvmpc_set(g_h, "lh", stage, (const double *)value, K);
    } else {
        std::fprintf(stderr, "acados shim: constraints field '%s' is fixed by the OCP definition\n", f.c_str());
        return 1;
    }
    return rc ? fail("ocp_nlp_constraints_model_set") : 0;
}

int ocp_nlp_cost_model_set(ocp_nlp_config *, ocp_nlp_dims *, ocp_nlp_in *, int stage, const char *field, void *value)
{
    if (std::string(field ? field : "") != "yref") { std::fprintf(stderr, "acados shim: only yref can be set\n"); return 1; }
    return usvmpc_set(g_h, "yref", stage, (const double *)value, stage == N ? NX : NY) ? fail("ocp_nlp_cost_model_set") : 0;
}

// not an acados symbol: kernel time of the last acados_solve (HIP events on the solver's stream), for the overhead measurement of
// tests/shim_harness.cpp
int usvmpc_shim_last_kernel_ms(float *linearize_ms, float *qp_ms)
{
    return g_h ? usvmpc_last_kernel_ms(g_h, linearize_ms, qp_ms) : -1;
}

void ocp_nlp_out_get(ocp_nlp_config *, ocp_nlp_dims *, ocp_nlp_out *, int stage, const char *field, void *value)
{
    const std::string f(field ? field : "");
    if (f == "x") { if (usvmpc_get(g_h, "x", stage, (double *)value, NX)) fail("ocp_nlp_out_get"); }
    else if (f == "u") { if (usvmpc_get(g_h, "u", stage, (double *)value, NU)) fail("ocp_nlp_out_get"); }
    else std::fprintf(stderr, "acados shim: ocp_nlp_out_get field '%s' not provided\n", f.c_str());
}

} // extern "C"
