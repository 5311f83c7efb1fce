/*
 * acados_solver_usv_model_guidance_ca1.h — stand-in for the header acados generates for the model
 * `usv_model_guidance_ca1` (2020-era global-state API, no capsule), implemented on libusvmpc.so.
 * With `-I<this dir>` the reference node catkin_ws/src/nmpc_ca/src/nmpc_guidance_ca1.cpp finds every
 * acados header it includes (:18-27) and links against libacados_ocp_solver_usv_model_guidance_ca1.so
 * (CMakeLists.txt:178-185) unchanged.  The node defines the nlp_* globals itself (:44-52);
 * acados_create() fills them with opaque tokens.
 */
#ifndef ACADOS_SOLVER_USV_MODEL_GUIDANCE_CA1_H_
#define ACADOS_SOLVER_USV_MODEL_GUIDANCE_CA1_H_

#ifdef __cplusplus
extern "C" {
#endif

/* opaque acados types: the node only passes these pointers back into the functions below */
typedef struct ocp_nlp_in ocp_nlp_in;
typedef struct ocp_nlp_out ocp_nlp_out;
typedef struct ocp_nlp_solver ocp_nlp_solver;
typedef struct ocp_nlp_plan ocp_nlp_plan;
typedef struct ocp_nlp_config ocp_nlp_config;
typedef struct ocp_nlp_dims ocp_nlp_dims;
typedef struct external_function_param_casadi external_function_param_casadi;

/* defined by the caller (nmpc_guidance_ca1.cpp:44-52) */
extern ocp_nlp_in *nlp_in;
extern ocp_nlp_out *nlp_out;
extern ocp_nlp_solver *nlp_solver;
extern void *nlp_opts;
extern ocp_nlp_plan *nlp_solver_plan;
extern ocp_nlp_config *nlp_config;
extern ocp_nlp_dims *nlp_dims;

/* generated solver entry points (nmpc_guidance_ca1.cpp:165,220,570,577) */
int acados_create(void);
int acados_solve(void);
int acados_free(void);
int acados_update_params(int stage, double *value, int np_);

/* libacados accessors used by the node (:515-516,569-573,583-586) */
int ocp_nlp_constraints_model_set(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_in *in, int stage,
                                  const char *field, void *value);
int ocp_nlp_cost_model_set(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_in *in, int stage,
                           const char *field, void *value);
void ocp_nlp_out_get(ocp_nlp_config *config, ocp_nlp_dims *dims, ocp_nlp_out *out, int stage,
                     const char *field, void *value);

#ifdef __cplusplus
}
#endif
#endif
