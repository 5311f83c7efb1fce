/* usv_guidance_oracle.h — see usv_guidance_oracle.c. TEST INFRASTRUCTURE ONLY. */
#ifndef USV_GUIDANCE_ORACLE_H
#define USV_GUIDANCE_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
#define USV_GUIDANCE_LMAX 64

/* new waypoint list: k = 1, past_psied = wrap(psi - ak) */
void usv_guidance_reset_ref(const double *waypoints, double psi, int *k, float *past_psied);
/* body-frame obstacle list obs[n][3] = (x, y, R) -> p_obs[2K] (NED, single-precision values),
 * r_obs[K] (R + boat radius); chosen[K] (may be NULL) = source index or -1 for padding */
void usv_guidance_obstacles_ref(int K, double psi, double nedx, double nedy, const double *obs, int n,
                                double *p_obs, double *r_obs, int *chosen);
/* one tick, input side: returns 1 if a control tick is due (k < npts), else 0 */
int usv_guidance_prepare_ref(int K, const double *vel_uv, const double *pose, const double *waypoints, int npts,
                             const double *obs, int n_obs, int *k, float *past_psied,
                             double *x0, double *p_obs, double *r_obs, double *ak_out, double *ye_out);
/* one tick, output side */
void usv_guidance_publish_ref(double x1_psied, double u0, double ak, float *past_psied,
                              double *heading, double *r_des, double *speed);
/* obstacle simulator (catkin_ws/src/simulation/scripts/obstacle_sim_node.py:56-81,101-117): the world obstacles
 * (X, Y, R) within max_radius of the vessel, in the body frame, in list order; returns their number (<= lmax). */
int usv_obstacle_sim_ref(const double *pose, const double *world, int n_world, double max_radius, int lmax,
                         double *obstacles);

#ifdef __cplusplus
}
#endif
#endif
