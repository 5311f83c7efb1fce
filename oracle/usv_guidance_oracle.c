/*
 * usv_guidance_oracle.c — CPU restatement of the arithmetic either side of the solver call in the
 * reference's obstacle-avoidance ROS node (class NMPC in
 * /root/reference/catkin_ws/src/nmpc_ca/src/nmpc_guidance_ca1.cpp).  TEST INFRASTRUCTURE ONLY (same
 * rules as usv_oracle.c).  The node cannot be built here (ROS, Eigen, generated acados solver), and
 * the reference has no tests for it, so this follows the source line by line:
 *
 *   velocityCallback      :223-230   u == 0 -> 0.001
 *   obstaclesCallback     :252-346   > K obstacles: keep the K with the smallest
 *                                    sqrt(x^2+y^2) - (R + boat_radius); else pad with (1000,1000,0)
 *   body2NED              :348-363   Eigen::Matrix3f * Vector3f (single precision!) + NED position
 *   initializeObstacles   :365-376
 *   sortVec               :422-438   std::sort on indices (ties: broken here by index, i.e. stable)
 *   waypoint_manager      :441-491   segment selection, switch radius 1 m, psied re-referencing
 *   control (input part)  :493-574   beta, chie wrap, x0, p_obs, r_obs
 *   control (output part) :583-600   psid = float(x1[psied] + ak) wrapped, past_psied, desired r
 *   main (new waypoints)  :616-632   k = 1, past_psied = wrap(psi - ak)
 *
 * Reproduced quirks: `past_psied` is a float member (:156), so every value stored in it is rounded
 * to single precision; `if (v*v + u*u > 0)` (:496) tests the enum constants u=0, v=1 and is always
 * true, so beta = atan2(v, u) without the 0.001; pow(x, 0.5) (:451) is taken as sqrt.
 */
#define _DEFAULT_SOURCE /* M_PI under -std=c99 */
#include "usv_guidance_oracle.h"
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define BOAT_RADIUS 0.5
#define INIT_OBS_POS 1000.0
#define D_SPEED 0.7

static double wrap_pi(double a)
{ /* (a/|a|)*(|a| - 2pi) when |a| > pi, as written at :479-482, :500-502, :629-631 */
    if (fabs(a) > M_PI) a = (a / fabs(a)) * (fabs(a) - 2.0 * M_PI);
    return a;
}

void usv_guidance_reset_ref(const double *waypoints, double psi, int *k, float *past_psied)
{ /* main(): :616-632 */
    const double x1 = waypoints[0], y1 = waypoints[1], x2 = waypoints[2], y2 = waypoints[3];
    const double ak = atan2(y2 - y1, x2 - x1);
    *k = 1;
    *past_psied = (float)(psi - ak);
    if (fabs(*past_psied) > M_PI) *past_psied = (float)((*past_psied / fabs(*past_psied)) * (fabs(*past_psied) - 2 * M_PI));
}

static void body2ned(double psi, double nedx, double nedy, double bx, double by, float *ox, float *oy)
{ /* :348-363: R (float) * body (float), row i = (R_i0*b0 + R_i1*b1) + R_i2*b2, then + double position */
    const float c = (float)cos(psi), s = (float)sin(psi);
    const float b0 = (float)bx, b1 = (float)by, b2 = 0.0f;
    volatile float p00 = c * b0, p01 = (-s) * b1, p02 = 0.0f * b2;
    volatile float p10 = s * b0, p11 = c * b1;
    volatile float r0 = p00 + p01, r1 = p10 + p11;
    const float nx = r0 + p02, ny = r1 + p02;
    *ox = (float)((double)nx + nedx);
    *oy = (float)((double)ny + nedy);
}

void usv_guidance_obstacles_ref(int K, double psi, double nedx, double nedy, const double *obs, int n,
                                double *p_obs, double *r_obs, int *chosen)
{ /* :252-346 */
    int i, j;
    for (i = 0; i < K; i++) { /* initializeObstacles */
        p_obs[2 * i] = (double)(float)INIT_OBS_POS;
        p_obs[2 * i + 1] = (double)(float)INIT_OBS_POS;
        r_obs[i] = 0.0;
        if (chosen) chosen[i] = -1;
    }
    if (n > K) {
        int idx[USV_GUIDANCE_LMAX];
        double dist[USV_GUIDANCE_LMAX];
        for (i = 0; i < n; i++) {
            const double radius = obs[3 * i + 2] + BOAT_RADIUS;
            dist[i] = sqrt(obs[3 * i] * obs[3 * i] + obs[3 * i + 1] * obs[3 * i + 1]) - radius;
            idx[i] = i;
        }
        for (i = 1; i < n; i++) { /* stable insertion sort, ascending distance */
            const int t = idx[i];
            for (j = i - 1; j >= 0 && dist[idx[j]] > dist[t]; j--) idx[j + 1] = idx[j];
            idx[j + 1] = t;
        }
        for (i = 0; i < K; i++) {
            const int s = idx[i];
            float ox, oy;
            body2ned(psi, nedx, nedy, obs[3 * s], obs[3 * s + 1], &ox, &oy);
            p_obs[2 * i] = ox; p_obs[2 * i + 1] = oy;
            r_obs[i] = (double)(float)(obs[3 * s + 2] + BOAT_RADIUS);
            if (chosen) chosen[i] = s;
        }
    } else {
        for (i = 0; i < n; i++) {
            float ox, oy;
            body2ned(psi, nedx, nedy, obs[3 * i], obs[3 * i + 1], &ox, &oy);
            p_obs[2 * i] = ox; p_obs[2 * i + 1] = oy;
            r_obs[i] = (double)(float)(obs[3 * i + 2] + BOAT_RADIUS);
            if (chosen) chosen[i] = i;
        }
    }
}

int usv_guidance_prepare_ref(int K, const double *vel_uv, const double *pose, const double *waypoints, int npts,
                             const double *obs, int n_obs, int *k, float *past_psied,
                             double *x0, double *p_obs, double *r_obs, double *ak_out, double *ye_out)
{
    const double u_cb = (vel_uv[0] == 0.0) ? 0.001 : vel_uv[0]; /* :225-228 */
    const double v_cb = vel_uv[1];
    const double nedx = pose[0], nedy = pose[1], psi = pose[2];
    double ak, ye;
    usv_guidance_obstacles_ref(K, psi, nedx, nedy, obs, n_obs, p_obs, r_obs, 0);
    /* waypoint_manager: :441-491 */
    if (!(*k < npts)) return 0; /* d_speed = 0, no control tick */
    {
        double x1 = waypoints[2 * *k - 2], y1 = waypoints[2 * *k - 1];
        double x2 = waypoints[2 * *k], y2 = waypoints[2 * *k + 1];
        const double distance = sqrt((x2 - nedx) * (x2 - nedx) + (y2 - nedy) * (y2 - nedy));
        ak = atan2(y2 - y1, x2 - x1);
        if (distance > 1) {
            ye = -(nedx - x1) * sin(ak) + (nedy - y1) * cos(ak);
        } else {
            double ak2;
            *k += 1;
            if (!(*k < npts)) return 0; /* the reference would read past the list here */
            x1 = waypoints[2 * *k - 2]; y1 = waypoints[2 * *k - 1];
            x2 = waypoints[2 * *k]; y2 = waypoints[2 * *k + 1];
            ak2 = atan2(y2 - y1, x2 - x1);
            ye = -(nedx - x1) * sin(ak2) + (nedy - y1) * cos(ak2);
            *past_psied = (float)(*past_psied - ak2 + ak);
            if (fabs(*past_psied) > M_PI)
                *past_psied = (float)((*past_psied / fabs(*past_psied)) * (fabs(*past_psied) - 2 * M_PI));
            ak = ak2;
        }
    }
    { /* control(): :495-511 */
        const double beta = atan2(v_cb, u_cb);
        const double chie = wrap_pi(psi + beta - ak);
        x0[0] = u_cb; x0[1] = v_cb; x0[2] = ye; x0[3] = chie; x0[4] = (double)*past_psied;
        x0[5] = nedx; x0[6] = nedy; x0[7] = psi;
    }
    *ak_out = ak;
    *ye_out = ye;
    return 1;
}

void usv_guidance_publish_ref(double x1_psied, double u0, double ak, float *past_psied,
                              double *heading, double *r_des, double *speed)
{ /* control(): :587-598 */
    float psid = (float)(x1_psied + ak);
    if (fabs(psid) > M_PI) psid = (float)((psid / fabs(psid)) * (fabs(psid) - 2 * M_PI));
    *past_psied = (float)x1_psied;
    *heading = (double)psid;
    *r_des = u0;
    *speed = D_SPEED;
}

/* obstacle_sim_node.py simulate() :56-81 with ned_to_body :101-117.  The node inverts the 2x2 rotation with
 * numpy.linalg.inv (LAPACK); the adjugate / determinant form used here agrees with it to rounding. */
int usv_obstacle_sim_ref(const double *pose, const double *world, int n_world, double max_radius, int lmax,
                         double *obstacles)
{
    const double nedx = pose[0], nedy = pose[1], yaw = pose[2];
    const double c = cos(yaw), s = sin(yaw);
    const double det = c * c - (-s) * s;
    const double i00 = c / det, i01 = s / det, i10 = -s / det, i11 = c / det;
    int i, n = 0;
    for (i = 0; i < n_world && n < lmax; i++) {
        const double dx = world[3 * i] - nedx, dy = world[3 * i + 1] - nedy;
        const double dist = pow(dx * dx + dy * dy, 0.5);
        if (dist < max_radius) {
            obstacles[3 * n + 0] = i00 * dx + i01 * dy;
            obstacles[3 * n + 1] = i10 * dx + i11 * dy;
            obstacles[3 * n + 2] = world[3 * i + 2];
            n++;
        }
    }
    return n;
}
