// guidance.hpp — the arithmetic either side of the solver call in the reference's obstacle-avoidance
// ROS node (class NMPC, /root/reference/catkin_ws/src/nmpc_ca/src/nmpc_guidance_ca1.cpp), batched:
// one thread per instance.  Input side: velocityCallback :223-230, obstaclesCallback :252-346
// (nearest-K selection by sqrt(x^2+y^2) - (R + boat radius), padding at (1000,1000,0)), body2NED
// :348-363 (single-precision Eigen product), waypoint_manager :441-491, control :493-574.  Output
// side: control :583-600.  New waypoint list: main :616-632.  Quirks of the node are kept: the
// float `past_psied` member (:156), the always-true enum test at :496 (beta = atan2(v,u)).
#pragma once
#include "params.hpp"
#include <hip/hip_runtime.h>

namespace usv {

constexpr int GUIDANCE_LMAX = 64;   // obstacles per instance on input
constexpr double BOAT_RADIUS = 0.5; // :139
constexpr double INIT_OBS_POS = 1000.0;
constexpr double D_SPEED = 0.7;     // :453

struct GuidancePtrs {
    const double *vel;   // [B][2] u, v
    const double *pose;  // [B][3] nedx, nedy, psi
    const double *wp;    // [B][2*npts]
    int npts;
    const double *obs;   // [B][lmax][3] body x, y, R
    const int *nobs;     // [B]
    int lmax;
    int *k;              // [B] waypoint index (state)
    float *past_psied;   // [B] (state; float as in the node)
    double *ak, *ye;     // [B]
    int *active;         // [B]
    double *heading, *rdes, *speed; // [B] outputs of the publish side
};

__device__ __forceinline__ double wrap_pi(double a)
{
    if (fabs(a) > M_PI) a = (a / fabs(a)) * (fabs(a) - 2.0 * M_PI);
    return a;
}
__device__ __forceinline__ float wrap_pi_f(float a)
{ // float variable, double arithmetic, rounded on assignment (as the node's float members)
    if (fabs((double)a) > M_PI) a = (float)(((double)a / fabs((double)a)) * (fabs((double)a) - 2.0 * M_PI));
    return a;
}

__device__ __forceinline__ void body2ned(double psi, double nedx, double nedy, double bx, double by,
                                         double &ox, double &oy)
{ // Matrix3f * Vector3f, row i = (R_i0*b0 + R_i1*b1) + R_i2*b2, no FMA contraction
    const float c = (float)cos(psi), s = (float)sin(psi);
    const float b0 = (float)bx, b1 = (float)by;
    const float r0 = __fadd_rn(__fadd_rn(__fmul_rn(c, b0), __fmul_rn(-s, b1)), 0.0f);
    const float r1 = __fadd_rn(__fadd_rn(__fmul_rn(s, b0), __fmul_rn(c, b1)), 0.0f);
    ox = (double)(float)((double)r0 + nedx);
    oy = (double)(float)((double)r1 + nedy);
}

// The obstacle-simulator node's simulate() (/root/reference/catkin_ws/src/simulation/scripts/
// obstacle_sim_node.py:56-81, ned_to_body :101-117): every world obstacle (X, Y, R) closer to the vessel than
// max_visible_radius is reported in the body frame, in list order.  The node inverts the rotation matrix with
// numpy.linalg.inv; here the inverse is written out (adjugate / determinant), which agrees to rounding.
// The body-frame list is what obstaclesCallback of the NMPC node receives: it is written straight into the
// front end's input buffers, so a scenario sweep needs no host round trip between "sensor" and solver.
static __global__ void usv_obstacle_sim(GuidancePtrs G, const double *world, int nw, double max_radius, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double nedx = G.pose[b * 3 + 0], nedy = G.pose[b * 3 + 1], yaw = G.pose[b * 3 + 2];
    const double c = cos(yaw), s = sin(yaw);
    const double det = c * c - (-s) * s;
    const double i00 = c / det, i01 = s / det, i10 = -s / det, i11 = c / det;
    double *out = const_cast<double *>(G.obs) + (long)b * G.lmax * 3;
    const double *w = world + (long)b * nw * 3;
    int n = 0;
    for (int i = 0; i < nw && n < G.lmax; i++) {
        const double dx = w[3 * i] - nedx, dy = w[3 * i + 1] - nedy;
        const double dist = sqrt(dx * dx + dy * dy);
        if (dist < max_radius) {
            out[3 * n + 0] = __dadd_rn(__dmul_rn(i00, dx), __dmul_rn(i01, dy));
            out[3 * n + 1] = __dadd_rn(__dmul_rn(i10, dx), __dmul_rn(i11, dy));
            out[3 * n + 2] = w[3 * i + 2];
            n++;
        }
    }
    const_cast<int *>(G.nobs)[b] = n;
}

static __global__ void usv_guidance_reset(GuidancePtrs G, const double *psi, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *w = G.wp + (long)b * 2 * G.npts;
    const double ak = atan2(w[3] - w[1], w[2] - w[0]);
    G.k[b] = 1;
    G.past_psied[b] = wrap_pi_f((float)(psi[b] - ak));
}

// writes x0, stage-0 p and lh of the solver (static-obstacle mode) for instance b
static __global__ void usv_guidance_pre(DevPtrs P, GuidancePtrs G)
{
    const DevSpec &S = *P.spec;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    const int K = S.K, N = S.N;
    const double u_cb = (G.vel[2 * b] == 0.0) ? 0.001 : G.vel[2 * b];
    const double v_cb = G.vel[2 * b + 1];
    const double nedx = G.pose[3 * b], nedy = G.pose[3 * b + 1], psi = G.pose[3 * b + 2];
    double *p = const_cast<double *>(P.p) + (long)b * (N + 1) * 2 * K;
    double *lh = const_cast<double *>(P.lh) + (long)b * N * K;
    // ---- obstacles
    const double *ob = G.obs + (long)b * G.lmax * 3;
    int n = G.nobs[b];
    n = n < 0 ? 0 : (n > G.lmax ? G.lmax : n);
    for (int i = 0; i < K; i++) {
        p[2 * i] = (double)(float)INIT_OBS_POS;
        p[2 * i + 1] = (double)(float)INIT_OBS_POS;
        lh[i] = 0.0;
    }
    if (n > K) {
        // K smallest of sqrt(x^2+y^2) - (R + boat radius), ties by index: selection by rank
        for (int i = 0; i < n; i++) {
            const double di = sqrt(ob[3 * i] * ob[3 * i] + ob[3 * i + 1] * ob[3 * i + 1]) - (ob[3 * i + 2] + BOAT_RADIUS);
            int rank = 0;
            for (int j = 0; j < n; j++) {
                const double dj = sqrt(ob[3 * j] * ob[3 * j] + ob[3 * j + 1] * ob[3 * j + 1]) - (ob[3 * j + 2] + BOAT_RADIUS);
                rank += (dj < di || (dj == di && j < i)) ? 1 : 0;
            }
            if (rank < K) {
                double ox, oy;
                body2ned(psi, nedx, nedy, ob[3 * i], ob[3 * i + 1], ox, oy);
                p[2 * rank] = ox;
                p[2 * rank + 1] = oy;
                lh[rank] = (double)(float)(ob[3 * i + 2] + BOAT_RADIUS);
            }
        }
    } else {
        for (int i = 0; i < n; i++) {
            double ox, oy;
            body2ned(psi, nedx, nedy, ob[3 * i], ob[3 * i + 1], ox, oy);
            p[2 * i] = ox;
            p[2 * i + 1] = oy;
            lh[i] = (double)(float)(ob[3 * i + 2] + BOAT_RADIUS);
        }
    }
    // ---- waypoint manager
    int k = G.k[b];
    float pp = G.past_psied[b];
    const double *w = G.wp + (long)b * 2 * G.npts;
    int active = 0;
    double ak = 0.0, ye = 0.0;
    if (k < G.npts) {
        double x1 = w[2 * k - 2], y1 = w[2 * k - 1], x2 = w[2 * k], y2 = w[2 * k + 1];
        const double distance = sqrt((x2 - nedx) * (x2 - nedx) + (y2 - nedy) * (y2 - nedy));
        ak = atan2(y2 - y1, x2 - x1);
        active = 1;
        if (distance > 1) {
            ye = -(nedx - x1) * sin(ak) + (nedy - y1) * cos(ak);
        } else {
            k += 1;
            if (k < G.npts) {
                x1 = w[2 * k - 2]; y1 = w[2 * k - 1]; x2 = w[2 * k]; y2 = w[2 * k + 1];
                const double ak2 = atan2(y2 - y1, x2 - x1);
                ye = -(nedx - x1) * sin(ak2) + (nedy - y1) * cos(ak2);
                pp = wrap_pi_f((float)((double)pp - ak2 + ak));
                ak = ak2;
            } else {
                active = 0;
            }
        }
    }
    G.k[b] = k;
    G.active[b] = active;
    if (active) {
        G.past_psied[b] = pp;
        G.ak[b] = ak;
        G.ye[b] = ye;
        const double beta = atan2(v_cb, u_cb);
        double *x0 = const_cast<double *>(P.x0) + (long)b * 8;
        x0[0] = u_cb; x0[1] = v_cb; x0[2] = ye; x0[3] = wrap_pi(psi + beta - ak); x0[4] = (double)pp;
        x0[5] = nedx; x0[6] = nedy; x0[7] = psi;
    }
}

static __global__ void usv_guidance_post(DevPtrs P, GuidancePtrs G)
{
    const DevSpec &S = *P.spec;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    if (!G.active[b]) {
        G.heading[b] = 0.0; G.rdes[b] = 0.0; G.speed[b] = 0.0;
        return;
    }
    const double x1_psied = P.x[((long)b * (S.N + 1) + 1) * 8 + 4];
    const float psid = wrap_pi_f((float)(x1_psied + G.ak[b]));
    G.past_psied[b] = (float)x1_psied;
    G.heading[b] = (double)psid;
    G.rdes[b] = P.u[(long)b * S.N];
    G.speed[b] = D_SPEED;
}

} // namespace usv
