// advance.hpp — the closed-loop hand-over between two ticks, as the reference's callers do it on the host
// (x0 = get(1,"x"); set(0,"lbx",x0): /root/reference/catkin_ws/src/nmpc_ca/scripts/usv_guidance_ca1/main.py:169-175):
// the next initial state is the predicted x_1 plus an optional Gaussian disturbance (the commented "Add noise" hooks of
// scripts/usv_pf_ca/main.py:181-183).  No trajectory shift, as in the reference.
// One function for the stand-alone kernel (usv_advance, usvmpc.hip) and for the closed-loop launch, whose waves hand an
// instance over themselves when its QP has finished (qp_ipm.hpp): the same element gets the same number either way.
#pragma once
#include "lanes.hpp"

namespace usv {

USV_DEV unsigned long long splitmix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// x1: state j of the predicted x_1 of instance b; i = b * nx + j (the counter of the disturbance stream `seed`)
USV_DEV double advance_value(double x1, double sigma, unsigned long long seed, long i, bool disturbed)
{
    double v = x1;
    if (sigma != 0.0 && disturbed) {
        const unsigned long long h1 = splitmix64(seed ^ (unsigned long long)(2 * i));
        const unsigned long long h2 = splitmix64(seed ^ (unsigned long long)(2 * i + 1));
        const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740993.0);
        const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
        v += sigma * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
    return v;
}

} // namespace usv
