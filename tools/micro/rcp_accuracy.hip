#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *r0, double *r1, double *q0, double *q1, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    double r = __builtin_amdgcn_rcp(a);
    r0[i] = r;
    r1[i] = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    double y = __builtin_amdgcn_rsq(a);
    q0[i] = y;
    q1[i] = __builtin_fma(0.5 * y, __builtin_fma(-a * y, y, 1.0), y);
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), r0(n), r1(n), q0(n), q1(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-12, 12);
    for (auto &v : x) v = std::pow(10.0, u(g));
    double *dx, *d0, *d1, *e0, *e1;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&e0, n * 8); hipMalloc(&e1, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, e0, e1, n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(q0.data(), e0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q1.data(), e1, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, s0 = 0, s1 = 0;
    for (int i = 0; i < n; i++) {
        const double t = 1.0 / x[i], ts = 1.0 / std::sqrt(x[i]);
        m0 = std::fmax(m0, std::fabs(r0[i] - t) / t); m1 = std::fmax(m1, std::fabs(r1[i] - t) / t);
        s0 = std::fmax(s0, std::fabs(q0[i] - ts) / ts); s1 = std::fmax(s1, std::fabs(q1[i] - ts) / ts);
    }
    printf("rcp raw max rel err %.3e  one newton %.3e | rsq raw %.3e  one newton %.3e\n", m0, m1, s0, s1);
    return 0;
}
