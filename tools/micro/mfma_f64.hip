// v_mfma_f64_4x4x4_4b_f64 on gfx950: (1) the lane -> element map, found by brute force against a CPU product (four independent
// 4x4x4 blocks, one per 16-lane row - the granularity of "one OCP instance per DPP row"); (2) issue cost: a dependent chain and
// independent streams, alone and next to an FP64 VALU stream in a second wave of the same SIMD.  Evidence for DESIGN.md section 7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

__global__ void k_map(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

template <int MODE> // 0: dependent mfma chain, 1: 4 independent mfma streams, 2: fp64 fma stream (VALU), 3: mfma (even waves) + fma (odd waves)
__global__ void __launch_bounds__(128) k_rate(double *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const bool do_mfma = MODE == 0 || MODE == 1 || (MODE == 3 && (wave & 1) == 0);
    if (do_mfma) {
        for (int i = 0; i < iters; i++) {
            if (MODE == 0) {
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
            }
        }
    } else {
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_fma(a, b, c0); c1 = __builtin_fma(a, b, c1); c2 = __builtin_fma(a, b, c2); c3 = __builtin_fma(a, b, c3);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

int main()
{
    // ---- (1) layout
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    for (int l = 0; l < 64; l++) { ha[l] = 1.0 + 0.37 * l + 0.011 * l * l; hb[l] = 2.0 - 0.23 * l + 0.007 * l * l * l; }
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    // hypotheses: lane = 4^p0 * row + 4^p1 * col + 4^p2 * block with (p0, p1, p2) a permutation of (0, 1, 2), separately for A[i][k], B[k][j], D[i][j]
    const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    auto at = [&](const int *p, int r, int c, int blk) { return (r << (2 * p[0])) + (c << (2 * p[1])) + (blk << (2 * p[2])); };
    for (int ma = 0; ma < 6; ma++) for (int mb = 0; mb < 6; mb++) for (int md = 0; md < 6; md++) {
        double err = 0;
        for (int blk = 0; blk < 4; blk++) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += ha[at(perms[ma], i, k, blk)] * hb[at(perms[mb], k, j, blk)];
            err = fmax(err, fabs(s - hd[at(perms[md], i, j, blk)]) / fabs(s));
        }
        if (err < 1e-12)
            printf("layout (lane = row*4^a + col*4^b + block*4^c):  A[i][k] (a,b,c) = (%d,%d,%d);  B[k][j] = (%d,%d,%d);  D[i][j] = (%d,%d,%d)   max rel err %.1e\n",
                   perms[ma][0], perms[ma][1], perms[ma][2], perms[mb][0], perms[mb][1], perms[mb][2], perms[md][0], perms[md][1], perms[md][2], err);
    }
    // ---- (2) issue cost
    double *out; hipMalloc(&out, 8 * 1024 * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400000, blocks = 1024;  // 2 waves per block, 1024 blocks = 2 waves per SIMD; long enough for the clock to settle
    for (int w = 0; w < 4; w++) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(128), 0, 0, out, iters);   // ramp the clock up
    hipDeviceSynchronize();
    auto run = [&](int mode, const char *what) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(128), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(128), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(128), 0, 0, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(128), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
        }
        // per wave: 4 * iters instructions; waves per SIMD = blocks * 2 / 1024
        const double per = best * 1e-3 / (4.0 * iters) / (blocks * 2 / 1024.0);
        printf("%-58s %8.3f ms   %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.1 GHz)\n", what, best, per * 1e9, per * 2.1e9);
    };
    run(0, "mfma_f64_4x4x4_4b, dependent chain");
    run(1, "mfma_f64_4x4x4_4b, 4 independent accumulators");
    run(2, "v_fma_f64, 4 independent accumulators");
    run(3, "half the waves mfma, half v_fma_f64 (same total count)");
    return 0;
}
