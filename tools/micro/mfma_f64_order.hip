// v_mfma_f64_16x16x4_f64 on gfx950: in which order, and with which roundings, does one instruction add its four products to C?
// The wide (one-instance-per-wave) sweeps could form T = [B A]' P and G = H + T [B A] as chains of this instruction
// ONLY IF the result equals the 16-lane sweeps' chain of FMAs bit for bit: D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C)))) in some fixed
// order of k.  This program compares the instruction with all 24 orders of that chain (and with the unfused and the pairwise forms) on random
// and on cancellation-heavy data, and times a dependent chain of four against four dependent v_fma_f64.
// Layout (cdna_hip_programming.md): A lane l = A[l & 15][l >> 4], B lane l = B[l >> 4][l & 15], C / D reg v of lane l = [(l >> 4) + 4 v][l & 15].
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const double *A, const double *B, const double *C, double *D)
{
    const int l = threadIdx.x;
    v4d c;
    for (int v = 0; v < 4; v++) c[v] = C[((l >> 4) + 4 * v) * 16 + (l & 15)];
    const v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int v = 0; v < 4; v++) D[((l >> 4) + 4 * v) * 16 + (l & 15)] = d[v];
}

template <int MODE>
__global__ void k_rate(double *out, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    v4d c = {0, 0, 0, 0};
    double s = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { // four dependent MFMAs: one 16x16x16 product
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        } else {        // sixteen dependent FMAs: one column of the same product in the 16-lane layout
            for (int j = 0; j < 16; j++) s = __builtin_fma(a, b, s);
        }
    }
    out[threadIdx.x] = c[0] + c[1] + c[2] + c[3] + s;
}

int main()
{
    const int NT = 200;
    std::vector<double> A(64), B(64), C(256), D(256);
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 256 * 8);
    int perm[4] = {0, 1, 2, 3};
    std::vector<std::vector<int>> orders;
    do orders.push_back(std::vector<int>(perm, perm + 4)); while (std::next_permutation(perm, perm + 4));
    std::vector<long> hits(orders.size(), 0);
    long hit_unfused = 0, hit_pair = 0, hit_prodsum_first = 0, total = 0;
    srand(7);
    for (int t = 0; t < NT; t++) {
        const bool cancel = t % 2 == 1; // products that nearly cancel C: the order of the roundings shows
        for (auto &v : A) v = (rand() / (double)RAND_MAX - 0.5) * (cancel ? 1e3 : 2.0);
        for (auto &v : B) v = (rand() / (double)RAND_MAX - 0.5) * (cancel ? 1e3 : 2.0);
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                double c = (rand() / (double)RAND_MAX - 0.5);
                if (cancel) { c = 0; for (int k = 0; k < 4; k++) c -= A[i * 4 + k] * B[k * 16 + j]; c *= (1.0 + 1e-9 * (rand() % 7)); }
                C[i * 16 + j] = c;
            }
        hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                const double d = D[i * 16 + j];
                total++;
                for (size_t o = 0; o < orders.size(); o++) {
                    double s = C[i * 16 + j];
                    for (int q = 0; q < 4; q++) { const int k = orders[o][q]; s = std::fma(A[i * 4 + k], B[k * 16 + j], s); }
                    if (std::memcmp(&s, &d, 8) == 0) hits[o]++;
                }
                { double s = C[i * 16 + j]; for (int k = 0; k < 4; k++) { volatile double p = A[i * 4 + k] * B[k * 16 + j]; s = s + p; } if (std::memcmp(&s, &d, 8) == 0) hit_unfused++; }
                { double p01 = std::fma(A[i * 4 + 1], B[16 + j], A[i * 4] * B[j]), p23 = std::fma(A[i * 4 + 3], B[48 + j], A[i * 4 + 2] * B[32 + j]);
                  double s = (p01 + p23) + C[i * 16 + j]; if (std::memcmp(&s, &d, 8) == 0) hit_pair++; }
                { double s = 0; for (int k = 0; k < 4; k++) s = std::fma(A[i * 4 + k], B[k * 16 + j], s); s += C[i * 16 + j]; if (std::memcmp(&s, &d, 8) == 0) hit_prodsum_first++; }
            }
    }
    printf("v_mfma_f64_16x16x4_f64 against CPU forms, %ld entries (half of them with products that cancel C to 1e-9):\n", total);
    for (size_t o = 0; o < orders.size(); o++)
        if (hits[o] * 100 >= total * 60 || o == 0)
            printf("  fma chain from C in k order %d %d %d %d: %ld equal (%.1f %%)\n", orders[o][0], orders[o][1], orders[o][2], orders[o][3], hits[o], 100.0 * hits[o] / total);
    long best = 0; for (auto h : hits) best = std::max(best, h);
    printf("  best fma-chain order: %.1f %%; unfused (round each product, k ascending): %.1f %%; pairwise ((p0+p1)+(p2+p3))+C: %.1f %%; products summed from 0 then + C: %.1f %%\n",
           100.0 * best / total, 100.0 * hit_unfused / total, 100.0 * hit_pair / total, 100.0 * hit_prodsum_first / total);
    double *dout; hipMalloc(&dout, 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        const int iters = 200000;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(64), 0, 0, dout, iters); else hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(64), 0, 0, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  lone wave, %s: %.1f ns per 16x16x16 product-equivalent (%s)\n", mode == 0 ? "4 dependent v_mfma_f64_16x16x4" : "16 dependent v_fma_f64 (ONE column; the sweeps need 14 - 16 columns)", ms * 1e6 / iters,
               mode == 0 ? "all 256 entries" : "x 14 - 16 for the matrix");
    }
    return 0;
}
