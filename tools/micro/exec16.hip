// Does a wave64 FP64 VALU instruction get cheaper when only one 16-lane row of the wave is in EXEC?  (gfx950)
// If the hardware skips the passes of inactive rows, a lone instance per wave (small batches: one OCP instance = one 16-lane row)
// would issue up to 4x faster with its three idle rows switched off instead of frozen by selects.  s_memtime ticks per instruction, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v120","v121","v122","v123"
template <int T> __global__ void __launch_bounds__(64) k(long long *cyc, double *sink, int iters, int active)
{
    if ((int)threadIdx.x >= active) return;     // EXEC = the first `active` lanes for the rest of the kernel
    asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0x3ff00000\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0x3ff00000\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0x3ff00000\n"
                 "v_mov_b32 v106, 0\n v_mov_b32 v107, 0x3ff00000\n v_mov_b32 v108, 0\n v_mov_b32 v109, 0x3ff00000\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0x3ff00000\n"
                 "v_mov_b32 v112, 0\n v_mov_b32 v113, 0x3ff00000\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0x3ff00000\n v_mov_b32 v120, 0\n v_mov_b32 v121, 0x3ff00000\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0x3ff00000\n" ::: CLOB);
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (T == 0) asm volatile(R4("v_fma_f64 v[100:101], v[120:121], v[122:123], v[100:101]\n v_fma_f64 v[102:103], v[120:121], v[122:123], v[102:103]\n v_fma_f64 v[104:105], v[120:121], v[122:123], v[104:105]\n v_fma_f64 v[106:107], v[120:121], v[122:123], v[106:107]\n"
                                    "v_fma_f64 v[108:109], v[120:121], v[122:123], v[108:109]\n v_fma_f64 v[110:111], v[120:121], v[122:123], v[110:111]\n v_fma_f64 v[112:113], v[120:121], v[122:123], v[112:113]\n v_fma_f64 v[114:115], v[120:121], v[122:123], v[114:115]\n") ::: CLOB);
        if (T == 1) asm volatile(R8(R4("v_fma_f64 v[100:101], v[120:121], v[122:123], v[100:101]\n")) ::: CLOB);
        if (T == 2) asm volatile(R4("v_fmac_f64_dpp v[100:101], v[120:121], v[122:123] row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[102:103], v[120:121], v[122:123] row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[104:105], v[120:121], v[122:123] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[106:107], v[120:121], v[122:123] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                                    "v_fmac_f64_dpp v[108:109], v[120:121], v[122:123] row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[110:111], v[120:121], v[122:123] row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[112:113], v[120:121], v[122:123] row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[114:115], v[120:121], v[122:123] row_newbcast:8 row_mask:0xf bank_mask:0xf\n") ::: CLOB);
        if (T == 3) asm volatile(R4("v_cndmask_b32 v100, v120, v121, vcc\n v_cndmask_b32 v101, v120, v121, vcc\n v_cndmask_b32 v102, v120, v121, vcc\n v_cndmask_b32 v103, v120, v121, vcc\n v_cndmask_b32 v104, v120, v121, vcc\n v_cndmask_b32 v105, v120, v121, vcc\n v_cndmask_b32 v106, v120, v121, vcc\n v_cndmask_b32 v107, v120, v121, vcc\n") ::: CLOB, "vcc");
    }
    const long long t1 = __builtin_readcyclecounter();
    double r;
    asm volatile("v_add_f64 %0, v[100:101], v[114:115]" : "=v"(r) :: CLOB);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int T> static void run(const char *name, long long *cyc, double *sink)
{
    const int iters = 20000, blocks = 1024;
    for (int active : {64, 32, 16}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, cyc, sink, iters, active); (void)hipDeviceSynchronize(); }
        static long long h[1024];
        (void)hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < blocks; i++) mean += (double)h[i]; mean /= blocks;
        printf("%-44s EXEC = %2d lanes: %6.2f counter ticks per instruction of a wave\n", name, active, mean / ((double)iters * 32));
    }
}
int main()
{
    long long *cyc; double *sink; (void)hipMalloc(&cyc, 4096 * 8); (void)hipMalloc(&sink, 4096 * 64 * 8);
    for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(64), 0, 0, cyc, sink, 20000, 64);
    (void)hipDeviceSynchronize();
    run<0>("v_fma_f64, 8 accumulators", cyc, sink); run<1>("v_fma_f64, one accumulator (chain)", cyc, sink);
    run<2>("v_fmac_f64_dpp row_newbcast, 8 accumulators", cyc, sink); run<3>("v_cndmask_b32, 8 targets", cyc, sink);
    return 0;
}
