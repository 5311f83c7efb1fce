// Issue cost of scalar / control instructions inside a VALU stream (gfx950): a wave's instructions issue in order, so do the
// s_mov / s_and / s_waitcnt / s_nop / branches of the QP kernel's sweeps cost the wave a VALU-sized slot, or do they ride along?
// Each variant is 32 x (v_fma_f64 + X); reported: ns per (fma + X) pair per SIMD at 1 and 2 waves per SIMD, next to the fma alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(x) x x x x x x x x
#define R4(x) x x x x
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v120","v121","v122","v123","s40","s41","s42","s43","scc"
#define FMA4(X) "v_fma_f64 v[100:101], v[120:121], v[122:123], v[100:101]\n" X "v_fma_f64 v[102:103], v[120:121], v[122:123], v[102:103]\n" X "v_fma_f64 v[104:105], v[120:121], v[122:123], v[104:105]\n" X "v_fma_f64 v[106:107], v[120:121], v[122:123], v[106:107]\n" X
enum { NONE, SMOV, SAND64, SWAIT, SNOP0, SNOP1, BR_NOT_TAKEN, BR_TAKEN, TWO_SALU, NT };
static const char *NAMES[NT] = {"v_fma_f64 alone", "+ s_mov_b32", "+ s_and_b64", "+ s_waitcnt (nothing outstanding)", "+ s_nop 0", "+ s_nop 1", "+ s_cbranch_scc1 (not taken)", "+ s_branch to the next instruction (taken)", "+ 2 x s_and_b64"};
template <int T> __global__ void __launch_bounds__(64) k(double *sink, int iters)
{
    asm volatile("v_mov_b32 v120, 0\n v_mov_b32 v121, 0x3ff00000\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0x3ff00000\n s_mov_b64 s[40:41], -1\n s_mov_b64 s[42:43], -1\n s_cmp_eq_u32 0, 1\n"
                 "v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n" ::: CLOB);
    for (int i = 0; i < iters; i++) {
        if (T == NONE) asm volatile(R8(FMA4("")) ::: CLOB);
        if (T == SMOV) asm volatile(R8(FMA4("s_mov_b32 s40, s42\n")) ::: CLOB);
        if (T == SAND64) asm volatile(R8(FMA4("s_and_b64 s[40:41], s[42:43], s[40:41]\n")) ::: CLOB);
        if (T == SWAIT) asm volatile(R8(FMA4("s_waitcnt vmcnt(0) lgkmcnt(0)\n")) ::: CLOB);
        if (T == SNOP0) asm volatile(R8(FMA4("s_nop 0\n")) ::: CLOB);
        if (T == SNOP1) asm volatile(R8(FMA4("s_nop 1\n")) ::: CLOB);
        if (T == BR_NOT_TAKEN) { asm volatile("s_cmp_eq_u32 0, 1\n" R8(FMA4("s_cbranch_scc1 1f\n")) "1:\n" ::: CLOB); }
        if (T == BR_TAKEN) asm volatile(R8(FMA4("s_branch 2f\n2:\n")) ::: CLOB);
        if (T == TWO_SALU) asm volatile(R8(FMA4("s_and_b64 s[40:41], s[42:43], s[40:41]\n s_and_b64 s[42:43], s[42:43], s[40:41]\n")) ::: CLOB);
    }
    double r;
    asm volatile("v_add_f64 %0, v[100:101], v[106:107]" : "=v"(r) :: CLOB);
    sink[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int T> static void run(double *sink, hipEvent_t e0, hipEvent_t e1)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 2; wps++) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<T>, dim3(1024 * wps), dim3(64), 0, 0, sink, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-46s %d wave/SIMD: %6.3f ns per pair per SIMD\n", NAMES[T], wps, ms * 1e6 / ((double)iters * 32) / wps);
    }
}
int main()
{
    double *sink; (void)hipMalloc(&sink, 4096 * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 30; w++) hipLaunchKernelGGL(k<NONE>, dim3(2048), dim3(64), 0, 0, sink, 20000);
    (void)hipDeviceSynchronize();
    run<NONE>(sink, e0, e1); run<SMOV>(sink, e0, e1); run<SAND64>(sink, e0, e1); run<TWO_SALU>(sink, e0, e1); run<SWAIT>(sink, e0, e1); run<SNOP0>(sink, e0, e1); run<SNOP1>(sink, e0, e1);
    run<BR_NOT_TAKEN>(sink, e0, e1); run<BR_TAKEN>(sink, e0, e1);
    return 0;
}
