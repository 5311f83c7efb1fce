// Issue cost of the 32-bit VALU instructions a 64-bit select can be built from (gfx950), next to v_fma_f64: v_cndmask_b32 with the
// mask in VCC / in an SGPR pair, v_bfi_b32 / v_and_b32 with the mask in a VGPR, plain moves.  ns per instruction per SIMD with 1 and
// 2 waves per SIMD (see valu_f64.hip for the method).
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v120","v121","v122","v123","s40","s41","s42","s43"
#define EIGHT(OP, TAIL) OP " v100, " TAIL "\n" OP " v101, " TAIL "\n" OP " v102, " TAIL "\n" OP " v103, " TAIL "\n" OP " v104, " TAIL "\n" OP " v105, " TAIL "\n" OP " v106, " TAIL "\n" OP " v107, " TAIL "\n"
enum { CND_VCC, CND_SGPR, CND_SGPR_ALT, BFI, AND, MOV32, ADD32, FMA64, CND_CONST0, CND_MIX_FMA, BFI_MIX_FMA, NT };
static const char *NAMES[NT] = {"v_cndmask_b32 (mask in vcc)", "v_cndmask_b32_e64 (mask in s[40:41])", "v_cndmask_b32_e64 alternating s[40:41] / s[42:43]", "v_bfi_b32 (mask in a VGPR)",
    "v_and_b32", "v_mov_b32", "v_add_u32", "v_fma_f64", "v_cndmask_b32_e64 v, 0, v, s[40:41]", "2 x v_cndmask_b32_e64 + 1 x v_fma_f64 (per group of 3)", "2 x v_bfi_b32 + 1 x v_fma_f64 (per group of 3)"};
template <int T> __global__ void __launch_bounds__(64) k(double *sink, int iters)
{
    asm volatile("v_mov_b32 v120, 1\n v_mov_b32 v121, 2\n v_mov_b32 v122, -1\n v_mov_b32 v123, 0x3ff00000\n s_mov_b64 s[40:41], 0x5555\n s_mov_b64 s[42:43], 0x3333\n s_mov_b64 vcc, 0x5555\n"
                 "v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n" ::: CLOB, "vcc");
    for (int i = 0; i < iters; i++) {
        if (T == CND_VCC) asm volatile(R4(EIGHT("v_cndmask_b32", "v120, v121, vcc")) ::: CLOB, "vcc");
        if (T == CND_SGPR) asm volatile(R4(EIGHT("v_cndmask_b32_e64", "v120, v121, s[40:41]")) ::: CLOB);
        if (T == CND_SGPR_ALT) asm volatile(R4("v_cndmask_b32_e64 v100, v120, v121, s[40:41]\n v_cndmask_b32_e64 v101, v120, v121, s[42:43]\n v_cndmask_b32_e64 v102, v120, v121, s[40:41]\n v_cndmask_b32_e64 v103, v120, v121, s[42:43]\n"
                                               "v_cndmask_b32_e64 v104, v120, v121, s[40:41]\n v_cndmask_b32_e64 v105, v120, v121, s[42:43]\n v_cndmask_b32_e64 v106, v120, v121, s[40:41]\n v_cndmask_b32_e64 v107, v120, v121, s[42:43]\n") ::: CLOB);
        if (T == BFI) asm volatile(R4(EIGHT("v_bfi_b32", "v122, v120, v121")) ::: CLOB);
        if (T == AND) asm volatile(R4(EIGHT("v_and_b32", "v122, v120")) ::: CLOB);
        if (T == MOV32) asm volatile(R4(EIGHT("v_mov_b32", "v120")) ::: CLOB);
        if (T == ADD32) asm volatile(R4(EIGHT("v_add_u32", "v120, v121")) ::: CLOB);
        if (T == FMA64) asm volatile(R4("v_fma_f64 v[100:101], v[120:121], v[122:123], v[100:101]\n v_fma_f64 v[102:103], v[120:121], v[122:123], v[102:103]\n v_fma_f64 v[104:105], v[120:121], v[122:123], v[104:105]\n v_fma_f64 v[106:107], v[120:121], v[122:123], v[106:107]\n"
                                        "v_fma_f64 v[100:101], v[120:121], v[122:123], v[100:101]\n v_fma_f64 v[102:103], v[120:121], v[122:123], v[102:103]\n v_fma_f64 v[104:105], v[120:121], v[122:123], v[104:105]\n v_fma_f64 v[106:107], v[120:121], v[122:123], v[106:107]\n") ::: CLOB);
        if (T == CND_CONST0) asm volatile(R4(EIGHT("v_cndmask_b32_e64", "0, v121, s[40:41]")) ::: CLOB);
        if (T == CND_MIX_FMA) asm volatile(R4("v_cndmask_b32_e64 v100, v120, v121, s[40:41]\n v_cndmask_b32_e64 v101, v120, v121, s[40:41]\n v_fma_f64 v[102:103], v[120:121], v[122:123], v[102:103]\n"
                                              "v_cndmask_b32_e64 v104, v120, v121, s[42:43]\n v_cndmask_b32_e64 v105, v120, v121, s[42:43]\n v_fma_f64 v[106:107], v[120:121], v[122:123], v[106:107]\n") ::: CLOB);
        if (T == BFI_MIX_FMA) asm volatile(R4("v_bfi_b32 v100, v122, v120, v121\n v_bfi_b32 v101, v122, v120, v121\n v_fma_f64 v[102:103], v[120:121], v[122:123], v[102:103]\n"
                                              "v_bfi_b32 v104, v122, v120, v121\n v_bfi_b32 v105, v122, v120, v121\n v_fma_f64 v[106:107], v[120:121], v[122:123], v[106:107]\n") ::: CLOB);
    }
    double r;
    asm volatile("v_add_f64 %0, v[100:101], v[106:107]" : "=v"(r) :: CLOB);
    sink[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int T> static void run(double *sink, hipEvent_t e0, hipEvent_t e1)
{
    const int iters = 20000;
    const int per = (T == CND_MIX_FMA || T == BFI_MIX_FMA) ? 8 : 32;   // groups of 3 for the mixes
    for (int wps = 1; wps <= 2; wps++) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<T>, dim3(1024 * wps), dim3(64), 0, 0, sink, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-60s %d wave/SIMD: %6.3f ns per instruction%s per SIMD\n", NAMES[T], wps, ms * 1e6 / ((double)iters * per) / wps, per == 8 ? " group" : "");
    }
}
int main()
{
    double *sink; (void)hipMalloc(&sink, 4096 * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 30; w++) hipLaunchKernelGGL(k<FMA64>, dim3(2048), dim3(64), 0, 0, sink, 20000);
    (void)hipDeviceSynchronize();
    run<FMA64>(sink, e0, e1); run<CND_VCC>(sink, e0, e1); run<CND_SGPR>(sink, e0, e1); run<CND_SGPR_ALT>(sink, e0, e1); run<CND_CONST0>(sink, e0, e1); run<BFI>(sink, e0, e1); run<AND>(sink, e0, e1);
    run<MOV32>(sink, e0, e1); run<ADD32>(sink, e0, e1); run<CND_MIX_FMA>(sink, e0, e1); run<BFI_MIX_FMA>(sink, e0, e1);
    return 0;
}
