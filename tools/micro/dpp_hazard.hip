// Which operands of v_fmac_f64_dpp are subject to the "VALU writes VGPR -> DPP reads it: 2 wait states" rule on gfx950?
// Each test writes one operand with a VALU instruction, issues the DPP FMA n instructions later (n = 0, 1, 2 independent VALU
// instructions in between) and compares with the value computed with the hazard padded out (s_nop 7).
#include <hip/hip_runtime.h>
#include <cstdio>

// out[0]: reference, out[1..3]: gap 0, 1, 2
template <int WHICH> __global__ void k(double *out)
{
    const double lanev = 1.0 + threadIdx.x;          // differs per lane
    double r[4];
#define BODY(GAP)                                                                                                   \
    {                                                                                                               \
        double acc = 100.0, b = lanev, a = 3.0, d0 = 1.0, d1 = 2.0;                                                 \
        asm volatile("s_nop 7\n"                                                                                    \
                     ".if " #GAP " == 9\n"                                                                          \
                     "  .if %5 == 0\n v_add_f64 %1, %1, %1\n .endif\n"                                              \
                     "  .if %5 == 1\n v_add_f64 %2, %2, %2\n .endif\n"                                              \
                     "  .if %5 == 2\n v_add_f64 %0, %0, %0\n .endif\n"                                              \
                     "  s_nop 7\n"                                                                                  \
                     ".else\n"                                                                                      \
                     "  .if %5 == 0\n v_add_f64 %1, %1, %1\n .endif\n"                                              \
                     "  .if %5 == 1\n v_add_f64 %2, %2, %2\n .endif\n"                                              \
                     "  .if %5 == 2\n v_add_f64 %0, %0, %0\n .endif\n"                                              \
                     "  .if " #GAP " >= 1\n v_add_f64 %3, %3, %3\n .endif\n"                                        \
                     "  .if " #GAP " >= 2\n v_add_f64 %4, %4, %4\n .endif\n"                                        \
                     ".endif\n"                                                                                     \
                     "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"                        \
                     "s_nop 7\n"                                                                                    \
                     : "+v"(acc), "+v"(b), "+v"(a), "+v"(d0), "+v"(d1) : "n"(WHICH));                               \
        r[(GAP) == 9 ? 0 : (GAP) + 1] = acc;                                                                        \
    }
    BODY(9) BODY(0) BODY(1) BODY(2)
    for (int i = 0; i < 4; i++) out[i * 64 + threadIdx.x] = r[i];
}

// a chain of four v_fmac_f64_dpp on ONE accumulator, back to back, against the same chain with the pipeline drained in between
__global__ void k_chain(double *out)
{
    double acc0 = 0.5, acc1 = 0.5, b = 1.0 + threadIdx.x, a0 = 3.0, a1 = 0.25 * threadIdx.x, a2 = -1.5, a3 = 7.0;
    asm volatile("s_nop 7\n"
                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %0, %2, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %0, %2, %6 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
                 "s_nop 7\n"
                 "v_fmac_f64_dpp %1, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n s_nop 7\n"
                 "v_fmac_f64_dpp %1, %2, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n s_nop 7\n"
                 "v_fmac_f64_dpp %1, %2, %5 row_newbcast:11 row_mask:0xf bank_mask:0xf\n s_nop 7\n"
                 "v_fmac_f64_dpp %1, %2, %6 row_newbcast:15 row_mask:0xf bank_mask:0xf\n s_nop 7\n"
                 : "+&v"(acc0), "+&v"(acc1) : "v"(b), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    out[threadIdx.x] = acc0; out[64 + threadIdx.x] = acc1;
}

int main()
{
    double *d, h[256]; (void)hipMalloc(&d, sizeof(h));
    const char *names[3] = {"DPP source (src0) written just before", "plain source (src1) written just before", "accumulator written just before"};
    for (int w = 0; w < 3; w++) {
        if (w == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d);
        if (w == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d);
        if (w == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-44s:", names[w]);
        for (int gap = 0; gap < 3; gap++) {
            int bad = 0; for (int l = 0; l < 64; l++) bad += h[(gap + 1) * 64 + l] != h[l];
            printf("  %d instruction(s) between: %s (%d lanes differ)", gap, bad ? "WRONG" : "ok", bad);
        }
        printf("   [lane 0 reference %.1f]\n", h[0]);
    }
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0; for (int l = 0; l < 64; l++) bad += h[l] != h[64 + l];
    printf("four dependent v_fmac_f64_dpp back to back on one accumulator: %s (%d lanes differ from the drained chain; lane 5: %.3f)\n", bad ? "WRONG" : "ok", bad, h[5]);
    return 0;
}
