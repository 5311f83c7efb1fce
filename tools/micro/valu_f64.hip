// Issue cost of the instructions the QP kernel is made of (gfx950), measured with 1 and with 2 waves per SIMD: s_memtime cycles of a
// wave per instruction of that wave.  With 1 wave the figure is the dependent-issue cost, with 2 waves (both streaming the same mix)
// twice the SIMD's issue cost per instruction if the pipe is saturated.  Evidence for DESIGN.md section 4 (what bounds usv_qp_rti).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)

enum { FMA_INDEP, FMA_CHAIN, FMAC_INDEP, MUL_INDEP, ADD_INDEP, MOVDPP64, MOVDPP32x2, BCAST_FMA, BCAST_FMA_CHAIN, FMA_DPP, MOV64, CNDMASK, DSREAD, DSWRITE_READ, RCP, RSQ, NT };
static const char *NAMES[NT] = {
    "v_fma_f64, 8 accumulators", "v_fma_f64, one accumulator (chain)", "v_fmac_f64, 8 accumulators", "v_mul_f64, 8 targets", "v_add_f64, 8 targets",
    "v_mov_b64 row_newbcast, 8 targets", "2 x v_mov_b32 row_newbcast, 8 targets (per pair)", "v_mov_b64 row_newbcast + v_fma_f64, 4 accumulators (per pair)",
    "v_mov_b64 row_newbcast + v_fma_f64, one accumulator (per pair)", "v_fma_f64 with row_newbcast operand (DPP), 8 accumulators", "v_mov_b64 (no DPP), 8 targets",
    "v_cndmask_b32, 8 targets", "ds_read_b64, 8 in flight (per read)", "ds_write_b64 + ds_read_b64 pair", "v_rcp_f64, 8 targets", "v_rsq_f64, 8 targets"};
static const int PER_BODY[NT] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32};

#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139"

template <int T> __global__ void __launch_bounds__(64) k(long long *cyc, double *sink, int iters)
{
    __shared__ double lds[64 * 8];
    lds[threadIdx.x] = 1.0;
    const unsigned la = threadIdx.x * 8;
    asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0x3ff00000\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0x3ff00000\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0x3ff00000\n"
                 "v_mov_b32 v106, 0\n v_mov_b32 v107, 0x3ff00000\n v_mov_b32 v108, 0\n v_mov_b32 v109, 0x3ff00000\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0x3ff00000\n"
                 "v_mov_b32 v112, 0\n v_mov_b32 v113, 0x3ff00000\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0x3ff00000\n"
                 "v_mov_b32 v120, 0\n v_mov_b32 v121, 0x3ff00000\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0x3ff00000\n v_mov_b32 v124, 0\n v_mov_b32 v125, 0x3ff00000\n"
                 "v_mov_b32 v126, 0\n v_mov_b32 v127, 0x3ff00000\n v_mov_b32 v128, 0\n v_mov_b32 v129, 0x3ff00000\n v_mov_b32 v130, 0\n v_mov_b32 v131, 0x3ff00000\n"
                 "v_mov_b32 v132, 0\n v_mov_b32 v133, 0x3ff00000\n v_mov_b32 v134, 0\n v_mov_b32 v135, 0x3ff00000\n v_mov_b32 v136, 0\n v_mov_b32 v137, 0x3ff00000\n" ::: CLOB);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (T == FMA_INDEP) asm volatile(R4("v_fma_f64 v[100:101], v[120:121], v[122:123], v[100:101]\n v_fma_f64 v[102:103], v[120:121], v[122:123], v[102:103]\n v_fma_f64 v[104:105], v[120:121], v[122:123], v[104:105]\n v_fma_f64 v[106:107], v[120:121], v[122:123], v[106:107]\n"
                                            "v_fma_f64 v[108:109], v[120:121], v[122:123], v[108:109]\n v_fma_f64 v[110:111], v[120:121], v[122:123], v[110:111]\n v_fma_f64 v[112:113], v[120:121], v[122:123], v[112:113]\n v_fma_f64 v[114:115], v[120:121], v[122:123], v[114:115]\n") ::: CLOB);
        if (T == FMA_CHAIN) asm volatile(R8(R4("v_fma_f64 v[100:101], v[120:121], v[122:123], v[100:101]\n")) ::: CLOB);
        if (T == FMAC_INDEP) asm volatile(R4("v_fmac_f64 v[100:101], v[120:121], v[122:123]\n v_fmac_f64 v[102:103], v[120:121], v[122:123]\n v_fmac_f64 v[104:105], v[120:121], v[122:123]\n v_fmac_f64 v[106:107], v[120:121], v[122:123]\n"
                                             "v_fmac_f64 v[108:109], v[120:121], v[122:123]\n v_fmac_f64 v[110:111], v[120:121], v[122:123]\n v_fmac_f64 v[112:113], v[120:121], v[122:123]\n v_fmac_f64 v[114:115], v[120:121], v[122:123]\n") ::: CLOB);
        if (T == MUL_INDEP) asm volatile(R4("v_mul_f64 v[100:101], v[120:121], v[122:123]\n v_mul_f64 v[102:103], v[120:121], v[122:123]\n v_mul_f64 v[104:105], v[120:121], v[122:123]\n v_mul_f64 v[106:107], v[120:121], v[122:123]\n"
                                            "v_mul_f64 v[108:109], v[120:121], v[122:123]\n v_mul_f64 v[110:111], v[120:121], v[122:123]\n v_mul_f64 v[112:113], v[120:121], v[122:123]\n v_mul_f64 v[114:115], v[120:121], v[122:123]\n") ::: CLOB);
        if (T == ADD_INDEP) asm volatile(R4("v_add_f64 v[100:101], v[120:121], v[122:123]\n v_add_f64 v[102:103], v[120:121], v[122:123]\n v_add_f64 v[104:105], v[120:121], v[122:123]\n v_add_f64 v[106:107], v[120:121], v[122:123]\n"
                                            "v_add_f64 v[108:109], v[120:121], v[122:123]\n v_add_f64 v[110:111], v[120:121], v[122:123]\n v_add_f64 v[112:113], v[120:121], v[122:123]\n v_add_f64 v[114:115], v[120:121], v[122:123]\n") ::: CLOB);
        if (T == MOVDPP64) asm volatile(R4("v_mov_b64_dpp v[100:101], v[120:121] row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[102:103], v[120:121] row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[104:105], v[120:121] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[106:107], v[120:121] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                                           "v_mov_b64_dpp v[108:109], v[120:121] row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[110:111], v[120:121] row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[112:113], v[120:121] row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[114:115], v[120:121] row_newbcast:8 row_mask:0xf bank_mask:0xf\n") ::: CLOB);
        if (T == MOVDPP32x2) asm volatile(R4("v_mov_b32_dpp v100, v120 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v101, v121 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v102, v120 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v103, v121 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                                             "v_mov_b32_dpp v104, v120 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v105, v121 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v106, v120 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v107, v121 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                                             "v_mov_b32_dpp v108, v120 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v109, v121 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v110, v120 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v111, v121 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                                             "v_mov_b32_dpp v112, v120 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v113, v121 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v114, v120 row_newbcast:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v115, v121 row_newbcast:8 row_mask:0xf bank_mask:0xf\n") ::: CLOB);
        if (T == BCAST_FMA) asm volatile(R8("v_mov_b64_dpp v[130:131], v[120:121] row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fma_f64 v[100:101], v[130:131], v[122:123], v[100:101]\n"
                                            "v_mov_b64_dpp v[132:133], v[120:121] row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fma_f64 v[102:103], v[132:133], v[122:123], v[102:103]\n"
                                            "v_mov_b64_dpp v[134:135], v[120:121] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fma_f64 v[104:105], v[134:135], v[122:123], v[104:105]\n"
                                            "v_mov_b64_dpp v[136:137], v[120:121] row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fma_f64 v[106:107], v[136:137], v[122:123], v[106:107]\n") ::: CLOB);
        if (T == BCAST_FMA_CHAIN) asm volatile(R8(R4("v_mov_b64_dpp v[130:131], v[100:101] row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fma_f64 v[100:101], v[130:131], v[122:123], v[100:101]\n")) ::: CLOB);
        if (T == FMA_DPP) asm volatile(R4("v_fmac_f64_dpp v[100:101], v[120:121], v[122:123] row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[102:103], v[120:121], v[122:123] row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[104:105], v[120:121], v[122:123] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[106:107], v[120:121], v[122:123] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                                          "v_fmac_f64_dpp v[108:109], v[120:121], v[122:123] row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[110:111], v[120:121], v[122:123] row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[112:113], v[120:121], v[122:123] row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[114:115], v[120:121], v[122:123] row_newbcast:8 row_mask:0xf bank_mask:0xf\n") ::: CLOB);
        if (T == MOV64) asm volatile(R4("v_mov_b64 v[100:101], v[120:121]\n v_mov_b64 v[102:103], v[120:121]\n v_mov_b64 v[104:105], v[120:121]\n v_mov_b64 v[106:107], v[120:121]\n v_mov_b64 v[108:109], v[120:121]\n v_mov_b64 v[110:111], v[120:121]\n v_mov_b64 v[112:113], v[120:121]\n v_mov_b64 v[114:115], v[120:121]\n") ::: CLOB);
        if (T == CNDMASK) asm volatile(R4("v_cndmask_b32 v100, v120, v121, vcc\n v_cndmask_b32 v101, v120, v121, vcc\n v_cndmask_b32 v102, v120, v121, vcc\n v_cndmask_b32 v103, v120, v121, vcc\n v_cndmask_b32 v104, v120, v121, vcc\n v_cndmask_b32 v105, v120, v121, vcc\n v_cndmask_b32 v106, v120, v121, vcc\n v_cndmask_b32 v107, v120, v121, vcc\n") ::: CLOB, "vcc");
        if (T == DSREAD) asm volatile(R4("ds_read_b64 v[100:101], %0\n ds_read_b64 v[102:103], %0 offset:512\n ds_read_b64 v[104:105], %0 offset:1024\n ds_read_b64 v[106:107], %0 offset:1536\n ds_read_b64 v[108:109], %0 offset:2048\n ds_read_b64 v[110:111], %0 offset:2560\n ds_read_b64 v[112:113], %0 offset:3072\n ds_read_b64 v[114:115], %0 offset:3584\n s_waitcnt lgkmcnt(4)\n") "s_waitcnt lgkmcnt(0)\n" :: "v"(la) : CLOB, "memory");
        if (T == DSWRITE_READ) asm volatile(R8(R4("ds_write_b64 %0, v[120:121]\n ds_read_b64 v[100:101], %0 offset:512\n") "s_waitcnt lgkmcnt(0)\n") :: "v"(la) : CLOB, "memory");
        if (T == RCP) asm volatile(R4("v_rcp_f64 v[100:101], v[120:121]\n v_rcp_f64 v[102:103], v[120:121]\n v_rcp_f64 v[104:105], v[120:121]\n v_rcp_f64 v[106:107], v[120:121]\n v_rcp_f64 v[108:109], v[120:121]\n v_rcp_f64 v[110:111], v[120:121]\n v_rcp_f64 v[112:113], v[120:121]\n v_rcp_f64 v[114:115], v[120:121]\n") ::: CLOB);
        if (T == RSQ) asm volatile(R4("v_rsq_f64 v[100:101], v[120:121]\n v_rsq_f64 v[102:103], v[120:121]\n v_rsq_f64 v[104:105], v[120:121]\n v_rsq_f64 v[106:107], v[120:121]\n v_rsq_f64 v[108:109], v[120:121]\n v_rsq_f64 v[110:111], v[120:121]\n v_rsq_f64 v[112:113], v[120:121]\n v_rsq_f64 v[114:115], v[120:121]\n") ::: CLOB);
    }
    const long long t1 = __builtin_readcyclecounter();
    double r;
    asm volatile("v_add_f64 %0, v[100:101], v[114:115]" : "=v"(r) :: CLOB);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = r + lds[threadIdx.x];
}

template <int T> static void run(long long *cyc, double *sink, hipEvent_t e0, hipEvent_t e1)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 2; wps++) {
        const int blocks = 1024 * wps;      // 64-thread blocks: 1024 = one wave per SIMD, 2048 = two
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, cyc, sink, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        static long long h[4096];
        (void)hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < blocks; i++) mean += (double)h[i]; mean /= blocks;
        const double n = (double)iters * PER_BODY[T];
        printf("%-68s %d wave/SIMD: %7.2f counter ticks per instr of a wave, %6.3f ns wall per instr per SIMD\n", NAMES[T], wps, mean / n, ms * 1e6 / n / wps);
    }
}

int main()
{
    long long *cyc; double *sink; (void)hipMalloc(&cyc, 4096 * 8); (void)hipMalloc(&sink, 4096 * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 30; w++) hipLaunchKernelGGL(k<FMA_INDEP>, dim3(2048), dim3(64), 0, 0, cyc, sink, 20000);   // ramp the clock
    (void)hipDeviceSynchronize();
    run<FMA_INDEP>(cyc, sink, e0, e1); run<FMA_CHAIN>(cyc, sink, e0, e1); run<FMAC_INDEP>(cyc, sink, e0, e1); run<MUL_INDEP>(cyc, sink, e0, e1); run<ADD_INDEP>(cyc, sink, e0, e1);
    run<MOVDPP64>(cyc, sink, e0, e1); run<MOVDPP32x2>(cyc, sink, e0, e1); run<BCAST_FMA>(cyc, sink, e0, e1); run<BCAST_FMA_CHAIN>(cyc, sink, e0, e1); run<FMA_DPP>(cyc, sink, e0, e1);
    run<MOV64>(cyc, sink, e0, e1); run<CNDMASK>(cyc, sink, e0, e1); run<DSREAD>(cyc, sink, e0, e1); run<DSWRITE_READ>(cyc, sink, e0, e1); run<RCP>(cyc, sink, e0, e1); run<RSQ>(cyc, sink, e0, e1);
    return 0;
}
