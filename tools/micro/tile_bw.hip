// The QP kernel's memory pattern without its arithmetic: every wave walks tiles of NPT planes (512 B each); per
// tile it reads NRD planes (one batch, ascending) and writes NWR planes.  How much of the 6.3 TB/s that pure
// contiguous reads reach survives the writes and the partial coverage of the tile?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
// EVERY: the NWR planes are written only in every EVERY-th tile (same plane count per write, rarer write bursts)
template <int NPT, int NRD, int NWR, bool SPLIT, int EVERY = 1, bool INDEP = false>
__global__ void __launch_bounds__(64, 2) k(u2 *buf, int ntiles, int iters, unsigned *out)
{
    const unsigned lane = threadIdx.x;
    unsigned acc = 0;
    // tiles of one wave are NBLK tiles apart (as stages are in the solver: [stage][wave][plane][lane])
    const long stride = (long)gridDim.x * NPT * 64;
    u2 *t = buf + (long)blockIdx.x * NPT * 64 + lane;
    for (int it = 0; it < iters; it++) {
        u2 *p = t + (long)(it % ntiles) * stride;
        u2 v[NRD];
#pragma unroll
        for (int d = 0; d < NRD; d++) v[d] = p[(SPLIT ? (d * NPT) / NRD : d) * 64];
#pragma unroll
        for (int d = 0; d < NRD; d++) acc += v[d].x;
        u2 w; w.x = INDEP ? lane : acc; w.y = it; // INDEP: the stored value does not depend on this tile's loads
        if (it % EVERY == EVERY - 1) {
#pragma unroll
            for (int d = 0; d < NWR; d++) p[(NPT - 1 - d) * 64] = w;
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}
template <int NPT, int NRD, int NWR, bool SPLIT, int EVERY = 1, bool INDEP = false>
void run(u2 *buf, unsigned *out, const char *tag)
{
    const int blocks = 16384, ntiles = 41, iters = 41 * 12;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NPT, NRD, NWR, SPLIT, EVERY, INDEP>), dim3(blocks), dim3(64), 0, 0, buf, ntiles, 41, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NPT, NRD, NWR, SPLIT, EVERY, INDEP>), dim3(blocks), dim3(64), 0, 0, buf, ntiles, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double gb = (double)blocks * iters * (NRD + (double)NWR / EVERY) * 512 / 1e9;
    printf("%-52s tile %2d planes, read %2d write %2d : %6.1f GB in %6.2f ms = %5.2f TB/s\n", tag, NPT, NRD, NWR, gb, ms, gb / ms);
}
int main()
{
    u2 *buf; unsigned *out;
    const size_t bytes = (size_t)16384 * 41 * 28 * 512;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&out, 16384 * 64 * 4);
    run<27, 27, 0, false>(buf, out, "whole tile, reads only");
    run<27, 20, 0, false>(buf, out, "first 20 planes of the tile, reads only");
    run<27, 20, 0, true>(buf, out, "20 planes spread over the tile, reads only");
    run<27, 20, 4, false>(buf, out, "first 20 planes read, 4 planes written");
    run<27, 20, 1, false>(buf, out, "first 20 planes read, 1 plane written");
    run<27, 12, 0, false>(buf, out, "first 12 planes, reads only");
    run<27, 20, 4, false, 4>(buf, out, "20 planes read, 4 planes written every 4th tile");
    run<27, 20, 8, false, 8>(buf, out, "20 planes read, 8 planes written every 8th tile");
    run<27, 20, 4, false, 1, true>(buf, out, "20 planes read, 4 written (value independent of loads)");
    run<27, 20, 1, false, 1, true>(buf, out, "20 planes read, 1 written (value independent of loads)");
    return 0;
}
