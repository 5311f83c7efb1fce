// Does the cache policy of the stores matter when a wave streams reads and writes through a tile?
// (raw buffer stores with aux = 0 | 1 (sc0) | 2 (nt) | 3, loads with aux 0 | 2)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
template <int NPT, int NRD, int NWR, int SAUX, int LAUX>
__global__ void __launch_bounds__(64, 2) k(u2 *buf, int ntiles, int iters, unsigned *out)
{
    const unsigned lane = threadIdx.x;
    unsigned acc = 0;
    const long stride = (long)gridDim.x * NPT * 64;
    for (int it = 0; it < iters; it++) {
        u2 *tile = buf + (long)blockIdx.x * NPT * 64 + (long)(it % ntiles) * stride;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(tile, 0, NPT * 512, 0x00020000);
        u2 v[NRD];
#pragma unroll
        for (int d = 0; d < NRD; d++) v[d] = __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, d * 512, LAUX);
#pragma unroll
        for (int d = 0; d < NRD; d++) acc += v[d].x;
        u2 w; w.x = acc; w.y = it;
#pragma unroll
        for (int d = 0; d < NWR; d++) __builtin_amdgcn_raw_buffer_store_b64(w, r, lane * 8, (NPT - 1 - d) * 512, SAUX);
    }
    out[blockIdx.x * 64 + lane] = acc;
}
template <int NWR, int SAUX, int LAUX>
void run(u2 *buf, unsigned *out)
{
    const int blocks = 16384, ntiles = 41, iters = 41 * 12;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<27, 20, NWR, SAUX, LAUX>), dim3(blocks), dim3(64), 0, 0, buf, ntiles, 41, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<27, 20, NWR, SAUX, LAUX>), dim3(blocks), dim3(64), 0, 0, buf, ntiles, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double gb = (double)blocks * iters * (20 + NWR) * 512 / 1e9;
    printf("read 20 write %d  store aux %d load aux %d : %6.1f GB in %6.2f ms = %5.2f TB/s\n", NWR, SAUX, LAUX, gb, ms, gb / ms);
}
int main()
{
    u2 *buf; unsigned *out;
    hipMalloc(&buf, (size_t)16384 * 41 * 28 * 512); hipMemset(buf, 1, (size_t)16384 * 41 * 28 * 512); hipMalloc(&out, 16384 * 64 * 4);
    run<0, 0, 0>(buf, out); run<0, 0, 2>(buf, out);
    run<4, 0, 0>(buf, out); run<4, 1, 0>(buf, out); run<4, 2, 0>(buf, out); run<4, 3, 0>(buf, out); run<4, 2, 2>(buf, out);
    run<1, 0, 0>(buf, out); run<1, 2, 0>(buf, out);
    return 0;
}
