// HBM read bandwidth as a function of the contiguous chunk a wave fetches per instruction, for a access stream
// shaped like the QP kernel's: every wave walks its own sequence of chunks scattered over an 8 GiB buffer,
// `depth` loads in flight before it consumes them.  chunk = 512 B (dwordx2 per lane) vs 1024 B (dwordx4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int W> struct V;
template <> struct V<2> { typedef unsigned t __attribute__((ext_vector_type(2))); };
template <> struct V<4> { typedef unsigned t __attribute__((ext_vector_type(4))); };

template <int W, int DEPTH, bool SEQ>
__global__ void __launch_bounds__(64, 2) k(const unsigned *buf, unsigned long nchunks, int iters, unsigned *out)
{
    typedef typename V<W>::t vec;
    const unsigned lane = threadIdx.x;
    unsigned long s = (unsigned long)blockIdx.x * 0x9E3779B97F4A7C15ul + 12345;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        vec v[DEPTH];
        unsigned long base = 0;
        if (SEQ) { s = s * 6364136223846793005ul + 1442695040888963407ul; base = (s >> 20) % (nchunks - DEPTH); }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            unsigned long c;
            if (SEQ) c = base + d; // DEPTH consecutive chunks = one contiguous tile
            else { s = s * 6364136223846793005ul + 1442695040888963407ul; c = (s >> 20) % nchunks; }
            v[d] = *(const vec *)(buf + c * (64ul * W) + lane * W);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc += v[d].x;
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int W, int DEPTH, bool SEQ>
void run(const unsigned *buf, size_t bytes, unsigned *out, const char *tag)
{
    const int blocks = 16384, iters = 200;
    const unsigned long nchunks = bytes / (64ul * W * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<W, DEPTH, SEQ>), dim3(blocks), dim3(64), 0, 0, buf, nchunks, 20, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<W, DEPTH, SEQ>), dim3(blocks), dim3(64), 0, 0, buf, nchunks, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double gb = (double)blocks * iters * DEPTH * 64 * W * 4 / 1e9;
    printf("%-44s chunk %4d B depth %2d : %7.1f GB in %7.2f ms = %6.2f TB/s\n", tag, 64 * W * 4, DEPTH, gb, ms, gb / ms);
}
int main()
{
    const size_t bytes = 8ul << 30;
    unsigned *buf, *out;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&out, 16384 * 64 * 4);
    run<2, 16, false>(buf, bytes, out, "scattered chunks");
    run<4, 8, false>(buf, bytes, out, "scattered chunks");
    run<4, 16, false>(buf, bytes, out, "scattered chunks");
    run<2, 16, true>(buf, bytes, out, "16 consecutive chunks (one 8 KB tile)");
    run<4, 8, true>(buf, bytes, out, "8 consecutive chunks (one 8 KB tile)");
    run<2, 32, true>(buf, bytes, out, "32 consecutive chunks (16 KB tile)");
    return 0;
}
