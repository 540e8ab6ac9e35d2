// Throughput of 16-byte global loads at 16-byte-aligned vs merely 4-byte-aligned addresses (the convolution kernels'
// gathers are the latter), from an L2 / Infinity-Cache resident buffer, so that the vector memory pipe - not HBM - is
// what is measured.   hipcc --offload-arch=gfx950 -O3 -o unaligned_load unaligned_load.hip && ./unaligned_load
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int ITER>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, float* __restrict__ out, int off, size_t n4) {
    // each wave reads 64 consecutive quads (+ off floats), ITER times at a stride of one wave-row; rows re-used across blocks
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
        const size_t q = (i + (size_t)it * gridDim.x * blockDim.x) % n4;
        const f32x4u v = *reinterpret_cast<const f32x4u*>(x + q * 4 + off);
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[i] = s;
}
int main() {
    const size_t n = 8u << 20;  // 32 MB: fits the 32 MB of aggregate L2 / sits in the Infinity Cache
    float *x, *o;
    hipMalloc(&x, (n + 16) * 4); hipMalloc(&o, 1 << 24);
    hipMemset(x, 0, (n + 16) * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 2048, ITER = 256;
    for (int off = 0; off < 4; ++off) {
        k<ITER><<<grid, 256>>>(x, o, off, n / 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 5; ++r) k<ITER><<<grid, 256>>>(x, o, off, n / 4);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = 5.0 * grid * 256.0 * ITER * 16;
        printf("{\"offset_floats\": %d, \"GBps\": %.0f, \"ms\": %.3f}\n", off, bytes / ms / 1e6, ms / 5);
    }
    return 0;
}
