// Round 6: what bounds the three-read + one-write streaming kernels (ReLU / dropout / MSE backward `dx += f(g, x)`, SGD) at HBM sizes?
// They sit at 5.3 - 5.5 TB/s where a copy reaches 6.3.  One f32 element-wise body, 1 GiB per tensor, variants of
// (a) stream count and in-place-ness, (b) loads in flight per lane (float4s per thread and trip, all loads issued before the first use),
// (c) walk order (grid-stride vs one contiguous span per block), (d) cache policy (plain / nt loads, plain / nt stores), (e) block size.
//   hipcc --offload-arch=gfx950 -O3 -o rmw_stream rmw_stream.hip && ./rmw_stream          -> one JSON line per variant
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ v4 ld(const v4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st(v4* p, v4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// READS = 1: d = x; 2: d = x + g; 3: d = d + (x > 0 ? g : 0 g) (in place, the ReLU backward); 4: o = d + (x > 0 ? g : 0 g) (same streams, NOT in place)
template <int READS, int U, bool SPAN, bool NTL, bool NTS>
__global__ void body(v4* __restrict__ d, const v4* __restrict__ g, const v4* __restrict__ x, v4* __restrict__ o, size_t n4) {
    const size_t T = (size_t)gridDim.x * blockDim.x;
    size_t i, end, step;
    if (SPAN) {  // one contiguous span of n4 / gridDim.x float4s per block, walked blockDim.x * U at a time
        const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
        i = blockIdx.x * per + threadIdx.x; end = (blockIdx.x + 1) * per < n4 ? (blockIdx.x + 1) * per : n4; step = blockDim.x;
    } else {
        i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; end = n4; step = T;
    }
    for (; i + (U - 1) * step < end; i += U * step) {
        v4 xv[U], gv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = ld<NTL>(x + i + u * step);
            if (READS >= 2) gv[u] = ld<NTL>(g + i + u * step);
            if (READS >= 3) dv[u] = ld<NTL>(d + i + u * step);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v4 r = xv[u];
            if (READS == 2) r = xv[u] + gv[u];
            if (READS >= 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = dv[u][e] + (xv[u][e] > 0.f ? gv[u][e] : 0.f * gv[u][e]);
            }
            st<NTS>((READS == 4 ? o : d) + i + u * step, r);
        }
    }
    for (; i < end; i += step) {  // remainder trips
        v4 r = ld<NTL>(x + i);
        if (READS == 2) r = r + ld<NTL>(g + i);
        if (READS >= 3) {
            const v4 gv = ld<NTL>(g + i), dv = ld<NTL>(d + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = dv[e] + (r[e] > 0.f ? gv[e] : 0.f * gv[e]);
        }
        st<NTS>((READS == 4 ? o : d) + i, r);
    }
}

static float *D, *G, *X, *O;
static size_t N4;
static hipEvent_t e0, e1;

template <int READS, int U, bool SPAN, bool NTL, bool NTS>
static void run(const char* name, int grid, int block) {
    auto launch = [&]() { hipLaunchKernelGGL((body<READS, U, SPAN, NTL, NTS>), dim3(grid), dim3(block), 0, 0, (v4*)D, (const v4*)G, (const v4*)X, (v4*)O, N4); };
    for (int r = 0; r < 20; ++r) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 40;
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int streams = READS == 1 ? 2 : READS == 2 ? 3 : 4;
    const double bytes = (double)reps * streams * N4 * 16;
    printf("{\"variant\": \"%s\", \"streams\": \"%d reads + 1 write%s\", \"floats4_in_flight_per_lane_and_stream\": %d, \"walk\": \"%s\", \"nt_loads\": %d, \"nt_stores\": %d, "
           "\"grid\": %d, \"block\": %d, \"ms\": %.4f, \"GBps\": %.0f, \"frac_of_8TBps\": %.4f}\n",
           name, streams - 1, READS == 3 ? " (in place)" : "", U, SPAN ? "span per block" : "grid stride", (int)NTL, (int)NTS, grid, block, ms / reps,
           bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const size_t n = (size_t)(argc > 1 ? atoll(argv[1]) : 256) << 20;  // floats per tensor: 256 Mi = 1 GiB
    N4 = n / 4;
    hipMalloc(&D, n * 4); hipMalloc(&G, n * 4); hipMalloc(&X, n * 4); hipMalloc(&O, n * 4);
    hipMemset(D, 0, n * 4); hipMemset(G, 0, n * 4); hipMemset(X, 0x3f, n * 4); hipMemset(O, 0, n * 4);
    hipEventCreate(&e0); hipEventCreate(&e1);
    if (argc > 2) {  // second sweep: the span walk over grids, loads in flight and stream counts (what session d's first table pointed at)
        run<3, 4, true, true, true>("relu_bwd span U4 b256", 512, 256);
        run<3, 4, true, true, true>("relu_bwd span U4 b256", 1024, 256);
        run<3, 4, true, true, true>("relu_bwd span U4 b256", 2048, 256);
        run<3, 4, true, true, true>("relu_bwd span U4 b256", 4096, 256);
        run<3, 4, true, true, true>("relu_bwd span U4 b256", 8192, 256);
        run<3, 2, true, true, true>("relu_bwd span U2 b256", 2048, 256);
        run<3, 2, true, true, true>("relu_bwd span U2 b256", 4096, 256);
        run<3, 8, true, true, true>("relu_bwd span U8 b256", 1024, 256);
        run<3, 8, true, true, true>("relu_bwd span U8 b256", 2048, 256);
        run<3, 4, true, true, true>("relu_bwd span U4 b512", 1024, 512);
        run<3, 4, true, true, true>("relu_bwd span U4 b512", 2048, 512);
        run<3, 2, true, true, true>("relu_bwd span U2 b512", 2048, 512);
        run<3, 4, true, false, true>("relu_bwd span U4 b256 plain loads", 2048, 256);
        run<3, 4, true, true, false>("relu_bwd span U4 b256 plain stores", 2048, 256);
        run<3, 4, true, false, false>("relu_bwd span U4 b256 plain both", 2048, 256);
        run<4, 4, true, true, true>("relu_bwd out of place span U4 b256", 2048, 256);
        run<1, 4, true, true, true>("copy span U4 b256", 2048, 256);
        run<1, 8, true, true, true>("copy span U8 b256", 2048, 256);
        run<1, 4, true, true, true>("copy span U4 b256", 1024, 256);
        run<1, 4, true, false, true>("copy span U4 b256 plain loads", 2048, 256);
        run<2, 4, true, true, true>("add span U4 b256", 2048, 256);
        run<2, 4, true, true, true>("add span U4 b256", 1024, 256);
        run<2, 8, true, true, true>("add span U8 b256", 2048, 256);
        return 0;
    }
    // (a) stream count, the library's form: grid stride, 8192 blocks of 256, one float4 per lane and trip, nt loads + nt stores
    run<1, 1, false, true, true>("copy", 8192, 256);
    run<2, 1, false, true, true>("add", 8192, 256);
    run<3, 1, false, true, true>("relu_bwd (library form)", 8192, 256);
    run<4, 1, false, true, true>("relu_bwd, out of place", 8192, 256);
    // (b) loads in flight
    run<3, 2, false, true, true>("relu_bwd U2", 8192, 256);
    run<3, 4, false, true, true>("relu_bwd U4", 8192, 256);
    run<3, 4, false, true, true>("relu_bwd U4 grid 2048", 2048, 256);
    run<3, 8, false, true, true>("relu_bwd U8 grid 2048", 2048, 256);
    run<1, 4, false, true, true>("copy U4 grid 2048", 2048, 256);
    run<2, 4, false, true, true>("add U4 grid 2048", 2048, 256);
    // (c) walk order
    run<3, 1, true, true, true>("relu_bwd span", 8192, 256);
    run<3, 4, true, true, true>("relu_bwd span U4", 2048, 256);
    run<3, 4, true, true, true>("relu_bwd span U4 grid 1024", 1024, 256);
    // (d) cache policy
    run<3, 1, false, false, true>("relu_bwd plain loads", 8192, 256);
    run<3, 1, false, true, false>("relu_bwd plain stores", 8192, 256);
    run<3, 1, false, false, false>("relu_bwd plain both", 8192, 256);
    run<3, 4, false, false, true>("relu_bwd U4 plain loads", 2048, 256);
    // (e) block size / grid
    run<3, 1, false, true, true>("relu_bwd block 512", 8192, 512);
    run<3, 1, false, true, true>("relu_bwd block 1024", 4096, 1024);
    run<3, 1, false, true, true>("relu_bwd grid 2048", 2048, 256);
    run<3, 1, false, true, true>("relu_bwd grid 16384", 16384, 256);
    run<3, 2, false, true, true>("relu_bwd U2 block 512 grid 4096", 4096, 512);
    return 0;
}
