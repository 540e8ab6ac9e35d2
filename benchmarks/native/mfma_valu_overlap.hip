// Do f32 MFMA and plain VALU instructions of one SIMD overlap on gfx950?  One block of 8 waves per CU (LDS-limited), waves
// 0-3 and 4-7 land pairwise on the four SIMDs.  Three questions:
//   mode 0: waves 0-3 issue N dependent v_mfma_f32_32x32x2_f32, waves 4-7 idle                  -> MFMA time
//   mode 1: waves 0-3 idle, waves 4-7 issue M independent v_fma_f32 chains                       -> VALU time
//   mode 2: both at once (different waves of the same SIMD)                                      -> max or sum?
//   mode 3: ONE wave per SIMD interleaves 1 MFMA with V independent v_fma_f32 (same wave)        -> do they hide?
//   mode 4/5: like 1/2 with v_mad_u64_u32 (the Philox multiply) instead of v_fma_f32             -> its issue rate
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int per) {
    __shared__ float pad[36 * 1024];  // 144 KB: one block per CU
    const int w = threadIdx.x >> 6;
    if (threadIdx.x == 0) pad[0] = 0.f;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    unsigned long long u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    const bool mf = (MODE == 0 || MODE == 2 || MODE == 5) && w < 4;
    const bool va = (MODE == 1 || MODE == 2) && w >= 4;
    const bool vi = (MODE == 4 || MODE == 5) && w >= 4;
    if (mf) {
        for (int i = 0; i < n_mfma; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (va) {
        for (int i = 0; i < n_valu; ++i) {
            v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a); v3 = __builtin_fmaf(v3, b, a);
            v4 = __builtin_fmaf(v4, b, a); v5 = __builtin_fmaf(v5, b, a); v6 = __builtin_fmaf(v6, b, a); v7 = __builtin_fmaf(v7, b, a);
        }
    }
    if (vi) {
        for (int i = 0; i < n_valu; ++i) {
            u0 = (unsigned long long)(unsigned)u0 * 0xD2511F53u + (u0 >> 32); u1 = (unsigned long long)(unsigned)u1 * 0xCD9E8D57u + (u1 >> 32);
            u2 = (unsigned long long)(unsigned)u2 * 0xD2511F53u + (u2 >> 32); u3 = (unsigned long long)(unsigned)u3 * 0xCD9E8D57u + (u3 >> 32);
        }
    }
    if (MODE == 3 && w < 4) {
        for (int i = 0; i < n_mfma; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            for (int j = 0; j < per; ++j) {   // `per` x 8 independent fma after each MFMA (wave-uniform trip count)
                v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a); v3 = __builtin_fmaf(v3, b, a);
                v4 = __builtin_fmaf(v4, b, a); v5 = __builtin_fmaf(v5, b, a); v6 = __builtin_fmaf(v6, b, a); v7 = __builtin_fmaf(v7, b, a);
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(u0 + u1 + u2 + u3);
    for (int e = 0; e < 16; ++e) s += acc[e];
    if (s == 123.456f) out[threadIdx.x] = s + pad[0];
}

template <int MODE>
float run(float* o, int n_mfma, int n_valu, int per) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 512>>>(o, n_mfma, n_valu, per);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) k<MODE><<<256, 512>>>(o, n_mfma, n_valu, per);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main() {
    float* o; hipMalloc(&o, 1 << 16);
    const int NM = 20000;           // 20000 MFMAs x 64 cycles = 1.28 M cycles
    const int NV = 40000;           // 40000 x 8 fma x 4 cycles = 1.28 M cycles
    const int NI = 20000;           // 20000 x 4 v_mad_u64_u32
    const float t0 = run<0>(o, NM, 0, 0), t1 = run<1>(o, 0, NV, 0), t2 = run<2>(o, NM, NV, 0);
    printf("{\"mfma_only_ms\": %.3f, \"valu_only_ms\": %.3f, \"both_other_wave_ms\": %.3f}\n", t0, t1, t2);
    for (int per = 0; per <= 2; ++per) printf("{\"same_wave_fma_per_mfma\": %d, \"ms\": %.3f}\n", per * 8, run<3>(o, NM, 0, per));
    const float t4 = run<4>(o, 0, NI, 0), t5 = run<5>(o, NM, NI, 0);
    printf("{\"mad_u64_only_ms\": %.3f, \"cycles_per_mad_u64_at_2.1GHz\": %.1f, \"mfma_plus_mad_u64_other_wave_ms\": %.3f}\n", t4, t4 * 2.1e6 / (NI * 4.0), t5);
    return 0;
}
