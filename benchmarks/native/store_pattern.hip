// Store-only experiment: 128x128 f32 tiles of a [512][1024][1024] tensor written (a) in the accumulator layout of the
// 32x32 MFMA (64 four-byte stores per lane, a wave store = 2 rows x 128 B) and (b) row-major with 16-byte stores (a wave
// store = 2 rows x 512 B).  How much of the K = 64 attention GEMMs' time is the shape of their epilogue?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void store_mfma_layout(float* C, int tiles_n, long long ldc, long long sC) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    float* Cb = C + blockIdx.z * sC + (long long)tm * 128 * ldc + tn * 128;
    const int c = lane & 31, h = lane >> 5;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wr * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col = (wc * 2 + j) * 32 + c;
                Cb[row * ldc + col] = (float)(row + col);
            }
}
__global__ __launch_bounds__(256, 2) void store_rowmajor_v4(float* C, int tiles_n, long long ldc, long long sC) {
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    float* Cb = C + blockIdx.z * sC + (long long)tm * 128 * ldc + tn * 128;
    const int t = threadIdx.x, q = t & 31, r0 = t >> 5;  // 32 quads per row, 8 rows per pass
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int row = r0 + 8 * p;
        *reinterpret_cast<float4*>(Cb + row * ldc + q * 4) = make_float4(row, q, p, 1.f);
    }
}
int main() {
    const int B = 512, S = 1024;
    float* C;
    CK(hipMalloc(&C, (size_t)B * S * S * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid(64, 1, B), block(256);
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 5; ++it) {
                if (v == 0) hipLaunchKernelGGL(store_mfma_layout, grid, block, 0, 0, C, 8, (long long)S, (long long)S * S);
                else hipLaunchKernelGGL(store_rowmajor_v4, grid, block, 0, 0, C, 8, (long long)S, (long long)S * S);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("%s: %.1f us per launch, %.2f TB/s\n", v == 0 ? "mfma-layout 4-byte stores" : "row-major 16-byte stores", ms / 5 * 1e3,
                                 (double)B * S * S * 4 / (ms / 5 * 1e-3) / 1e12);
        }
    }
    return 0;
}
