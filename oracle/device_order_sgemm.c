/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): nothing under neuronika_amd/, host/ or include/ may use it.
 *
 * A bit-level CPU model of the SUMMATION ORDER of the device GEMM (neuronika_amd/csrc/nk_gemm.hip), so that parity
 * tests can tell apart the two things that separate the HIP result from the reference's `general_mat_mul`
 * (/root/reference/neuronika-variable/src/node/matrix_matrix_mul/mod.rs:33,65,97 -> crate matrixmultiply ^0.3, whose
 * sgemm packs K in blocks of kc = 256 and adds each block's register-accumulated product to C): rounding of the
 * single products is identical everywhere (one fused multiply-add per product), what differs is the ORDER of the sums.
 *
 * The device order: `v_mfma_f32_32x32x2_f32` is an exact f32 fma chain (MI355X_MICROARCH.md, "Matrix cores"); a wave
 * feeds it k in pairs - inside each group of 8 consecutive k, step s takes k = s and k = 4 + s (nk_mma.h, mma_tile) -
 * and every `kc` values of k (kc = 0: never) the running accumulator is folded into a second one that starts at zero
 * (`sgemm_kernel`'s K-blocked accumulation).  The value of C[i][j] is therefore
 *     total = 0; acc = 0
 *     for each k-group of 8 (ascending):  for s in 0..3:  acc = fmaf(a[k8+s], b[k8+s], acc); acc = fmaf(a[k8+4+s], b[k8+4+s], acc)
 *         after every kc values of k:     total = total + acc; acc = 0
 *     C = total + acc            (when K is not a multiple of 8 the tail is zero-padded, as the kernel's guarded loader does)
 * A is M x K row-major (lda), B is K x N row-major (ldb), C is M x N row-major (ldc): callers transpose on the host.
 *
 * Build: gcc -O3 -march=native -fopenmp -fno-math-errno -ffp-contract=off -shared -fPIC (oracle/build_c.py).
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#define JB 256

void nk_oracle_sgemm_device_order(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                                  int kc, int pair_second_first) {
    const int K8 = (K + 7) / 8 * 8;
#pragma omp parallel for schedule(dynamic, 1)
    for (int j0 = 0; j0 < N; j0 += JB) {
        const int jn = N - j0 < JB ? N - j0 : JB;
        float acc[JB], tot[JB];
        for (int i = 0; i < M; ++i) {
            memset(acc, 0, sizeof acc);
            memset(tot, 0, sizeof tot);
            int since = 0;
            for (int k8 = 0; k8 < K8; k8 += 8) {
                for (int s = 0; s < 4; ++s) {
                    for (int h = 0; h < 2; ++h) {
                        const int k = k8 + s + 4 * (pair_second_first ? 1 - h : h);
                        if (k >= K) continue; /* zero-padded tail: fmaf(0, 0, acc) == acc */
                        const float a = A[(size_t)i * lda + k];
                        const float* b = B + (size_t)k * ldb + j0;
                        for (int j = 0; j < jn; ++j) acc[j] = __builtin_fmaf(a, b[j], acc[j]);
                    }
                }
                since += 8;
                if (kc > 0 && since >= kc && k8 + 8 < K8) {
                    for (int j = 0; j < jn; ++j) { tot[j] += acc[j]; acc[j] = 0.f; }
                    since = 0;
                }
            }
            float* c = C + (size_t)i * ldc + j0;
            for (int j = 0; j < jn; ++j) c[j] = kc > 0 ? tot[j] + acc[j] : acc[j];
        }
    }
}
