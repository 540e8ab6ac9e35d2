// Matrix-vector, vector-matrix and vector-vector products with their backward passes.  All of them
// read the matrix exactly once -> HBM-bound (8 TB/s), not MFMA work: 2 flop per 4 bytes.
//   MatrixVectorMul   node/matrix_vector_mul/mod.rs:31-41 (y = A.x), :63-69 (dA += g (x) x), :92-102 (dx += A^T.g)
//   VectorMatrixMul   node/vector_matrix_mul/mod.rs:31-41 (y = v.B), :63-73 (dv += B.g), :95-101 (dB += v (x) g)
//   VectorVectorMul   node/vector_vector_mul/mod.rs:31-34 (dot), :57-63 (d_op += other * g)
// Row-major matrices.  Three primitives:
//   rows_dot  : y[i] (+)= sum_j A[i][j] x[j]   one wave (or one block) per row, 16-B loads, wave64 shuffles
//   cols_comb : y[j] (+)= sum_i A[i][j] x[i]   thread per 4 columns (coalesced), rows split over blockIdx.y,
//                                               partials in the workspace, summed in split order (deterministic)
//   outer_add : D[i][j] += u[i] v[j]
#include "nk_common.h"

namespace {

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool ACC, bool VEC, bool BLOCK_PER_ROW>
__global__ void rows_dot_kernel(const float* __restrict__ A, const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row_step = BLOCK_PER_ROW ? gridDim.x : gridDim.x * 4;
    for (int r = BLOCK_PER_ROW ? blockIdx.x : blockIdx.x * 4 + wave; r < rows; r += row_step) {
        const float* a = A + (size_t)r * cols;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const int first = BLOCK_PER_ROW ? threadIdx.x : lane, step = BLOCK_PER_ROW ? 256 : 64;
        if (VEC) {
            for (int c = first; c < cols / 4; c += step) {
                const float4 av = reinterpret_cast<const float4*>(a)[c], xv = reinterpret_cast<const float4*>(x)[c];
                s0 = fmaf(av.x, xv.x, s0); s1 = fmaf(av.y, xv.y, s1); s2 = fmaf(av.z, xv.z, s2); s3 = fmaf(av.w, xv.w, s3);
            }
        } else {
            for (int c = first; c < cols; c += step) s0 = fmaf(a[c], x[c], s0);
        }
        float s = (s0 + s1) + (s2 + s3);
        if (BLOCK_PER_ROW) {
            s = nk_block_sum<256>(s, red);
            if (threadIdx.x == 0) y[r] = ACC ? y[r] + s : s;
            __syncthreads();
        } else {
            s = nk_wave_sum(s);
            if (lane == 0) y[r] = ACC ? y[r] + s : s;
        }
    }
}

// partial[split][j..j+3] = sum over the split's rows of A[i][j] * x[i]; when gridDim.y == 1 it writes y directly.
template <bool ACC, bool VEC>
__global__ void cols_comb_kernel(const float* __restrict__ A, const float* __restrict__ x, float* __restrict__ y,
                                 float* __restrict__ part, int rows, int cols, int rows_per_split) {
    const int r0 = blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    float* dst = gridDim.y == 1 ? y : part + (size_t)blockIdx.y * cols;
    const bool direct_acc = ACC && gridDim.y == 1;
    if (VEC) {
        const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
        if (c >= cols) return;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int r = r0; r < r1; ++r) {
            const float4 av = *reinterpret_cast<const float4*>(A + (size_t)r * cols + c);
            const float xv = x[r];
            s.x = fmaf(av.x, xv, s.x); s.y = fmaf(av.y, xv, s.y); s.z = fmaf(av.z, xv, s.z); s.w = fmaf(av.w, xv, s.w);
        }
        if (direct_acc) { const float4 o = *reinterpret_cast<float4*>(dst + c); s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        *reinterpret_cast<float4*>(dst + c) = s;
    } else {
        const int c = blockIdx.x * blockDim.x + threadIdx.x;
        if (c >= cols) return;
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s = fmaf(A[(size_t)r * cols + c], x[r], s);
        dst[c] = direct_acc ? dst[c] + s : s;
    }
}

template <bool ACC>
__global__ void cols_final_kernel(const float* __restrict__ part, float* __restrict__ y, int cols, int splits) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * cols + c];
    y[c] = ACC ? y[c] + s : s;
}

template <bool VEC>
__global__ void outer_add_kernel(float* __restrict__ D, const float* __restrict__ u, const float* __restrict__ v, int rows, int cols) {
    const int cq = VEC ? cols / 4 : cols;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const float ur = u[r];
        float* d = D + (size_t)r * cols;
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cq; c += gridDim.x * blockDim.x) {
            if (VEC) {
                float4 o = reinterpret_cast<float4*>(d)[c];
                const float4 vv = reinterpret_cast<const float4*>(v)[c];
                o.x = fmaf(ur, vv.x, o.x); o.y = fmaf(ur, vv.y, o.y); o.z = fmaf(ur, vv.z, o.z); o.w = fmaf(ur, vv.w, o.w);
                reinterpret_cast<float4*>(d)[c] = o;
            } else {
                d[c] = fmaf(ur, v[c], d[c]);
            }
        }
    }
}

__global__ void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s = fmaf(a[i], b[i], s);
    s = nk_block_sum<256>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void dot_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
    s = nk_block_sum<256>(s, red);
    if (threadIdx.x == 0) out[0] = s;
}
__global__ void scaled_add_kernel(float* __restrict__ d, const float* __restrict__ other, const float* __restrict__ gs, size_t n) {
    const float g = gs[0];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = fmaf(other[i], g, d[i]);
}

// y[i] (+)= A[i,:] . x
template <bool ACC>
int rows_dot(nk_device* dev, const float* A, const float* x, float* y, int rows, int cols) {
    NK_USE(dev);
    NK_CHECK(rows >= 0 && cols >= 0, "negative extent");
    if (rows == 0) return NK_OK;
    NK_CHECK(y && (cols == 0 || (A && x)), "null pointer");
    const bool vec = cols % 4 == 0 && al16(A) && al16(x);
    const bool bpr = rows < 2048;  // few rows: a whole block per row keeps the chip busy
    const int grid = bpr ? rows : min((rows + 3) / 4, 8192);
#define NK_LAUNCH_RD(V, B) hipLaunchKernelGGL((rows_dot_kernel<ACC, V, B>), dim3(grid), dim3(256), 0, dev->compute, A, x, y, rows, cols)
    if (vec) { if (bpr) NK_LAUNCH_RD(true, true); else NK_LAUNCH_RD(true, false); }
    else     { if (bpr) NK_LAUNCH_RD(false, true); else NK_LAUNCH_RD(false, false); }
#undef NK_LAUNCH_RD
    NK_LAUNCH_CHECK();
    return NK_OK;
}

// y[j] (+)= sum_i A[i][j] x[i]
template <bool ACC>
int cols_comb(nk_device* dev, const float* A, const float* x, float* y, int rows, int cols) {
    NK_USE(dev);
    NK_CHECK(rows >= 0 && cols >= 0, "negative extent");
    if (cols == 0) return NK_OK;
    NK_CHECK(y && (rows == 0 || (A && x)), "null pointer");
    const bool vec = cols % 4 == 0 && al16(A) && al16(y);
    const int per_block = vec ? 1024 : 256;
    const int col_blocks = (cols + per_block - 1) / per_block;
    int splits = 2048 / col_blocks;                         // ~8 blocks per CU in flight
    splits = max(1, min(splits, (rows + 31) / 32));         // at least 32 rows per split
    const int rps = splits ? (rows + splits - 1) / splits : rows;
    splits = rps ? (rows + rps - 1) / rps : 1;
    if (splits < 1) splits = 1;
    float* part = nullptr;
    if (splits > 1) {
        void* ws = nullptr;
        int rc = nk_workspace(dev, (size_t)splits * cols * sizeof(float), &ws);
        if (rc) return rc;
        part = (float*)ws;
    }
    const dim3 grid(col_blocks, splits);
    if (vec) hipLaunchKernelGGL((cols_comb_kernel<ACC, true>), grid, dim3(256), 0, dev->compute, A, x, y, part, rows, cols, rps);
    else hipLaunchKernelGGL((cols_comb_kernel<ACC, false>), grid, dim3(256), 0, dev->compute, A, x, y, part, rows, cols, rps);
    NK_LAUNCH_CHECK();
    if (splits > 1) {
        hipLaunchKernelGGL((cols_final_kernel<ACC>), dim3((cols + 255) / 256), dim3(256), 0, dev->compute, part, y, cols, splits);
        NK_LAUNCH_CHECK();
    }
    return NK_OK;
}

int outer_add(nk_device* dev, float* D, const float* u, const float* v, int rows, int cols) {
    NK_USE(dev);
    NK_CHECK(rows >= 0 && cols >= 0, "negative extent");
    if ((size_t)rows * cols == 0) return NK_OK;
    NK_CHECK(D && u && v, "null pointer");
    const bool vec = cols % 4 == 0 && al16(D) && al16(v);
    const int cq = vec ? cols / 4 : cols;
    const dim3 grid(min((cq + 255) / 256, 64), min(rows, 65535));
    if (vec) hipLaunchKernelGGL(outer_add_kernel<true>, grid, dim3(256), 0, dev->compute, D, u, v, rows, cols);
    else hipLaunchKernelGGL(outer_add_kernel<false>, grid, dim3(256), 0, dev->compute, D, u, v, rows, cols);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

}  // namespace

extern "C" {

int nk_mv_fwd(nk_device* dev, const float* A, const float* x, float* y, int n, int m) { return rows_dot<false>(dev, A, x, y, n, m); }
int nk_mv_bwd_left(nk_device* dev, float* dA, const float* g, const float* x, int n, int m) { return outer_add(dev, dA, g, x, n, m); }
int nk_mv_bwd_right(nk_device* dev, float* dx, const float* A, const float* g, int n, int m) { return cols_comb<true>(dev, A, g, dx, n, m); }

int nk_vm_fwd(nk_device* dev, const float* v, const float* B, float* y, int m, int o) { return cols_comb<false>(dev, B, v, y, m, o); }
int nk_vm_bwd_left(nk_device* dev, float* dv, const float* B, const float* g, int m, int o) { return rows_dot<true>(dev, B, g, dv, m, o); }
int nk_vm_bwd_right(nk_device* dev, float* dB, const float* v, const float* g, int m, int o) { return outer_add(dev, dB, v, g, m, o); }

int nk_vv_fwd(nk_device* dev, const float* l, const float* r, size_t n, float* out) {
    NK_USE(dev);
    NK_CHECK(out != nullptr, "null output scalar");
    NK_CHECK(n == 0 || (l && r), "null operand");
    void* ws = nullptr;
    int rc = nk_workspace(dev, 1024 * sizeof(float), &ws);
    if (rc) return rc;
    int parts = nk_stream_grid(n + 1, 256);
    if (parts > 1024) parts = 1024;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(parts), dim3(256), 0, dev->compute, l, r, n, (float*)ws);
    NK_LAUNCH_CHECK();
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, dev->compute, (const float*)ws, parts, out);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

int nk_vv_bwd(nk_device* dev, float* d_operand, const float* other, const float* g, size_t n) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(d_operand && other && g, "null pointer");
    hipLaunchKernelGGL(scaled_add_kernel, dim3(nk_stream_grid(n, 256)), dim3(256), 0, dev->compute, d_operand, other, g, n);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

}  // extern "C"
