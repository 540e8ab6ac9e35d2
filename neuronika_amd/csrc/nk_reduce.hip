// Reductions and (log-)softmax: HBM-bound, wave64 shuffles, rows cached in registers so every
// element is read once and written once (bound: 8 TB/s HBM3E).
//   Sum / Mean          node/sum/mod.rs:28-35,60-67 ; node/mean/mod.rs:28-35,60-72
//   SquaredError        node/squared_error/mod.rs:42-59,94-123
//   Softmax / LogSoftmax node/softmax/mod.rs:37-53,84-104 ; node/logsoftmax/mod.rs:37-53,84-102
#include "nk_common.h"

namespace {

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int RB = 256;       // reduction block
constexpr int MAX_PART = 1024;  // partial sums per full reduction

// MODE 0: sum x ; MODE 1: sum (x-t)^2
template <int MODE>
__global__ void reduce_partial_kernel(const float* __restrict__ x, const float* __restrict__ t, size_t n,
                                      float* __restrict__ part) {
    __shared__ float red[RB / 64];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const size_t n4 = n / 4;
    // (span walk, nk_common.h: a lane sums the quads of its block's span in address order - a fixed order, function of (n, grid))
    struct R { float4 x, t; };
    nk_span_walk<4>(n4, [&](size_t i) {
        R r;
        r.x = reinterpret_cast<const float4*>(x)[i];
        r.t = MODE == 1 ? reinterpret_cast<const float4*>(t)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        return r;
    }, [&](size_t, const R& r) {
        float4 v = r.x;
        if (MODE == 1) {
            v.x -= r.t.x; v.y -= r.t.y; v.z -= r.t.z; v.w -= r.t.w;
            v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w;
        }
        a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
    });
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        float v = x[i];
        if (MODE == 1) { v -= t[i]; v *= v; }
        a0 += v;
    }
    const float s = nk_block_sum<RB>((a0 + a1) + (a2 + a3), red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void reduce_final_kernel(const float* __restrict__ part, int nparts, float scale_den, float* __restrict__ out) {
    __shared__ float red[RB / 64];
    float a = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) a += part[i];
    const float s = nk_block_sum<RB>(a, red);
    if (threadIdx.x == 0) out[0] = scale_den > 0.f ? s / scale_den : s;
}

// MODE 0: dx += g            (SumBackward)
// MODE 1: dx += g / den      (MeanBackward: `grad_el / den`)
// MODE 2: dx += (2*(x-t))*g / den   (SquaredErrorBackward, Mean)
// MODE 3: dx += (2*(x-t))*g         (SquaredErrorBackward, Sum)
template <int MODE>
__global__ void scalar_bwd_kernel(float* __restrict__ dx, const float* __restrict__ gs, const float* __restrict__ x,
                                  const float* __restrict__ t, size_t n, float den, int assign) {
    const float g = gs[0];
    const size_t n4 = n / 4;
    // bit 1 of `assign`: `nt` loads (operands beyond the Infinity Cache, nk_common.h)
    struct R { float4 d, x, t; };
    nk_span_walk<4>(n4, [&](size_t i) {
        R r;
        r.d = (assign & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : nk_load_stream(reinterpret_cast<const float4*>(dx) + i, assign & 2);
        r.x = r.t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE >= 2) {
            r.x = nk_load_stream(reinterpret_cast<const float4*>(x) + i, assign & 2);
            r.t = nk_load_stream(reinterpret_cast<const float4*>(t) + i, assign & 2);
        }
        return r;
    }, [&](size_t i, const R& r) {
        float4 d = r.d;
        const float4 xv = r.x, tv = r.t;
        if (MODE == 0) { d.x += g; d.y += g; d.z += g; d.w += g; }
        if (MODE == 1) { const float v = g / den; d.x += v; d.y += v; d.z += v; d.w += v; }
        if (MODE == 2) {
            d.x += (2.f * (xv.x - tv.x)) * g / den; d.y += (2.f * (xv.y - tv.y)) * g / den;
            d.z += (2.f * (xv.z - tv.z)) * g / den; d.w += (2.f * (xv.w - tv.w)) * g / den;
        }
        if (MODE == 3) {
            d.x += (2.f * (xv.x - tv.x)) * g; d.y += (2.f * (xv.y - tv.y)) * g;
            d.z += (2.f * (xv.z - tv.z)) * g; d.w += (2.f * (xv.w - tv.w)) * g;
        }
        nk_store_stream(reinterpret_cast<float4*>(dx) + i, d);
    });
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        const float d0 = (assign & 1) ? 0.f : dx[i];
        if (MODE == 0) dx[i] = d0 + g;
        if (MODE == 1) dx[i] = d0 + g / den;
        if (MODE == 2) dx[i] = d0 + (2.f * (x[i] - t[i])) * g / den;
        if (MODE == 3) dx[i] = d0 + (2.f * (x[i] - t[i])) * g;
    }
}

template <int MODE>
int full_reduce(nk_device* dev, const float* x, const float* t, size_t n, float den, float* out) {
    NK_USE(dev);
    NK_CHECK(out != nullptr, "null output scalar");
    NK_CHECK(n == 0 || x != nullptr, "null input");
    NK_CHECK(al16(x) && (MODE == 0 || al16(t)), "reduction inputs must be 16-byte aligned");
    void* ws = nullptr;
    int rc = nk_workspace(dev, MAX_PART * sizeof(float), &ws);
    if (rc) return rc;
    int parts = nk_stream_grid(n / 4 + 1, RB);
    if (parts > MAX_PART) parts = MAX_PART;
    hipLaunchKernelGGL((reduce_partial_kernel<MODE>), dim3(parts), dim3(RB), 0, dev->compute, x, t, n, (float*)ws);
    NK_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(RB), 0, dev->compute, (const float*)ws, parts, den, out);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

template <int MODE>
int scalar_bwd(nk_device* dev, float* dx, const float* g, const float* x, const float* t, size_t n, float den, int assign = 0) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(dx && g, "null pointer");
    NK_CHECK(al16(dx) && (MODE < 2 || (al16(x) && al16(t))), "gradient buffers must be 16-byte aligned");
    assign = (assign ? 1 : 0) | ((MODE >= 2 && nk_streams_past_cache(n * (assign ? 12 : 16))) ? 2 : 0);
    hipLaunchKernelGGL((scalar_bwd_kernel<MODE>), dim3(nk_stream_grid(n / 4 + 1, 256)), dim3(256), 0, dev->compute, dx, g, x, t, n, den, assign);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

// --------------------------------------------------------------------------- softmax ----------
// Row kernels (axis is the innermost, contiguous one): ONE WAVE PER ROW, the row lives in
// registers (V float4 per lane, L <= 256*V), max / sum by wave64 xor-shuffles.
// LOG: log-softmax.   The max fold starts from f32::MIN (finite), softmax/mod.rs:45.
constexpr float F32_MIN = -3.40282347e+38f;


// The V quads of a lane's row slice, ALL issued before any is used: a load inside `if (c < L) { load; use; }` gets its
// own vmcnt(0), i.e. V serialised memory round trips per row.  Lanes beyond the row re-read its first quad (a row has
// at least one) and are masked by the `c < L` tests of the compute loops.
template <int V>
__device__ __forceinline__ void row_load(float4 (&v)[V], const float* __restrict__ row, int lane, int L, bool stream = false) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = nk_load_stream(reinterpret_cast<const float4*>(row + (c < L ? c : 0)), stream);  // nk_common.h: `nt` past the cache
    }
}

template <int V, bool LOG>
__global__ __launch_bounds__(256) void softmax_fwd_row_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * L;
    float* yr = y + row * L;
    float4 v[V];
    row_load<V>(v, xr, lane, L);
    float m = F32_MIN;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    }
    m = nk_wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) {
            float4 e;
            e.x = expf(v[i].x - m); e.y = expf(v[i].y - m); e.z = expf(v[i].z - m); e.w = expf(v[i].w - m);
            s += (e.x + e.y) + (e.z + e.w);
            if (!LOG) v[i] = e;
        }
    }
    s = nk_wave_sum(s);
    const float lse = LOG ? logf(s) : 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) {
            float4 o;
            if (LOG) { o.x = v[i].x - lse - m; o.y = v[i].y - lse - m; o.z = v[i].z - lse - m; o.w = v[i].w - lse - m; }
            else { o.x = v[i].x / s; o.y = v[i].y / s; o.z = v[i].z / s; o.w = v[i].w / s; }
            nk_store_stream(reinterpret_cast<float4*>(yr + c), o);
        }
    }
}

// softmax: dx += y*(g - sum(g*y)) ; log-softmax: dx += g - exp(y)*sum(g)
template <int V, bool LOG>
__global__ __launch_bounds__(256) void softmax_bwd_row_kernel(int assign, float* __restrict__ dx, const float* __restrict__ g, const float* __restrict__ y,
                                       long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* gr = g + row * L;
    const float* yr = y + row * L;
    float* dr = dx + row * L;
    float4 gv[V], yv[V], dv[V];
    const bool nt = assign & 2;  // bit 1 of `assign`: operands beyond the Infinity Cache (launch-time choice)
    assign &= 1;
    row_load<V>(gv, gr, lane, L, nt);
    row_load<V>(yv, yr, lane, L, nt);
    if (!assign) row_load<V>(dv, dr, lane, L, nt);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) {
            if (LOG) s += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
            else s += (gv[i].x * yv[i].x + gv[i].y * yv[i].y) + (gv[i].z * yv[i].z + gv[i].w * yv[i].w);
        }
    }
    s = nk_wave_sum(s);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) {
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!assign) { d.x = dv[i].x; d.y = dv[i].y; d.z = dv[i].z; d.w = dv[i].w; }
            if (LOG) {
                d.x += gv[i].x - expf(yv[i].x) * s; d.y += gv[i].y - expf(yv[i].y) * s;
                d.z += gv[i].z - expf(yv[i].z) * s; d.w += gv[i].w - expf(yv[i].w) * s;
            } else {
                d.x += yv[i].x * (gv[i].x - s); d.y += yv[i].y * (gv[i].y - s);
                d.z += yv[i].z * (gv[i].z - s); d.w += yv[i].w * (gv[i].w - s);
            }
            nk_store_stream(reinterpret_cast<float4*>(dr + c), d);
        }
    }
}

// General kernels: lanes of length L with element stride `inner` (any axis, any L).
// inner == 1: one 256-thread block per lane, threads stride along the lane.
// inner  > 1: one thread per (outer, inner) lane, threads along `inner` (coalesced).
template <bool LOG>
__global__ void softmax_fwd_block_kernel(const float* __restrict__ x, float* __restrict__ y, int L) {
    __shared__ float red[4];
    const float* xr = x + (long long)blockIdx.x * L;
    float* yr = y + (long long)blockIdx.x * L;
    float m = F32_MIN;
    for (int c = threadIdx.x; c < L; c += blockDim.x) m = fmaxf(m, xr[c]);
    m = nk_block_max<256>(m, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < L; c += blockDim.x) s += expf(xr[c] - m);
    s = nk_block_sum<256>(s, red);
    const float lse = LOG ? logf(s) : 0.f;
    for (int c = threadIdx.x; c < L; c += blockDim.x) yr[c] = LOG ? xr[c] - lse - m : expf(xr[c] - m) / s;
}

template <bool LOG>
__global__ void softmax_bwd_block_kernel(int assign, float* __restrict__ dx, const float* __restrict__ g, const float* __restrict__ y, int L) {
    __shared__ float red[4];
    const long long o = (long long)blockIdx.x * L;
    float s = 0.f;
    for (int c = threadIdx.x; c < L; c += blockDim.x) s += LOG ? g[o + c] : g[o + c] * y[o + c];
    s = nk_block_sum<256>(s, red);
    for (int c = threadIdx.x; c < L; c += blockDim.x)
        dx[o + c] = (assign ? 0.f : dx[o + c]) + (LOG ? g[o + c] - expf(y[o + c]) * s : y[o + c] * (g[o + c] - s));
}

template <bool LOG>
__global__ void softmax_fwd_strided_kernel(const float* __restrict__ x, float* __restrict__ y, long long lanes, int L,
                                           long long inner) {
    for (long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x; id < lanes;
         id += (long long)gridDim.x * blockDim.x) {
        const long long base = (id / inner) * L * inner + id % inner;
        float m = F32_MIN;
        for (int c = 0; c < L; ++c) m = fmaxf(m, x[base + c * inner]);
        float s = 0.f;
        for (int c = 0; c < L; ++c) s += expf(x[base + c * inner] - m);
        const float lse = LOG ? logf(s) : 0.f;
        for (int c = 0; c < L; ++c) {
            const float v = x[base + c * inner];
            y[base + c * inner] = LOG ? v - lse - m : expf(v - m) / s;
        }
    }
}

template <bool LOG>
__global__ void softmax_bwd_strided_kernel(int assign, float* __restrict__ dx, const float* __restrict__ g, const float* __restrict__ y,
                                           long long lanes, int L, long long inner) {
    for (long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x; id < lanes;
         id += (long long)gridDim.x * blockDim.x) {
        const long long base = (id / inner) * L * inner + id % inner;
        float s = 0.f;
        for (int c = 0; c < L; ++c) s += LOG ? g[base + c * inner] : g[base + c * inner] * y[base + c * inner];
        for (int c = 0; c < L; ++c) {
            const long long o = base + c * inner;
            dx[o] = (assign ? 0.f : dx[o]) + (LOG ? g[o] - expf(y[o]) * s : y[o] * (g[o] - s));
        }
    }
}

// ---------------------------------------------------------------- fused attention probabilities --
// One wave per row, the row in registers (V float4 per lane): scale, max, exp, sum, normalise,
// Philox mask, dropout — scores are read once, probs / out written once.  MASK: 0 = no dropout
// (eval / p == 0: out = probs), 1 = Bernoulli mask, 2 = p == 1 (out = 0).
template <int V, int MASK, bool STORE_NOISE>
__global__ __launch_bounds__(256) void attn_probs_fwd_kernel(const float* __restrict__ s, float* __restrict__ probs, float* __restrict__ out,
                                      float* __restrict__ noise, long long rows, int L, float scale, unsigned keep_lt,
                                      float dscale, unsigned long long seed, unsigned long long offset) {
    // No fma contraction: which products get fused depends on the instantiation (MASK / RECOMP / ...), and the stored-
    // probabilities and recomputed-probabilities paths must produce the same bits; it is also what the reference's
    // separately rounded node-by-node arithmetic does.
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long rb = row * L;
    float4 v[V];
    row_load<V>(v, s + rb, lane, L);
    float m = F32_MIN;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) {
            // Multiplication node: a separately ROUNDED product (no fma with the `- m` below) - the reference's node
            // boundary, and what keeps this kernel and the recomputing backward kernel bit-identical
            v[i].x = __fmul_rn(v[i].x, scale); v[i].y = __fmul_rn(v[i].y, scale);
            v[i].z = __fmul_rn(v[i].z, scale); v[i].w = __fmul_rn(v[i].w, scale);
            m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        }
    }
    m = nk_wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) {
            float4 e;
            e.x = expf(__fsub_rn(v[i].x, m)); e.y = expf(__fsub_rn(v[i].y, m));
            e.z = expf(__fsub_rn(v[i].z, m)); e.w = expf(__fsub_rn(v[i].w, m));
            sum += (e.x + e.y) + (e.z + e.w);
            v[i] = e;
        }
    }
    sum = nk_wave_sum(sum);
    // ONE IEEE division per row, a multiplication per element (and the dropout scale as a multiplication by the host's
    // 1 / (1 - p)): the two per-element divisions of the node-by-node form (e / sum, then / (1 - p)) were ~20 of the
    // kernel's ~60 VALU instructions per element and made it VALU-bound at 4.8 TB/s.  Each replaced division differs by
    // at most one rounding from the reference's node arithmetic - inside the stated f32 tolerance (rtol 1e-5), checked
    // against the oracle; the recomputing backward kernel uses the identical sequence, so forward and backward still see
    // the same probabilities bit for bit.
    const float inv_sum = 1.f / sum;
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) {
            float4 y;
            y.x = v[i].x * inv_sum; y.y = v[i].y * inv_sum; y.z = v[i].z * inv_sum; y.w = v[i].w * inv_sum;   // Softmax node
            if (probs) nk_store_stream(reinterpret_cast<float4*>(probs + rb + c), y);  // not stored when the backward pass recomputes it
            float4 o = y;
            if (MASK == 1) {                                                                   // Dropout node
                // (a lane's quads are 1 KB apart: each takes its half of the call that covers it - the row kernels pay a
                //  call per four elements as before; the flat kernel and the fused attention core use all eight draws)
                const float4 nz = nk_keep4_at((unsigned long long)(rb + c), offset, key, keep_lt);
                o.x = (y.x * nz.x) * dscale; o.y = (y.y * nz.y) * dscale; o.z = (y.z * nz.z) * dscale; o.w = (y.w * nz.w) * dscale;
                if (STORE_NOISE) nk_store_stream(reinterpret_cast<float4*>(noise + rb + c), nz);
            } else if (MASK == 2) {
                o = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            nk_store_stream(reinterpret_cast<float4*>(out + rb + c), o);
        }
    }
}

// MASK 0: g_p = g ; 1: g_p = g * mask (stored or regenerated)
// RECOMP: `probs` holds the SCORES; the probabilities are recomputed with the forward kernel's exact sequence
// (scale, wave max, expf, wave sum, divide) instead of being read back - the forward then never writes them.
template <int V, int MASK, bool LOAD_NOISE, bool RECOMP>
__global__ __launch_bounds__(256) void attn_probs_bwd_kernel(float* __restrict__ ds, const float* __restrict__ g, const float* __restrict__ probs,
                                      const float* __restrict__ noise, long long rows, int L, float scale, unsigned keep_lt,
                                      unsigned long long seed, unsigned long long offset, int assign) {
    // No fma contraction: which products get fused depends on the instantiation (MASK / RECOMP / ...), and the stored-
    // probabilities and recomputed-probabilities paths must produce the same bits; it is also what the reference's
    // separately rounded node-by-node arithmetic does.
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long rb = row * L;
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
    float4 gp[V], y[V], nzv[V], dv[V];
    const bool nt = assign & 2;  // bit 1 of `assign`: operands beyond the Infinity Cache (launch-time choice)
    assign &= 1;
    row_load<V>(y, probs + rb, lane, L, nt);  // probabilities, or the scores they are recomputed from
    row_load<V>(gp, g + rb, lane, L, nt);
    if (MASK == 1 && LOAD_NOISE) row_load<V>(nzv, noise + rb, lane, L, nt);
    if (!assign) row_load<V>(dv, ds + rb, lane, L, nt);
    if (RECOMP) {
        float m = F32_MIN;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < L) {
                y[i].x = __fmul_rn(y[i].x, scale); y[i].y = __fmul_rn(y[i].y, scale);   // as in the forward kernel
                y[i].z = __fmul_rn(y[i].z, scale); y[i].w = __fmul_rn(y[i].w, scale);
                m = fmaxf(m, fmaxf(fmaxf(y[i].x, y[i].y), fmaxf(y[i].z, y[i].w)));
            }
        }
        m = nk_wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < L) {
                float4 e;
                e.x = expf(__fsub_rn(y[i].x, m)); e.y = expf(__fsub_rn(y[i].y, m));
                e.z = expf(__fsub_rn(y[i].z, m)); e.w = expf(__fsub_rn(y[i].w, m));
                sum += (e.x + e.y) + (e.z + e.w);
                y[i] = e;
            }
        }
        sum = nk_wave_sum(sum);
        const float inv_sum = 1.f / sum;  // as in the forward kernel
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < L) { y[i].x *= inv_sum; y[i].y *= inv_sum; y[i].z *= inv_sum; y[i].w *= inv_sum; }
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) {
            float4 gv = gp[i];
            if (MASK == 1) {                                                                   // DropoutBackward
                float4 nz;
                if (LOAD_NOISE) nz = nzv[i];
                else {
                    nz = nk_keep4_at((unsigned long long)(rb + c), offset, key, keep_lt);
                }
                gv.x *= nz.x; gv.y *= nz.y; gv.z *= nz.z; gv.w *= nz.w;
            }
            gp[i] = gv;
            dot += (gv.x * y[i].x + gv.y * y[i].y) + (gv.z * y[i].z + gv.w * y[i].w);
        }
    }
    dot = nk_wave_sum(dot);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < L) {
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!assign) { d.x = dv[i].x; d.y = dv[i].y; d.z = dv[i].z; d.w = dv[i].w; }
            d.x += (y[i].x * (gp[i].x - dot)) * scale;   // SoftmaxBackward then MultiplicationBackwardLeft
            d.y += (y[i].y * (gp[i].y - dot)) * scale;
            d.z += (y[i].z * (gp[i].z - dot)) * scale;
            d.w += (y[i].w * (gp[i].w - dot)) * scale;
            nk_store_stream(reinterpret_cast<float4*>(ds + rb + c), d);
        }
    }
}

int lane_geometry(const int* shape, int nd, int axis, long long* outer, int* L, long long* inner) {
    NK_CHECK(nd >= 1 && nd <= NK_MAX_DIMS, "bad rank %d", nd);
    NK_CHECK(axis >= 0 && axis < nd, "axis %d out of range for rank %d", axis, nd);
    *outer = 1; *inner = 1;
    for (int i = 0; i < axis; ++i) *outer *= shape[i];
    for (int i = axis + 1; i < nd; ++i) *inner *= shape[i];
    *L = shape[axis];
    return NK_OK;
}

template <bool LOG>
int softmax_fwd(nk_device* dev, const float* x, float* y, const int* shape, int nd, int axis) {
    NK_USE(dev);
    long long outer, inner; int L;
    int rc = lane_geometry(shape, nd, axis, &outer, &L, &inner);
    if (rc) return rc;
    if (outer * inner * L == 0) return NK_OK;
    NK_CHECK(x && y, "null pointer");
    if (inner == 1) {
        const bool vec = (L % 4 == 0) && al16(x) && al16(y) && L <= 2048;
        if (vec) {
            const int wpb = 4;
            const dim3 grid((unsigned)((outer + wpb - 1) / wpb)), block(64 * wpb);
            if (L <= 256) hipLaunchKernelGGL((softmax_fwd_row_kernel<1, LOG>), grid, block, 0, dev->compute, x, y, outer, L);
            else if (L <= 512) hipLaunchKernelGGL((softmax_fwd_row_kernel<2, LOG>), grid, block, 0, dev->compute, x, y, outer, L);
            else if (L <= 1024) hipLaunchKernelGGL((softmax_fwd_row_kernel<4, LOG>), grid, block, 0, dev->compute, x, y, outer, L);
            else hipLaunchKernelGGL((softmax_fwd_row_kernel<8, LOG>), grid, block, 0, dev->compute, x, y, outer, L);
        } else {
            hipLaunchKernelGGL((softmax_fwd_block_kernel<LOG>), dim3((unsigned)outer), dim3(256), 0, dev->compute, x, y, L);
        }
    } else {
        const long long lanes = outer * inner;
        hipLaunchKernelGGL((softmax_fwd_strided_kernel<LOG>), dim3(nk_stream_grid((size_t)lanes, 256)), dim3(256), 0,
                           dev->compute, x, y, lanes, L, inner);
    }
    NK_LAUNCH_CHECK();
    return NK_OK;
}

template <bool LOG>
int softmax_bwd(nk_device* dev, float* dx, const float* g, const float* y, const int* shape, int nd, int axis, int assign = 0) {
    NK_USE(dev);
    long long outer, inner; int L;
    int rc = lane_geometry(shape, nd, axis, &outer, &L, &inner);
    if (rc) return rc;
    if (outer * inner * L == 0) return NK_OK;
    NK_CHECK(dx && g && y, "null pointer");
    if (inner == 1) {
        const bool vec = (L % 4 == 0) && al16(dx) && al16(g) && al16(y) && L <= 2048;
        if (vec) {
            const int wpb = 4;
            const dim3 grid((unsigned)((outer + wpb - 1) / wpb)), block(64 * wpb);
            const int nt2 = nk_streams_past_cache((size_t)outer * L * 12) ? 2 : 0;  // g, y (, dx) read once, dx written
            if (L <= 256) hipLaunchKernelGGL((softmax_bwd_row_kernel<1, LOG>), grid, block, 0, dev->compute, assign | nt2, dx, g, y, outer, L);
            else if (L <= 512) hipLaunchKernelGGL((softmax_bwd_row_kernel<2, LOG>), grid, block, 0, dev->compute, assign | nt2, dx, g, y, outer, L);
            else if (L <= 1024) hipLaunchKernelGGL((softmax_bwd_row_kernel<4, LOG>), grid, block, 0, dev->compute, assign | nt2, dx, g, y, outer, L);
            else hipLaunchKernelGGL((softmax_bwd_row_kernel<8, LOG>), grid, block, 0, dev->compute, assign | nt2, dx, g, y, outer, L);
        } else {
            hipLaunchKernelGGL((softmax_bwd_block_kernel<LOG>), dim3((unsigned)outer), dim3(256), 0, dev->compute, assign, dx, g, y, L);
        }
    } else {
        const long long lanes = outer * inner;
        hipLaunchKernelGGL((softmax_bwd_strided_kernel<LOG>), dim3(nk_stream_grid((size_t)lanes, 256)), dim3(256), 0,
                           dev->compute, assign, dx, g, y, lanes, L, inner);
    }
    NK_LAUNCH_CHECK();
    return NK_OK;
}

}  // namespace

extern "C" {

int nk_sum_fwd(nk_device* dev, const float* x, size_t n, float* out) { return full_reduce<0>(dev, x, nullptr, n, 0.f, out); }
int nk_mean_fwd(nk_device* dev, const float* x, size_t n, float* out) { return full_reduce<0>(dev, x, nullptr, n, (float)n, out); }
int nk_sum_bwd(nk_device* dev, float* dx, size_t n, const float* g) { return scalar_bwd<0>(dev, dx, g, nullptr, nullptr, n, 1.f); }
int nk_mean_bwd(nk_device* dev, float* dx, size_t n, const float* g) { return scalar_bwd<1>(dev, dx, g, nullptr, nullptr, n, (float)n); }

int nk_sum_bwd_assign(nk_device* dev, float* dx, size_t n, const float* g) { return scalar_bwd<0>(dev, dx, g, nullptr, nullptr, n, 1.f, 1); }
int nk_mean_bwd_assign(nk_device* dev, float* dx, size_t n, const float* g) { return scalar_bwd<1>(dev, dx, g, nullptr, nullptr, n, (float)n, 1); }

int nk_mse_fwd(nk_device* dev, const float* x, const float* target, size_t n, int reduction, float* out) {
    NK_CHECK(reduction == NK_REDUCTION_SUM || reduction == NK_REDUCTION_MEAN, "unknown reduction %d", reduction);
    NK_CHECK(n == 0 || target != nullptr, "null target");
    return full_reduce<1>(dev, x, target, n, reduction == NK_REDUCTION_MEAN ? (float)n : 0.f, out);
}

static int mse_bwd(nk_device* dev, float* dx, const float* g, const float* x, const float* target, size_t n, int reduction, int assign) {
    NK_CHECK(reduction == NK_REDUCTION_SUM || reduction == NK_REDUCTION_MEAN, "unknown reduction %d", reduction);
    NK_CHECK(n == 0 || (x && target), "null input/target");
    return reduction == NK_REDUCTION_MEAN ? scalar_bwd<2>(dev, dx, g, x, target, n, (float)n, assign)
                                          : scalar_bwd<3>(dev, dx, g, x, target, n, 1.f, assign);
}
int nk_mse_bwd(nk_device* dev, float* dx, const float* g, const float* x, const float* target, size_t n, int reduction) {
    return mse_bwd(dev, dx, g, x, target, n, reduction, 0);
}
int nk_mse_bwd_assign(nk_device* dev, float* dx, const float* g, const float* x, const float* target, size_t n, int reduction) {
    return mse_bwd(dev, dx, g, x, target, n, reduction, 1);
}

int nk_softmax_fwd(nk_device* dev, const float* x, float* y, const int* shape, int nd, int axis) {
    return softmax_fwd<false>(dev, x, y, shape, nd, axis);
}
int nk_softmax_bwd(nk_device* dev, float* dx, const float* g, const float* y, const int* shape, int nd, int axis) {
    return softmax_bwd<false>(dev, dx, g, y, shape, nd, axis);
}
int nk_softmax_bwd_assign(nk_device* dev, float* dx, const float* g, const float* y, const int* shape, int nd, int axis) {
    return softmax_bwd<false>(dev, dx, g, y, shape, nd, axis, 1);
}
int nk_log_softmax_bwd_assign(nk_device* dev, float* dx, const float* g, const float* y, const int* shape, int nd, int axis) {
    return softmax_bwd<true>(dev, dx, g, y, shape, nd, axis, 1);
}
int nk_log_softmax_fwd(nk_device* dev, const float* x, float* y, const int* shape, int nd, int axis) {
    return softmax_fwd<true>(dev, x, y, shape, nd, axis);
}
int nk_log_softmax_bwd(nk_device* dev, float* dx, const float* g, const float* y, const int* shape, int nd, int axis) {
    return softmax_bwd<true>(dev, dx, g, y, shape, nd, axis);
}

int nk_scale_softmax_dropout_fwd(nk_device* dev, const float* scores, float* probs, float* out, float* noise,
                                 long long rows, int L, float scale, double p, int train, uint64_t seed,
                                 uint64_t offset) {
    NK_USE(dev);
    NK_CHECK(p >= 0.0 && p <= 1.0, "Wrong probability received: %g.", p);
    NK_CHECK(rows >= 0 && L >= 0, "negative extent");
    if (rows == 0 || L == 0) return NK_OK;
    NK_CHECK(scores && out, "null pointer in nk_scale_softmax_dropout_fwd");  // probs may be NULL: not stored
    NK_CHECK(L % 4 == 0 && L <= 2048 && al16(scores) && (!probs || al16(probs)) && al16(out) && (!noise || al16(noise)),
             "fused attention probabilities need L %% 4 == 0, L <= 2048 and 16-byte aligned buffers (L=%d)", L);
    const int mask = (!train || p == 0.0) ? 0 : (1.0 - p == 0.0 ? 2 : 1);
    if (mask == 1)
        if (int rc = nk_refuse_capture(dev, "nk_scale_softmax_dropout_fwd: the Philox offset (every replay would draw the same mask)",
                                       "run the training-mode dropout eagerly, or capture the evaluation graph")) return rc;
    const unsigned keep_lt = nk_keep_threshold(1.0 - p);   // Bernoulli::new(1. - p), dropout/mod.rs:46 (nk_common.h)
    const float dscale = 1.f / (1.f - (float)p);           // multiplied in: 1 / (1 - p) rounded once on the host
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define NK_AP(V)                                                                                                   \
    do {                                                                                                           \
        if (mask == 0) hipLaunchKernelGGL((attn_probs_fwd_kernel<V, 0, false>), grid, block, 0, dev->compute, scores, probs, out, noise, rows, L, scale, keep_lt, dscale, (unsigned long long)seed, (unsigned long long)offset); \
        else if (mask == 2) hipLaunchKernelGGL((attn_probs_fwd_kernel<V, 2, false>), grid, block, 0, dev->compute, scores, probs, out, noise, rows, L, scale, keep_lt, dscale, (unsigned long long)seed, (unsigned long long)offset); \
        else if (noise) hipLaunchKernelGGL((attn_probs_fwd_kernel<V, 1, true>), grid, block, 0, dev->compute, scores, probs, out, noise, rows, L, scale, keep_lt, dscale, (unsigned long long)seed, (unsigned long long)offset); \
        else hipLaunchKernelGGL((attn_probs_fwd_kernel<V, 1, false>), grid, block, 0, dev->compute, scores, probs, out, noise, rows, L, scale, keep_lt, dscale, (unsigned long long)seed, (unsigned long long)offset); \
    } while (0)
    if (L <= 256) NK_AP(1); else if (L <= 512) NK_AP(2); else if (L <= 1024) NK_AP(4); else NK_AP(8);
#undef NK_AP
    NK_LAUNCH_CHECK();
    return NK_OK;
}

}  // extern "C"

static int attn_probs_bwd(nk_device* dev, float* d_scores, const float* g_out, const float* probs, const float* noise,
                          long long rows, int L, float scale, double p, int train, uint64_t seed, uint64_t offset, int assign,
                          bool recompute = false) {
    NK_USE(dev);
    NK_CHECK(p >= 0.0 && p <= 1.0, "Wrong probability received: %g.", p);
    NK_CHECK(rows >= 0 && L >= 0, "negative extent");
    if (rows == 0 || L == 0) return NK_OK;
    NK_CHECK(d_scores && g_out && probs, "null pointer in nk_scale_softmax_dropout_bwd");
    NK_CHECK(L % 4 == 0 && L <= 2048 && al16(d_scores) && al16(g_out) && al16(probs) && (!noise || al16(noise)),
             "fused attention probabilities need L %% 4 == 0, L <= 2048 and 16-byte aligned buffers (L=%d)", L);
    // p == 1: the reference's backward multiplies by the (untouched, zero) noise buffer -> no gradient
    const bool masked = train && p != 0.0;
    const bool all_dropped = 1.0 - p == 0.0;
    const unsigned keep_lt = all_dropped ? 0u : nk_keep_threshold(1.0 - p);  // threshold 0: every draw is "dropped"
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    assign |= nk_streams_past_cache((size_t)rows * L * 12) ? 2 : 0;  // bit 1: `nt` loads (operands beyond the Infinity Cache)
#define NK_AP(V)                                                                                                   \
    do {                                                                                                           \
        if (recompute) {                                                                                           \
            if (!masked) hipLaunchKernelGGL((attn_probs_bwd_kernel<V, 0, false, true>), grid, block, 0, dev->compute, d_scores, g_out, probs, noise, rows, L, scale, keep_lt, (unsigned long long)seed, (unsigned long long)offset, assign); \
            else if (noise && !all_dropped) hipLaunchKernelGGL((attn_probs_bwd_kernel<V, 1, true, true>), grid, block, 0, dev->compute, d_scores, g_out, probs, noise, rows, L, scale, keep_lt, (unsigned long long)seed, (unsigned long long)offset, assign); \
            else hipLaunchKernelGGL((attn_probs_bwd_kernel<V, 1, false, true>), grid, block, 0, dev->compute, d_scores, g_out, probs, noise, rows, L, scale, keep_lt, (unsigned long long)seed, (unsigned long long)offset, assign); \
        } else if (!masked) hipLaunchKernelGGL((attn_probs_bwd_kernel<V, 0, false, false>), grid, block, 0, dev->compute, d_scores, g_out, probs, noise, rows, L, scale, keep_lt, (unsigned long long)seed, (unsigned long long)offset, assign); \
        else if (noise && !all_dropped) hipLaunchKernelGGL((attn_probs_bwd_kernel<V, 1, true, false>), grid, block, 0, dev->compute, d_scores, g_out, probs, noise, rows, L, scale, keep_lt, (unsigned long long)seed, (unsigned long long)offset, assign); \
        else hipLaunchKernelGGL((attn_probs_bwd_kernel<V, 1, false, false>), grid, block, 0, dev->compute, d_scores, g_out, probs, noise, rows, L, scale, keep_lt, (unsigned long long)seed, (unsigned long long)offset, assign); \
    } while (0)
    if (L <= 256) NK_AP(1); else if (L <= 512) NK_AP(2); else if (L <= 1024) NK_AP(4); else NK_AP(8);
#undef NK_AP
    NK_LAUNCH_CHECK();
    return NK_OK;
}

extern "C" {
int nk_scale_softmax_dropout_bwd(nk_device* dev, float* d_scores, const float* g_out, const float* probs,
                                 const float* noise, long long rows, int L, float scale, double p, int train,
                                 uint64_t seed, uint64_t offset) {
    return attn_probs_bwd(dev, d_scores, g_out, probs, noise, rows, L, scale, p, train, seed, offset, 0);
}
int nk_scale_softmax_dropout_bwd_assign(nk_device* dev, float* d_scores, const float* g_out, const float* probs,
                                        const float* noise, long long rows, int L, float scale, double p, int train,
                                        uint64_t seed, uint64_t offset) {
    return attn_probs_bwd(dev, d_scores, g_out, probs, noise, rows, L, scale, p, train, seed, offset, 1);
}
int nk_scale_softmax_dropout_bwd_from_scores(nk_device* dev, float* d_scores, const float* g_out, const float* scores,
                                             const float* noise, long long rows, int L, float scale, double p, int train,
                                             uint64_t seed, uint64_t offset, int assign) {
    return attn_probs_bwd(dev, d_scores, g_out, scores, noise, rows, L, scale, p, train, seed, offset, assign ? 1 : 0, true);
}
}  // extern "C"
