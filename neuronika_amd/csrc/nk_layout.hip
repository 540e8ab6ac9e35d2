// Dropout (device RNG) and the data-movement glue of the hot path — all HBM-bound copies:
//   Dropout            node/dropout/mod.rs:53-79,113-128
//   Pad<Constant|Zero> node/pad/mod.rs:97-129,157-181 ; pad/constant/mod.rs:14-39
//   Chunk              node/chunk/mod.rs:48-64,99-113
//   MultiConcatenate   node/multi_concatenate/mod.rs:37-50,81-97
//   Transpose          node/transpose/mod.rs:28-37,62-69
//   head split/merge   = the Chunk / MultiConcatenate pattern of the composed attention
#include "nk_common.h"

namespace {

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// One thread = one aligned quad of elements = its half of the Philox call that covers it (nk_common.h: the draw layout);
// y = (x*noise)/scale.  A thread that took the whole call (two adjacent quads, half the Philox work) was measured at HALF the
// bandwidth: a lane's 32 contiguous bytes make every 16-byte load / store instruction touch each 128-byte line half-filled
// (3.3 vs 6.5 TB/s at 1 GB); the kernel is HBM-bound, the second half of the call is cheaper than the lost coalescing.
__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ noise,
                                   size_t n, unsigned keep_lt, float scale, unsigned long long seed,
                                   unsigned long long offset) {
    const size_t n4 = (n + 3) / 4;
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
    nk_span_walk<4>(n4, [&](size_t i) { return i * 4 + 3 < n ? reinterpret_cast<const float4*>(x)[i] : make_float4(0.f, 0.f, 0.f, 0.f); },
                    [&](size_t i, const float4& xv) {
        const float4 nz = nk_keep4_at((unsigned long long)i * 4, offset, key, keep_lt);
        if (i * 4 + 3 < n) {
            float4 o;
            o.x = (xv.x * nz.x) / scale; o.y = (xv.y * nz.y) / scale; o.z = (xv.z * nz.z) / scale; o.w = (xv.w * nz.w) / scale;
            nk_store_stream(reinterpret_cast<float4*>(y) + i, o);
            nk_store_stream(reinterpret_cast<float4*>(noise) + i, nz);
        } else {
            const float nn[4] = {nz.x, nz.y, nz.z, nz.w};
            for (int c = 0; c < 4; ++c) {
                const size_t e = i * 4 + c;
                if (e < n) { y[e] = (x[e] * nn[c]) / scale; noise[e] = nn[c]; }
            }
        }
    });
}

// MODE 0: dx += g ; MODE 1: dx += g * noise
template <int MODE>
__global__ void dropout_bwd_kernel(float* __restrict__ dx, const float* __restrict__ g, const float* __restrict__ noise, size_t n,
                                   int assign) {
    const size_t n4 = n / 4;
    // (round 3, grid-stride walk: two quads per trip - six loads in flight per lane - measured SLOWER, 4.7 - 5.1 vs 5.6 TB/s at 1 GB;
    // in the span walk (nk_common.h) four quads per trip - twelve loads in flight per lane - are the faster form: 4.4 -> 5.7 TB/s at 1 GiB)
    struct R { float4 d, g, nz; };
    nk_span_walk<4>(n4, [&](size_t i) {
        R r;
        r.d = (assign & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : nk_load_stream(reinterpret_cast<const float4*>(dx) + i, assign & 2);
        r.g = nk_load_stream(reinterpret_cast<const float4*>(g) + i, assign & 2);
        r.nz = MODE == 0 ? make_float4(1.f, 1.f, 1.f, 1.f) : nk_load_stream(reinterpret_cast<const float4*>(noise) + i, assign & 2);
        return r;
    }, [&](size_t i, const R& r) {
        float4 d = r.d;
        if (MODE == 0) { d.x += r.g.x; d.y += r.g.y; d.z += r.g.z; d.w += r.g.w; }
        else { d.x += r.g.x * r.nz.x; d.y += r.g.y * r.nz.y; d.z += r.g.z * r.nz.z; d.w += r.g.w * r.nz.w; }
        nk_store_stream(reinterpret_cast<float4*>(dx) + i, d);
    });
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        dx[i] = ((assign & 1) ? 0.f : dx[i]) + (MODE == 0 ? g[i] : g[i] * noise[i]);
    }
}

// ------------------------------------------------------------------ sub-block copies ----------
// `small` (contiguous, shape s[]) <-> the sub-block of `big` (shape b[]) starting at origin o[].
// DIR 0: small (=|+=) big[sub]      DIR 1: big[sub] (=|+=) small
struct Sub {
    int nd;
    int s[NK_MAX_DIMS];
    long long bstride[NK_MAX_DIMS];
    long long origin_off;
};

template <int DIR, bool ACC, bool VEC>
__global__ void subblock_kernel(float* __restrict__ small, float* __restrict__ big, Sub p, long long total) {
    constexpr int W = VEC ? 4 : 1;
    const long long groups = total / W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < groups;
         i += (long long)gridDim.x * blockDim.x) {
        long long rem = i * W, bo = p.origin_off;
#pragma unroll 1
        for (int d = p.nd - 1; d >= 0; --d) {
            const long long c = rem % p.s[d];
            rem /= p.s[d];
            bo += c * p.bstride[d];
        }
        if (VEC) {
            float4* sp = reinterpret_cast<float4*>(small + i * 4);
            float4* bp = reinterpret_cast<float4*>(big + bo);
            if (DIR == 0) {
                float4 v = *bp;
                if (ACC) { const float4 d = *sp; v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w; }
                nk_store_stream(sp, v);
            } else {
                float4 v = *sp;
                if (ACC) { const float4 d = *bp; v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w; }
                nk_store_stream(bp, v);
            }
        } else {
            if (DIR == 0) small[i] = ACC ? small[i] + big[bo] : big[bo];
            else big[bo] = ACC ? big[bo] + small[i] : small[i];
        }
    }
}

// i / d for 0 <= i < 2^24 through one f32 multiply and a +-1 correction (a 64-bit integer division per element
// is what kept the scalar copy kernels at ~2.3 TB/s).
__device__ __forceinline__ int fast_div(int i, int d, float inv_d) {
    int q = (int)((float)i * inv_d);
    const int r = i - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

// Unaligned sub-blocks (e.g. the centre of a padded gradient: origin 59, rows of 56 in rows of 58): a block walks
// one (H x W) plane at a time - the outer coordinates are decoded once per plane, the in-plane split is one
// fast_div - so both sides move W-float runs with consecutive lanes.
template <int DIR, bool ACC>
__global__ void subblock_plane_kernel(float* __restrict__ small, float* __restrict__ big, Sub p, long long planes, int H, int W,
                                      long long bs_h) {
    const int plane_elems = H * W;
    const float inv_w = 1.f / (float)W;
    for (long long pl = blockIdx.x; pl < planes; pl += gridDim.x) {
        long long rem = pl, bo = p.origin_off;
#pragma unroll 1
        for (int d = p.nd - 3; d >= 0; --d) {
            const long long c = rem % p.s[d];
            rem /= p.s[d];
            bo += c * p.bstride[d];
        }
        float* sp = small + pl * plane_elems;
        for (int i = threadIdx.x; i < plane_elems; i += blockDim.x) {
            const int h = fast_div(i, W, inv_w);
            float* bp = big + bo + h * bs_h + (i - h * W);
            if (DIR == 0) sp[i] = ACC ? sp[i] + *bp : *bp;
            else *bp = ACC ? *bp + sp[i] : sp[i];
        }
    }
}

// small_shape/big_shape/origin: nd entries each.
template <int DIR, bool ACC>
int subblock(nk_device* dev, float* small, const int* small_shape, float* big, const int* big_shape,
             const int* origin, int nd) {
    NK_USE(dev);
    NK_CHECK(nd >= 1 && nd <= NK_MAX_DIMS, "bad rank %d", nd);
    NK_CHECK(small && big, "null pointer");
    long long bst[NK_MAX_DIMS];
    long long acc = 1;
    for (int i = nd - 1; i >= 0; --i) { bst[i] = acc; acc *= big_shape[i]; }
    long long total = 1, off = 0;
    for (int i = 0; i < nd; ++i) {
        NK_CHECK(origin[i] >= 0 && origin[i] + small_shape[i] <= big_shape[i], "sub-block exceeds axis %d", i);
        total *= small_shape[i];
        off += origin[i] * bst[i];
    }
    if (total == 0) return NK_OK;
    // collapse dims where the sub-block spans the whole big axis
    Sub p{};
    int m = 0;
    for (int i = 0; i < nd; ++i) {
        if (m > 0 && small_shape[i] == big_shape[i]) {  // axis fully covered: merge into previous
            p.s[m - 1] *= small_shape[i];
            p.bstride[m - 1] = bst[i];
        } else {
            p.s[m] = small_shape[i];
            p.bstride[m] = bst[i];
            ++m;
        }
    }
    p.nd = m;
    p.origin_off = off;
    const bool vec = (p.s[m - 1] % 4 == 0) && (off % 4 == 0) && al16(small) && al16(big);
    bool strides_ok = true;
    for (int i = 0; i + 1 < m; ++i) if (p.bstride[i] % 4 != 0) strides_ok = false;
    const bool v = vec && strides_ok;
    if (!v && m >= 2 && (long long)p.s[m - 2] * p.s[m - 1] < (1 << 23) && p.bstride[m - 1] == 1) {
        const int H = p.s[m - 2], W = p.s[m - 1];
        const long long planes = total / ((long long)H * W);
        const int grid = (int)(planes < 8192 ? planes : 8192);
        hipLaunchKernelGGL((subblock_plane_kernel<DIR, ACC>), dim3(grid), dim3(256), 0, dev->compute, small, big, p, planes, H, W,
                           p.bstride[m - 2]);
        NK_LAUNCH_CHECK();
        return NK_OK;
    }
    const int grid = nk_stream_grid((size_t)(total / (v ? 4 : 1)), 256);
    if (v) hipLaunchKernelGGL((subblock_kernel<DIR, ACC, true>), dim3(grid), dim3(256), 0, dev->compute, small, big, p, total);
    else hipLaunchKernelGGL((subblock_kernel<DIR, ACC, false>), dim3(grid), dim3(256), 0, dev->compute, small, big, p, total);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

// ------------------------------------------------------------------ pad ------------------------
struct PadDesc {
    int nd;                  // spatial dims
    int in[3], out[3], pad[3];
};
// MODE 0: Constant/Zero (pad/constant/mod.rs:14-39), 1: Reflective (pad/reflective/mod.rs:9-136: left
// border reads index pad-i, right border 2(len-1)-(i-pad)), 2: Replicative (pad/replicative/mod.rs: clamp).
template <int MODE>
__global__ void pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, PadDesc p, long long planes,
                               long long out_plane, long long in_plane, float value) {
    const long long total = planes * out_plane;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long plane = i / out_plane;
        long long rem = i % out_plane, src = 0, mul = 1;
        bool inside = true;
        for (int d = p.nd - 1; d >= 0; --d) {
            int c = (int)(rem % p.out[d]) - p.pad[d];
            if (MODE == 0) {
                if (c < 0 || c >= p.in[d]) inside = false;
            } else if (MODE == 1) {
                c = c < 0 ? -c : (c >= p.in[d] ? 2 * (p.in[d] - 1) - c : c);
            } else {
                c = c < 0 ? 0 : (c >= p.in[d] ? p.in[d] - 1 : c);
            }
            rem /= p.out[d];
            src += c * mul;
            mul *= p.in[d];
        }
        y[i] = inside ? x[plane * in_plane + src] : value;
    }
}

// Same forward, one (N*C) plane per block iteration and 32-bit in-plane index math (fast_div): the generic kernel's
// 64-bit div/mod chain per element held it at ~2.2 TB/s.  Used when the padded plane has < 2^23 elements.
template <int MODE>
__global__ void pad_fwd_plane_kernel(const float* __restrict__ x, float* __restrict__ y, PadDesc p, long long planes,
                                     int out_plane, int in_plane, float value) {
    const int o1 = p.out[p.nd - 1], o2 = p.nd >= 2 ? p.out[p.nd - 2] : 1;
    const float inv1 = 1.f / (float)o1, inv2 = 1.f / (float)o2;
    for (long long pl = blockIdx.x; pl < planes; pl += gridDim.x) {
        const float* xp = x + pl * in_plane;
        float* yp = y + pl * out_plane;
        for (int i = threadIdx.x; i < out_plane; i += blockDim.x) {
            int c[3] = {0, 0, 0};
            int rem = fast_div(i, o1, inv1);
            c[p.nd - 1] = i - rem * o1;
            if (p.nd >= 2) {
                const int r2 = fast_div(rem, o2, inv2);
                c[p.nd - 2] = rem - r2 * o2;
                if (p.nd == 3) c[0] = r2;
            }
            int src = 0;
            bool inside = true;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (d < p.nd) {
                    int v = c[d] - p.pad[d];
                    if (MODE == 0) {
                        if (v < 0 || v >= p.in[d]) inside = false;
                    } else if (MODE == 1) {
                        v = v < 0 ? -v : (v >= p.in[d] ? 2 * (p.in[d] - 1) - v : v);
                    } else {
                        v = v < 0 ? 0 : (v >= p.in[d] ? p.in[d] - 1 : v);
                    }
                    src = src * p.in[d] + v;
                }
            }
            yp[i] = inside ? xp[src] : value;
        }
    }
}

// Constant / zero padding with an even padded row length (the common `k/2` padding of an even-width image): a thread
// writes TWO neighbouring outputs with one 8-byte store (a padded plane has an even element count, so every pair is 8-byte
// aligned) - half the store instructions of the element-wise plane kernel (C3: 58 -> ~40 us for 103 MB in, 110 MB out).
__global__ void pad_const_pairs_kernel(const float* __restrict__ x, float* __restrict__ y, PadDesc p, long long planes,
                                       int out_plane, int in_plane, float value) {
    const int o1 = p.out[p.nd - 1], h1 = o1 >> 1, o2 = p.nd >= 2 ? p.out[p.nd - 2] : 1;
    const float invh = 1.f / (float)h1, inv2 = 1.f / (float)o2;
    const int pairs = out_plane >> 1;
    for (long long pl = blockIdx.x; pl < planes; pl += gridDim.x) {
        const float* xp = x + pl * in_plane;
        float2* yp = reinterpret_cast<float2*>(y + pl * out_plane);
        for (int i = threadIdx.x; i < pairs; i += blockDim.x) {
            int c[3] = {0, 0, 0};
            int rem = fast_div(i, h1, invh);
            const int cw = (i - rem * h1) * 2 - p.pad[p.nd - 1];  // input column of the pair's first element
            if (p.nd >= 2) {
                const int r2 = fast_div(rem, o2, inv2);
                c[p.nd - 2] = rem - r2 * o2;
                if (p.nd == 3) c[0] = r2;
            }
            int src = 0;
            bool inside = true;
#pragma unroll
            for (int d = 0; d < 2; ++d) {  // the outer axes (the innermost one is handled per element below)
                if (d < p.nd - 1) {
                    const int v = c[d] - p.pad[d];
                    if (v < 0 || v >= p.in[d]) inside = false;
                    src = src * p.in[d] + (inside ? v : 0);
                }
            }
            const int w = p.in[p.nd - 1];
            src = src * w;
            const bool in0 = inside && cw >= 0 && cw < w, in1 = inside && cw + 1 >= 0 && cw + 1 < w;
            const float a = xp[in0 ? src + cw : 0], b = xp[in1 ? src + cw + 1 : 0];  // clamped addresses, masked values
            yp[i] = make_float2(in0 ? a : value, in1 ? b : value);
        }
    }
}

// ------------------------------------------------------------------ transpose -------------------
// 2-D: 32x32 tiles through LDS (+1 padding: conflict-free column reads), both sides coalesced.
template <bool ACC>
__global__ void transpose2d_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < R && c < C) tile[j][threadIdx.x] = in[(long long)r * C + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, r = r0 + threadIdx.x;  // out is C x R
        if (r < R && c < C) {
            float* q = &out[(long long)c * R + r];
            *q = ACC ? *q + tile[threadIdx.x][j] : tile[threadIdx.x][j];
        }
    }
}

// N-d reversed axes: out[i_{n-1},...,i_0] (=|+=) in[i_0,...,i_{n-1}].  Thread per out element.
struct Rev {
    int nd;
    int oshape[NK_MAX_DIMS];      // out shape = reversed in shape
    long long istride[NK_MAX_DIMS];  // stride in `in` of out axis d
};
template <bool ACC>
__global__ void transpose_nd_kernel(const float* __restrict__ in, float* __restrict__ out, Rev p, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long rem = i, src = 0;
        for (int d = p.nd - 1; d >= 0; --d) {
            src += (rem % p.oshape[d]) * p.istride[d];
            rem /= p.oshape[d];
        }
        out[i] = ACC ? out[i] + in[src] : in[src];
    }
}

template <bool ACC>
int transpose(nk_device* dev, const float* in, float* out, const int* in_shape, int nd) {
    NK_USE(dev);
    NK_CHECK(nd >= 1 && nd <= NK_MAX_DIMS, "bad rank %d", nd);
    const long long total = (long long)nk_numel(in_shape, nd);
    if (total == 0) return NK_OK;
    NK_CHECK(in && out, "null pointer");
    if (nd == 2) {
        const int R = in_shape[0], C = in_shape[1];
        hipLaunchKernelGGL((transpose2d_kernel<ACC>), dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, dev->compute, in, out, R, C);
    } else {
        Rev p{};
        p.nd = nd;
        long long st[NK_MAX_DIMS], acc = 1;
        for (int i = nd - 1; i >= 0; --i) { st[i] = acc; acc *= in_shape[i]; }
        for (int d = 0; d < nd; ++d) { p.oshape[d] = in_shape[nd - 1 - d]; p.istride[d] = st[nd - 1 - d]; }
        hipLaunchKernelGGL((transpose_nd_kernel<ACC>), dim3(nk_stream_grid((size_t)total, 256)), dim3(256), 0, dev->compute, in, out, p, total);
    }
    NK_LAUNCH_CHECK();
    return NK_OK;
}

// ------------------------------------------------------------------ head split / merge ----------
// flat[(b*S + s)*(H*dh) + h*dh + e]  <->  heads[((b*H + h)*S + s)*dh + e]
// TO_HEADS: heads (=|+=) flat ; else flat (=|+=) heads.  dh % 4 == 0 -> float4.
template <bool TO_HEADS, bool ACC, bool VEC>
__global__ void heads_kernel(float* __restrict__ flat, float* __restrict__ heads, int B, int S, int H, int dh) {
    constexpr int W = VEC ? 4 : 1;
    const long long total = (long long)B * S * H * dh / W;
    const int dq = dh / W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        // i indexes the flat layout (coalesced on the flat side, dh-contiguous on the heads side)
        const int e = (int)(i % dq);
        long long rem = i / dq;
        const int h = (int)(rem % H); rem /= H;
        const int s = (int)(rem % S);
        const int b = (int)(rem / S);
        const long long fo = i * W;
        const long long ho = (((long long)(b * H + h) * S + s) * dh) + e * W;
        if (VEC) {
            float4* fp = reinterpret_cast<float4*>(flat + fo);
            float4* hp = reinterpret_cast<float4*>(heads + ho);
            float4 v = TO_HEADS ? *fp : *hp;
            float4* dst = TO_HEADS ? hp : fp;
            if (ACC) { const float4 d = *dst; v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w; }
            nk_store_stream(dst, v);
        } else {
            const float v = TO_HEADS ? flat[fo] : heads[ho];
            float* dst = TO_HEADS ? &heads[ho] : &flat[fo];
            *dst = ACC ? *dst + v : v;
        }
    }
}

template <bool TO_HEADS, bool ACC>
int heads(nk_device* dev, float* flat, float* hd, int B, int S, int H, int dh) {
    NK_USE(dev);
    NK_CHECK(B >= 0 && S >= 0 && H >= 0 && dh >= 0, "negative extent");
    const long long total = (long long)B * S * H * dh;
    if (total == 0) return NK_OK;
    NK_CHECK(flat && hd, "null pointer");
    const bool vec = (dh % 4 == 0) && al16(flat) && al16(hd);
    const int grid = nk_stream_grid((size_t)(total / (vec ? 4 : 1)), 256);
    if (vec) hipLaunchKernelGGL((heads_kernel<TO_HEADS, ACC, true>), dim3(grid), dim3(256), 0, dev->compute, flat, hd, B, S, H, dh);
    else hipLaunchKernelGGL((heads_kernel<TO_HEADS, ACC, false>), dim3(grid), dim3(256), 0, dev->compute, flat, hd, B, S, H, dh);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

int chunk_origin(const int* x_shape, const int* chunk_shape, int nd, int chunk_no, int* origin) {
    NK_CHECK(nd >= 1 && nd <= NK_MAX_DIMS, "bad rank %d", nd);
    long long grid_total = 1;
    int grid[NK_MAX_DIMS];
    for (int i = 0; i < nd; ++i) {
        NK_CHECK(chunk_shape[i] >= 1 && chunk_shape[i] <= x_shape[i], "chunk axis %d: %d does not fit %d", i, chunk_shape[i], x_shape[i]);
        grid[i] = x_shape[i] / chunk_shape[i];  // exact_chunks: remainder skipped
        grid_total *= grid[i];
    }
    NK_CHECK(chunk_no >= 0 && chunk_no < grid_total, "chunk_no %d out of range (%lld chunks)", chunk_no, grid_total);
    int rem = chunk_no;
    for (int i = nd - 1; i >= 0; --i) { origin[i] = (rem % grid[i]) * chunk_shape[i]; rem /= grid[i]; }
    return NK_OK;
}

}  // namespace

template <int MODE>
static int pad_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, float* y, const int* padding, float value) {
    NK_USE(dev);
    NK_CHECK(nd >= 1 && nd <= 3, "pad supports 1-3 spatial dims, got %d", nd);
    PadDesc p{};
    p.nd = nd;
    long long in_plane = 1, out_plane = 1;
    for (int i = 0; i < nd; ++i) {
        NK_CHECK(padding[i] >= 0, "negative padding");
        // the reference indexes out of bounds (panics) past one reflection / on an empty axis
        NK_CHECK(MODE != 1 || padding[i] == 0 || padding[i] < x_shape[2 + i], "reflective padding %d needs input extent > padding (got %d)", padding[i], x_shape[2 + i]);
        NK_CHECK(MODE != 2 || padding[i] == 0 || x_shape[2 + i] > 0, "replicative padding of an empty axis");
        p.in[i] = x_shape[2 + i]; p.pad[i] = padding[i]; p.out[i] = x_shape[2 + i] + 2 * padding[i];
        in_plane *= p.in[i]; out_plane *= p.out[i];
    }
    const long long planes = (long long)x_shape[0] * x_shape[1];
    if (planes * out_plane == 0) return NK_OK;
    NK_CHECK(x && y, "null pointer in pad forward");
    if (out_plane < (1 << 23)) {
        if (MODE == 0 && p.out[nd - 1] % 2 == 0 && in_plane > 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
            hipLaunchKernelGGL(pad_const_pairs_kernel, dim3((unsigned)(planes < 8192 ? planes : 8192)), dim3(256), 0, dev->compute, x, y,
                               p, planes, (int)out_plane, (int)in_plane, value);
            NK_LAUNCH_CHECK();
            return NK_OK;
        }
        hipLaunchKernelGGL(pad_fwd_plane_kernel<MODE>, dim3((unsigned)(planes < 8192 ? planes : 8192)), dim3(256), 0, dev->compute, x,
                           y, p, planes, (int)out_plane, (int)in_plane, value);
        NK_LAUNCH_CHECK();
        return NK_OK;
    }
    hipLaunchKernelGGL(pad_fwd_kernel<MODE>, dim3(nk_stream_grid((size_t)(planes * out_plane), 256)), dim3(256), 0, dev->compute, x,
                       y, p, planes, out_plane, in_plane, value);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

extern "C" {

int nk_dropout_fwd(nk_device* dev, const float* x, float* y, float* noise, size_t n, double p, int train,
                   uint64_t seed, uint64_t offset) {
    NK_USE(dev);
    NK_CHECK(p >= 0.0 && p <= 1.0, "Wrong probability received: %g.", p);
    if (n == 0) return NK_OK;
    NK_CHECK(x && y, "null pointer in nk_dropout_fwd");
    if (!train || p == 0.0) {
        NK_HIP(hipMemcpyAsync(y, x, n * sizeof(float), hipMemcpyDeviceToDevice, dev->compute));
        return NK_OK;
    }
    if (1.0 - p == 0.0) {
        NK_HIP(hipMemsetAsync(y, 0, n * sizeof(float), dev->compute));
        return NK_OK;
    }
    NK_CHECK(noise != nullptr, "noise buffer required in training mode");
    NK_CHECK(al16(x) && al16(y) && al16(noise), "dropout buffers must be 16-byte aligned");
    if (int rc = nk_refuse_capture(dev, "nk_dropout_fwd: the Philox offset (every replay would draw the same mask)",
                                   "run the training-mode dropout eagerly, or capture the evaluation graph")) return rc;
    const unsigned keep_lt = nk_keep_threshold(1.0 - p);   // Bernoulli::new(1. - p), dropout/mod.rs:46
    const float scale = 1.f - (float)p;                    // `(1. - self.p as f32)`, dropout/mod.rs:76
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(nk_stream_grid((n + 3) / 4, 256)), dim3(256), 0, dev->compute, x, y, noise,
                       n, keep_lt, scale, (unsigned long long)seed, (unsigned long long)offset);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

static int dropout_bwd(nk_device* dev, float* dx, const float* g, const float* noise, size_t n, double p, int train, int assign) {
    NK_USE(dev);
    NK_CHECK(p >= 0.0 && p <= 1.0, "Wrong probability received: %g.", p);
    if (n == 0) return NK_OK;
    NK_CHECK(dx && g, "null pointer in nk_dropout_bwd");
    NK_CHECK(al16(dx) && al16(g), "dropout buffers must be 16-byte aligned");
    const int grid = nk_stream_grid(n / 4 + 1, 256);
    assign |= nk_streams_past_cache(n * 16) ? 2 : 0;  // bit 1: `nt` loads (g, noise, dx beyond the Infinity Cache)
    if (!train || p == 0.0) {
        hipLaunchKernelGGL((dropout_bwd_kernel<0>), dim3(grid), dim3(256), 0, dev->compute, dx, g, noise, n, assign);
    } else {
        NK_CHECK(noise != nullptr && al16(noise), "noise buffer required in training mode");
        hipLaunchKernelGGL((dropout_bwd_kernel<1>), dim3(grid), dim3(256), 0, dev->compute, dx, g, noise, n, assign);
    }
    NK_LAUNCH_CHECK();
    return NK_OK;
}
int nk_dropout_bwd(nk_device* dev, float* dx, const float* g, const float* noise, size_t n, double p, int train) {
    return dropout_bwd(dev, dx, g, noise, n, p, train, 0);
}
int nk_dropout_bwd_assign(nk_device* dev, float* dx, const float* g, const float* noise, size_t n, double p, int train) {
    return dropout_bwd(dev, dx, g, noise, n, p, train, 1);
}

int nk_pad_const_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, float* y, const int* padding, float value) {
    return pad_fwd<0>(dev, nd, x, x_shape, y, padding, value);
}
int nk_pad_reflective_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, float* y, const int* padding) {
    return pad_fwd<1>(dev, nd, x, x_shape, y, padding, 0.f);
}
int nk_pad_replicative_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, float* y, const int* padding) {
    return pad_fwd<2>(dev, nd, x, x_shape, y, padding, 0.f);
}

int nk_pad_bwd(nk_device* dev, int nd, float* dx, const int* x_shape, const float* g, const int* padding) {
    NK_CHECK(nd >= 1 && nd <= 3, "pad supports 1-3 spatial dims, got %d", nd);
    int gshape[5], origin[5] = {0, 0, 0, 0, 0};
    gshape[0] = x_shape[0]; gshape[1] = x_shape[1];
    for (int i = 0; i < nd; ++i) { gshape[2 + i] = x_shape[2 + i] + 2 * padding[i]; origin[2 + i] = padding[i]; }
    return subblock<0, true>(dev, dx, x_shape, const_cast<float*>(g), gshape, origin, nd + 2);  // dx += g[centre]
}
int nk_pad_bwd_assign(nk_device* dev, int nd, float* dx, const int* x_shape, const float* g, const int* padding) {
    NK_CHECK(nd >= 1 && nd <= 3, "pad supports 1-3 spatial dims, got %d", nd);
    int gshape[5], origin[5] = {0, 0, 0, 0, 0};
    gshape[0] = x_shape[0]; gshape[1] = x_shape[1];
    for (int i = 0; i < nd; ++i) { gshape[2 + i] = x_shape[2 + i] + 2 * padding[i]; origin[2 + i] = padding[i]; }
    return subblock<0, false>(dev, dx, x_shape, const_cast<float*>(g), gshape, origin, nd + 2);  // dx = g[centre]
}

int nk_chunk_fwd(nk_device* dev, const float* x, const int* x_shape, float* y, const int* chunk_shape, int nd, int chunk_no) {
    int origin[NK_MAX_DIMS];
    int rc = chunk_origin(x_shape, chunk_shape, nd, chunk_no, origin);
    if (rc) return rc;
    return subblock<0, false>(dev, y, chunk_shape, const_cast<float*>(x), x_shape, origin, nd);
}

int nk_chunk_bwd(nk_device* dev, float* dx, const int* x_shape, const float* g, const int* chunk_shape, int nd, int chunk_no) {
    int origin[NK_MAX_DIMS];
    int rc = chunk_origin(x_shape, chunk_shape, nd, chunk_no, origin);
    if (rc) return rc;
    return subblock<1, true>(dev, const_cast<float*>(g), chunk_shape, dx, x_shape, origin, nd);
}

int nk_concat_fwd_part(nk_device* dev, const float* operand, float* out, const int* out_shape, int nd, int axis,
                       int offset, int op_len) {
    NK_CHECK(nd >= 1 && nd <= NK_MAX_DIMS && axis >= 0 && axis < nd, "bad rank/axis");
    int s[NK_MAX_DIMS], origin[NK_MAX_DIMS] = {0};
    for (int i = 0; i < nd; ++i) s[i] = out_shape[i];
    s[axis] = op_len;
    origin[axis] = offset;
    return subblock<1, false>(dev, const_cast<float*>(operand), s, out, out_shape, origin, nd);
}

int nk_concat_bwd_part(nk_device* dev, float* d_operand, const float* g, const int* g_shape, int nd, int axis,
                       int offset, int op_len) {
    NK_CHECK(nd >= 1 && nd <= NK_MAX_DIMS && axis >= 0 && axis < nd, "bad rank/axis");
    int s[NK_MAX_DIMS], origin[NK_MAX_DIMS] = {0};
    for (int i = 0; i < nd; ++i) s[i] = g_shape[i];
    s[axis] = op_len;
    origin[axis] = offset;
    return subblock<0, true>(dev, d_operand, s, const_cast<float*>(g), g_shape, origin, nd);
}
int nk_concat_bwd_part_assign(nk_device* dev, float* d_operand, const float* g, const int* g_shape, int nd, int axis,
                              int offset, int op_len) {
    NK_CHECK(nd >= 1 && nd <= NK_MAX_DIMS && axis >= 0 && axis < nd, "bad rank/axis");
    int s[NK_MAX_DIMS], origin[NK_MAX_DIMS] = {0};
    for (int i = 0; i < nd; ++i) s[i] = g_shape[i];
    s[axis] = op_len;
    origin[axis] = offset;
    return subblock<0, false>(dev, d_operand, s, const_cast<float*>(g), g_shape, origin, nd);
}

int nk_transpose_fwd(nk_device* dev, const float* x, float* y, const int* x_shape, int nd) {
    return transpose<false>(dev, x, y, x_shape, nd);
}
int nk_transpose_bwd(nk_device* dev, float* dx, const float* g, const int* x_shape, int nd) {
    // g has the reversed shape; dx += g^T
    int gs[NK_MAX_DIMS];
    NK_CHECK(nd >= 1 && nd <= NK_MAX_DIMS, "bad rank %d", nd);
    for (int i = 0; i < nd; ++i) gs[i] = x_shape[nd - 1 - i];
    return transpose<true>(dev, g, dx, gs, nd);
}
int nk_transpose_bwd_assign(nk_device* dev, float* dx, const float* g, const int* x_shape, int nd) {
    int gs[NK_MAX_DIMS];
    NK_CHECK(nd >= 1 && nd <= NK_MAX_DIMS, "bad rank %d", nd);
    for (int i = 0; i < nd; ++i) gs[i] = x_shape[nd - 1 - i];
    return transpose<false>(dev, g, dx, gs, nd);
}

int nk_split_heads_fwd(nk_device* dev, const float* x, float* y, int B, int S, int H, int dh) {
    return heads<true, false>(dev, const_cast<float*>(x), y, B, S, H, dh);
}
int nk_split_heads_bwd(nk_device* dev, float* dx, const float* g, int B, int S, int H, int dh) {
    return heads<false, true>(dev, dx, const_cast<float*>(g), B, S, H, dh);
}
int nk_split_heads_bwd_assign(nk_device* dev, float* dx, const float* g, int B, int S, int H, int dh) {
    return heads<false, false>(dev, dx, const_cast<float*>(g), B, S, H, dh);
}
int nk_merge_heads_fwd(nk_device* dev, const float* x, float* y, int B, int S, int H, int dh) {
    return heads<false, false>(dev, y, const_cast<float*>(x), B, S, H, dh);
}
int nk_merge_heads_bwd(nk_device* dev, float* dx, const float* g, int B, int S, int H, int dh) {
    return heads<true, true>(dev, const_cast<float*>(g), dx, B, S, H, dh);
}
int nk_merge_heads_bwd_assign(nk_device* dev, float* dx, const float* g, int B, int S, int H, int dh) {
    return heads<true, false>(dev, const_cast<float*>(g), dx, B, S, H, dh);
}

}  // extern "C"
