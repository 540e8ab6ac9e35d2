// HBM-bound elementwise kernels of the hot path (bound: 8 TB/s HBM3E; every kernel moves
// 16 B per lane per access, grid-stride, no LDS):
//   broadcast binaries fwd  — node/{addition,subtraction,multiplication,division}/mod.rs:39-50
//   their backward          — local gradient fused with the un-broadcast reduction
//                             (`utils::accumulate`, utils.rs:152-192, intended semantics)
//   ReLU fwd/bwd            — node/relu/mod.rs:29-38, 67-79
//   fill / SGD step         — vardiff.rs:100-102,133 ; neuronika-optim/src/sgd/mod.rs:186-236
#include "nk_common.h"

namespace {

// Broadcast descriptor: the output index space, right-aligned and collapsed, with the element
// stride of each operand along every collapsed dim (0 where the operand is broadcast).
struct Bcast {
    int nd;
    int shape[NK_MAX_DIMS];
    long long ls[NK_MAX_DIMS];
    long long rs[NK_MAX_DIMS];
};

// strides of `shape` (nd dims, right-aligned into out_nd dims) in the out index space
void bstrides(const int* shape, int nd, const int* out_shape, int out_nd, long long* st) {
    (void)out_shape;
    long long acc = 1;
    for (int i = out_nd - 1; i >= 0; --i) {
        const int j = i - (out_nd - nd);
        if (j < 0) { st[i] = 0; continue; }
        st[i] = shape[j] == 1 ? 0 : acc;  // extent-1 (broadcast) axes never advance
        acc *= shape[j];
    }
}

int check_bcast(const int* out_shape, int out_nd, const int* s, int nd, const char* who) {
    NK_CHECK(nd <= out_nd && out_nd <= NK_MAX_DIMS, "%s: rank %d > output rank %d (max %d)", who, nd, out_nd, NK_MAX_DIMS);
    for (int j = 0; j < nd; ++j) {
        const int o = out_shape[j + out_nd - nd];
        NK_CHECK(s[j] == o || s[j] == 1, "The two tensors have incompatible shape. (%s axis %d: %d vs %d)", who, j, s[j], o);
    }
    return NK_OK;
}

// Collapse adjacent dims that are mergeable for every stride vector in `sts` (nvec vectors).
int collapse(const int* shape, int nd, long long** sts, int nvec, int* oshape) {
    int m = 0;
    for (int i = 0; i < nd; ++i) {
        if (shape[i] == 1) continue;  // drop unit dims
        bool merged = false;
        if (m > 0) {
            bool ok = true;
            for (int v = 0; v < nvec; ++v) {
                const long long prev = sts[v][m - 1], cur = sts[v][i];
                if (!((prev == 0 && cur == 0) || (cur != 0 && prev == cur * shape[i]))) ok = false;
            }
            if (ok) {
                oshape[m - 1] *= shape[i];
                for (int v = 0; v < nvec; ++v) sts[v][m - 1] = sts[v][i];
                merged = true;
            }
        }
        if (!merged) {
            oshape[m] = shape[i];
            for (int v = 0; v < nvec; ++v) sts[v][m] = sts[v][i];
            ++m;
        }
    }
    if (m == 0) {
        oshape[0] = 1;
        for (int v = 0; v < nvec; ++v) sts[v][0] = 0;
        m = 1;
    }
    return m;
}

template <int OP>
__device__ __forceinline__ float bin(float l, float r) {
    return OP == NK_ADD ? l + r : OP == NK_SUB ? l - r : OP == NK_MUL ? l * r : l / r;
}

__device__ __forceinline__ float4 ld4(const float* p, long long off, long long inner_stride) {
    if (inner_stride == 1) return *reinterpret_cast<const float4*>(p + off);
    const float v = p[off];
    return make_float4(v, v, v, v);
}

// out[i] = l (op) r over the collapsed index space; 4 consecutive inner elements per thread
// when VEC (inner extent % 4 == 0, operands' inner strides in {0,1}, 16-B aligned).
template <int OP, bool VEC>
__global__ void binary_fwd_kernel(float* __restrict__ out, const float* __restrict__ l,
                                  const float* __restrict__ r, Bcast b, long long total) {
    constexpr int W = VEC ? 4 : 1;
    const long long groups = total / W;
    struct R { float4 a, c; };
    nk_span_walk<VEC ? 4 : 1>((size_t)groups, [&](size_t g) {
        long long rem = (long long)g * W, lo = 0, ro = 0;
#pragma unroll 1
        for (int d = b.nd - 1; d >= 0; --d) {
            const long long c = rem % b.shape[d];
            rem /= b.shape[d];
            lo += c * b.ls[d];
            ro += c * b.rs[d];
        }
        R q;
        if (VEC) {
            q.a = ld4(l, lo, b.ls[b.nd - 1]); q.c = ld4(r, ro, b.rs[b.nd - 1]);
        } else {
            q.a = make_float4(l[lo], 0.f, 0.f, 0.f); q.c = make_float4(r[ro], 0.f, 0.f, 0.f);
        }
        return q;
    }, [&](size_t g, const R& q) {
        if (VEC) {
            float4 o;
            o.x = bin<OP>(q.a.x, q.c.x); o.y = bin<OP>(q.a.y, q.c.y); o.z = bin<OP>(q.a.z, q.c.z); o.w = bin<OP>(q.a.w, q.c.w);
            nk_store_stream(reinterpret_cast<float4*>(out + g * 4), o);
        } else {
            out[g] = bin<OP>(q.a.x, q.c.x);
        }
    });
}

// ---- backward: local gradient -------------------------------------------------------------
// MODE 0: g        (add left/right, sub left)
// MODE 1: -g       (sub right)
// MODE 2: g * o    (mul left: o = r ; mul right: o = l)
// MODE 3: g / o    (div left: o = r)
// MODE 4: -g*l/r^2 (div right: o = l, q = r)
template <int MODE>
__device__ __forceinline__ float local_grad(float g, float o, float q) {
    return MODE == 0 ? g : MODE == 1 ? -g : MODE == 2 ? g * o : MODE == 3 ? g / o : -g * o / (q * q);
}

// No reduction needed (operand shape == gradient shape): d += local, 16 B per lane.
// o / q follow their own (collapsed) broadcast strides.
struct Bcast3 {
    int nd;
    int shape[NK_MAX_DIMS];
    long long os[NK_MAX_DIMS];
    long long qs[NK_MAX_DIMS];
};

template <int MODE, bool VEC>
__global__ void binary_bwd_same_kernel(float* __restrict__ d, const float* __restrict__ g,
                                       const float* __restrict__ o, const float* __restrict__ q,
                                       Bcast3 b, long long total, int assign) {
    constexpr int W = VEC ? 4 : 1;
    const long long groups = total / W;
    struct R { float4 g, d, o, q; };
    nk_span_walk<VEC ? 4 : 1>((size_t)groups, [&](size_t i) {
        long long oo = 0, qo = 0;
        if (MODE >= 2) {
            long long rem = (long long)i * W;
#pragma unroll 1
            for (int k = b.nd - 1; k >= 0; --k) {
                const long long c = rem % b.shape[k];
                rem /= b.shape[k];
                oo += c * b.os[k];
                qo += c * b.qs[k];
            }
        }
        R r;
        r.o = make_float4(0, 0, 0, 0); r.q = make_float4(1, 1, 1, 1);
        if (VEC) {
            // bit 1 of `assign`: `nt` loads of g and d (operands beyond the Infinity Cache, nk_common.h)
            r.g = nk_load_stream(reinterpret_cast<const float4*>(g + i * 4), assign & 2);
            r.d = (assign & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : nk_load_stream(reinterpret_cast<const float4*>(d + i * 4), assign & 2);
            if (MODE >= 2) r.o = ld4(o, oo, b.os[b.nd - 1]);
            if (MODE == 4) r.q = ld4(q, qo, b.qs[b.nd - 1]);
        } else {
            r.g = make_float4(g[i], 0.f, 0.f, 0.f);
            r.d = make_float4((assign & 1) ? 0.f : d[i], 0.f, 0.f, 0.f);
            if (MODE >= 2) r.o.x = o[oo];
            if (MODE == 4) r.q.x = q[qo];
        }
        return r;
    }, [&](size_t i, const R& r) {
        if (VEC) {
            float4 dv = r.d;
            dv.x += local_grad<MODE>(r.g.x, r.o.x, r.q.x);
            dv.y += local_grad<MODE>(r.g.y, r.o.y, r.q.y);
            dv.z += local_grad<MODE>(r.g.z, r.o.z, r.q.z);
            dv.w += local_grad<MODE>(r.g.w, r.o.w, r.q.w);
            nk_store_stream(reinterpret_cast<float4*>(d + i * 4), dv);
        } else {
            d[i] = r.d.x + local_grad<MODE>(r.g.x, r.o.x, r.q.x);
        }
    });
}

// Reduction over an index space viewed as [R0][K][R1]: part[chunk][k] = sum over the chunk's
// r0 rows and all r1 of local(r0,k,r1).  g is contiguous; o/q have strides (s0,sk,s1).
struct Rkr {
    int R0, K, R1;
    long long os0, osk, os1, qs0, qsk, qs1;
    int rows_per_chunk;
    int vec;  // float4 path allowed (R1 % 4 == 0, 16-B aligned bases, unit/zero inner strides)
};

// R1 == 1: column reduction.  Threads along k (coalesced 256-B rows per wave); blockIdx.y
// splits R0.  Deterministic: fixed chunking, fixed order.
template <int MODE>
__global__ void reduce_cols_kernel(float* __restrict__ part, const float* __restrict__ g,
                                   const float* __restrict__ o, const float* __restrict__ q, Rkr p) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.K) return;
    const int r_beg = blockIdx.y * p.rows_per_chunk;
    const int r_end = min(p.R0, r_beg + p.rows_per_chunk);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int r = r_beg;
    for (; r + 3 < r_end; r += 4) {
#define NK_TERM(RR) local_grad<MODE>(g[(long long)(RR) * p.K + k], MODE >= 2 ? o[(RR) * p.os0 + k * p.osk] : 0.f, \
                                     MODE == 4 ? q[(RR) * p.qs0 + k * p.qsk] : 1.f)
        acc0 += NK_TERM(r);
        acc1 += NK_TERM(r + 1);
        acc2 += NK_TERM(r + 2);
        acc3 += NK_TERM(r + 3);
    }
    for (; r < r_end; ++r) acc0 += NK_TERM(r);
#undef NK_TERM
    part[(long long)blockIdx.y * p.K + k] = (acc0 + acc1) + (acc2 + acc3);
}

// Column reduction, wide form (K % 4 == 0, no operand data needed: add / sub gradients, i.e. the
// bias gradient of `x.mm_t(W) + b`): a block is 64 column-quads x 4 row-lanes, every lane moves
// 16 B per row (one wave = 1 KiB contiguous), the 4 row-lanes are folded through LDS.
template <int MODE>
__global__ void reduce_cols4_kernel(float* __restrict__ part, const float* __restrict__ g, Rkr p) {
    __shared__ float4 red[4][64];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int k = (blockIdx.x * 64 + cq) * 4;
    const int r_beg = blockIdx.y * p.rows_per_chunk;
    const int r_end = min(p.R0, r_beg + p.rows_per_chunk);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < p.K) {
        int r = r_beg + rl;
        // eight rows per trip, all eight loads issued before the first add (a chunk is ~16 rows per lane: with two loads per
        // trip the lane paid eight serial memory round trips - 13 us for the 64 MB of a C4 bias gradient, 3.5 - 4.6 TB/s);
        // the order of the adds is the two-load loop's: even rows into a, odd rows into b
        for (; r + 28 < r_end; r += 32) {
            float4 u[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) u[i] = *reinterpret_cast<const float4*>(g + (long long)(r + 4 * i) * p.K + k);
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                a.x += u[i].x; a.y += u[i].y; a.z += u[i].z; a.w += u[i].w;
                b.x += u[i + 1].x; b.y += u[i + 1].y; b.z += u[i + 1].z; b.w += u[i + 1].w;
            }
        }
        for (; r + 4 < r_end; r += 8) {
            const float4 u = *reinterpret_cast<const float4*>(g + (long long)r * p.K + k);
            const float4 v = *reinterpret_cast<const float4*>(g + (long long)(r + 4) * p.K + k);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        for (; r < r_end; r += 4) {
            const float4 u = *reinterpret_cast<const float4*>(g + (long long)r * p.K + k);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
    }
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    red[rl][cq] = a;
    __syncthreads();
    if (rl == 0 && k < p.K) {
        float4 s = red[0][cq];
#pragma unroll
        for (int i = 1; i < 4; ++i) { s.x += red[i][cq].x; s.y += red[i][cq].y; s.z += red[i][cq].z; s.w += red[i][cq].w; }
        if (MODE == 1) { s.x = -s.x; s.y = -s.y; s.z = -s.z; s.w = -s.w; }
        *reinterpret_cast<float4*>(part + (long long)blockIdx.y * p.K + k) = s;
    }
}

// R1 > 1: one block per (k, chunk); threads stride over (r0 in chunk) x r1 with r1 contiguous.
template <int MODE>
__global__ void reduce_rkr_kernel(float* __restrict__ part, const float* __restrict__ g,
                                  const float* __restrict__ o, const float* __restrict__ q, Rkr p) {
    __shared__ float red[4];
    const int k = blockIdx.x;
    const int r_beg = blockIdx.y * p.rows_per_chunk;
    const int r_end = min(p.R0, r_beg + p.rows_per_chunk);
    float acc = 0.f;
    const bool vec = p.vec != 0;
    for (int r = r_beg; r < r_end; ++r) {
        const float* gp = g + ((long long)r * p.K + k) * p.R1;
        const long long ob = r * p.os0 + k * p.osk, qb = r * p.qs0 + k * p.qsk;
        if (vec) {
            for (int c = threadIdx.x * 4; c < p.R1; c += blockDim.x * 4) {
                const float4 gv = *reinterpret_cast<const float4*>(gp + c);
                float4 ov = make_float4(0, 0, 0, 0), qv = make_float4(1, 1, 1, 1);
                if (MODE >= 2) ov = ld4(o, ob + c * p.os1, p.os1);
                if (MODE == 4) qv = ld4(q, qb + c * p.qs1, p.qs1);
                acc += (local_grad<MODE>(gv.x, ov.x, qv.x) + local_grad<MODE>(gv.y, ov.y, qv.y)) +
                       (local_grad<MODE>(gv.z, ov.z, qv.z) + local_grad<MODE>(gv.w, ov.w, qv.w));
            }
        } else {
            for (int c = threadIdx.x; c < p.R1; c += blockDim.x)
                acc += local_grad<MODE>(gp[c], MODE >= 2 ? o[ob + c * p.os1] : 0.f,
                                        MODE == 4 ? q[qb + c * p.qs1] : 1.f);
        }
    }
    acc = nk_block_sum<256>(acc, red);
    if (threadIdx.x == 0) part[(long long)blockIdx.y * p.K + k] = acc;
}

// d[k] += sum_c part[c][k].  A block is 64 columns x 4 chunk-lanes (lane j sums chunks j, j+4, ... with four
// independent accumulators), folded through LDS in a fixed order: K/64 blocks and chunks/4 loads deep instead of
// K/256 blocks and `chunks` dependent loads deep (that serial form cost 20-60 us for a few hundred KB).
__global__ void reduce_finish_kernel(float* __restrict__ d, const float* __restrict__ part, int K, int chunks, int assign) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (k < K) {
        int c = lane;
        for (; c + 12 < chunks; c += 16) {
            s0 += part[(long long)c * K + k];
            s1 += part[(long long)(c + 4) * K + k];
            s2 += part[(long long)(c + 8) * K + k];
            s3 += part[(long long)(c + 12) * K + k];
        }
        for (; c < chunks; c += 4) s0 += part[(long long)c * K + k];
    }
    red[lane][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (lane == 0 && k < K) d[k] = (assign ? 0.f : d[k]) + ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col]));
}

// Fallback for reduction patterns with more than three collapsed groups: one thread per kept
// element, sequential over the reduced index space.  Correct for any broadcast; slow.
struct Gen {
    int nd;
    int shape[NK_MAX_DIMS];
    long long ds[NK_MAX_DIMS];  // target strides (0 on reduced dims)
    long long os[NK_MAX_DIMS];
    long long qs[NK_MAX_DIMS];
    long long gs[NK_MAX_DIMS];
};
template <int MODE>
__global__ void reduce_generic_kernel(float* __restrict__ d, const float* __restrict__ g,
                                      const float* __restrict__ o, const float* __restrict__ q, Gen p,
                                      long long kept, long long reduced, int assign) {
    for (long long ki = blockIdx.x * (long long)blockDim.x + threadIdx.x; ki < kept;
         ki += (long long)gridDim.x * blockDim.x) {
        // decode kept coordinate
        long long rem = ki, dofs = 0, gofs = 0, oofs = 0, qofs = 0;
        for (int a = p.nd - 1; a >= 0; --a) {
            if (p.ds[a] == 0) continue;
            const long long c = rem % p.shape[a];
            rem /= p.shape[a];
            dofs += c * p.ds[a]; gofs += c * p.gs[a]; oofs += c * p.os[a]; qofs += c * p.qs[a];
        }
        float acc = 0.f;
        for (long long ri = 0; ri < reduced; ++ri) {
            long long rr = ri, go = gofs, oo = oofs, qo = qofs;
            for (int a = p.nd - 1; a >= 0; --a) {
                if (p.ds[a] != 0) continue;
                const long long c = rr % p.shape[a];
                rr /= p.shape[a];
                go += c * p.gs[a]; oo += c * p.os[a]; qo += c * p.qs[a];
            }
            acc += local_grad<MODE>(g[go], MODE >= 2 ? o[oo] : 0.f, MODE == 4 ? q[qo] : 1.f);
        }
        d[dofs] = (assign ? 0.f : d[dofs]) + acc;
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// d_target(t_shape) += unbroadcast( local<MODE>(g, o, q) ) over g_shape.
template <int MODE>
int bwd_dispatch(nk_device* dev, float* d, const int* t_shape, int t_nd, const float* g, const int* g_shape,
                 int g_nd, const float* o, const int* o_shape, int o_nd, const float* q, const int* q_shape,
                 int q_nd, int assign = 0) {
    NK_USE(dev);
    NK_CHECK(d && g, "null gradient pointer");
    int rc = check_bcast(g_shape, g_nd, t_shape, t_nd, "target");
    if (rc) return rc;
    if (MODE >= 2) { NK_CHECK(o, "operand data required"); rc = check_bcast(g_shape, g_nd, o_shape, o_nd, "operand"); if (rc) return rc; }
    if (MODE == 4) { NK_CHECK(q, "operand data required"); rc = check_bcast(g_shape, g_nd, q_shape, q_nd, "operand"); if (rc) return rc; }
    const long long total = (long long)nk_numel(g_shape, g_nd);
    if (total == 0) return NK_OK;

    long long ts[NK_MAX_DIMS], os[NK_MAX_DIMS] = {0}, qs[NK_MAX_DIMS] = {0}, gs[NK_MAX_DIMS];
    bstrides(t_shape, t_nd, g_shape, g_nd, ts);
    bstrides(g_shape, g_nd, g_shape, g_nd, gs);
    if (MODE >= 2) bstrides(o_shape, o_nd, g_shape, g_nd, os);
    if (MODE == 4) bstrides(q_shape, q_nd, g_shape, g_nd, qs);
    int cshape[NK_MAX_DIMS];
    long long* vecs[4] = {ts, os, qs, gs};
    const int nd = collapse(g_shape, g_nd, vecs, 4, cshape);

    bool any_reduced = false;
    for (int i = 0; i < nd; ++i) if (ts[i] == 0 && cshape[i] != 1) any_reduced = true;

    if (!any_reduced) {
        Bcast3 b{};
        b.nd = nd;
        for (int i = 0; i < nd; ++i) { b.shape[i] = cshape[i]; b.os[i] = os[i]; b.qs[i] = qs[i]; }
        const bool vec = (cshape[nd - 1] % 4 == 0) && al16(d) && al16(g) &&
                         (MODE < 2 || (os[nd - 1] <= 1 && al16(o))) && (MODE != 4 || (qs[nd - 1] <= 1 && al16(q)));
        const int grid = nk_stream_grid((size_t)(total / (vec ? 4 : 1)), 256);
        const int amode = (assign ? 1 : 0) | (nk_streams_past_cache((size_t)total * (assign ? 8 : 12)) ? 2 : 0);
        if (vec) hipLaunchKernelGGL((binary_bwd_same_kernel<MODE, true>), dim3(grid), dim3(256), 0, dev->compute, d, g, o, q, b, total, amode);
        else hipLaunchKernelGGL((binary_bwd_same_kernel<MODE, false>), dim3(grid), dim3(256), 0, dev->compute, d, g, o, q, b, total, amode);
        NK_LAUNCH_CHECK();
        return NK_OK;
    }

    // classify into [R0][K][R1]
    int pat[NK_MAX_DIMS];
    for (int i = 0; i < nd; ++i) pat[i] = ts[i] == 0 ? 0 : 1;  // 0 = reduced, 1 = kept
    int groups = 1;
    for (int i = 1; i < nd; ++i) if (pat[i] != pat[i - 1]) ++groups;
    // count of kept groups
    int kept_groups = 0;
    for (int i = 0; i < nd; ++i) if (pat[i] == 1 && (i == 0 || pat[i - 1] == 0)) ++kept_groups;
    const bool rkr = (nd <= 3) && kept_groups <= 1 && groups <= 3;
    if (rkr) {
        Rkr p{};
        p.R0 = 1; p.K = 1; p.R1 = 1;
        int ki = -1;
        for (int i = 0; i < nd; ++i) if (pat[i] == 1) ki = i;
        if (ki < 0) {  // everything reduced (scalar target): view the flat range as [R0][1][R1]
            if (nd != 1) goto generic;  // o/q strides prevented merging into one flat range
            long long r1 = 8192;
            while (r1 > 1 && total % r1 != 0) r1 >>= 1;
            if (total / r1 > 0x7fffffffLL) goto generic;
            p.R1 = (int)r1; p.R0 = (int)(total / r1);
            p.os1 = os[0]; p.qs1 = qs[0]; p.os0 = os[0] * r1; p.qs0 = qs[0] * r1;
        } else {
            if (ki > 1 || nd - 1 - ki > 1) goto generic;
            p.K = cshape[ki]; p.osk = os[ki]; p.qsk = qs[ki];
            if (ki == 1) { p.R0 = cshape[0]; p.os0 = os[0]; p.qs0 = qs[0]; }
            if (ki + 1 < nd) { p.R1 = cshape[ki + 1]; p.os1 = os[ki + 1]; p.qs1 = qs[ki + 1]; }
        }
        // chunking of R0 so that the grid fills the chip; partials in the workspace
        int chunks;
        const bool cols4 = p.R1 == 1 && MODE <= 1 && p.K % 4 == 0 && al16(g);
        if (p.R1 == 1) {
            const int colblocks = (p.K + 255) / 256;
            chunks = (1024 + colblocks - 1) / colblocks;
            if (chunks > (p.R0 + 31) / 32) chunks = (p.R0 + 31) / 32;
        } else {
            chunks = (1024 + p.K - 1) / p.K;
            if (chunks > p.R0) chunks = p.R0;
        }
        if (chunks < 1) chunks = 1;
        p.rows_per_chunk = (p.R0 + chunks - 1) / chunks;
        chunks = (p.R0 + p.rows_per_chunk - 1) / p.rows_per_chunk;
        p.vec = (p.R1 % 4 == 0) && al16(g) && (MODE < 2 || (p.os1 <= 1 && al16(o))) &&
                (MODE != 4 || (p.qs1 <= 1 && al16(q)));
        void* ws = nullptr;
        rc = nk_workspace(dev, (size_t)chunks * p.K * sizeof(float), &ws);
        if (rc) return rc;
        float* part = (float*)ws;
        if (cols4)
            hipLaunchKernelGGL((reduce_cols4_kernel<MODE>), dim3((p.K + 255) / 256, chunks), dim3(256), 0, dev->compute, part, g, p);
        else if (p.R1 == 1)
            hipLaunchKernelGGL((reduce_cols_kernel<MODE>), dim3((p.K + 255) / 256, chunks), dim3(256), 0, dev->compute, part, g, o, q, p);
        else
            hipLaunchKernelGGL((reduce_rkr_kernel<MODE>), dim3(p.K, chunks), dim3(256), 0, dev->compute, part, g, o, q, p);
        NK_LAUNCH_CHECK();
        hipLaunchKernelGGL(reduce_finish_kernel, dim3((p.K + 63) / 64), dim3(256), 0, dev->compute, d, part, p.K, chunks, assign);
        NK_LAUNCH_CHECK();
        return NK_OK;
    }
generic: {
        Gen p{};
        p.nd = nd;
        long long kept = 1, reduced = 1;
        for (int i = 0; i < nd; ++i) {
            p.shape[i] = cshape[i]; p.ds[i] = ts[i]; p.os[i] = os[i]; p.qs[i] = qs[i]; p.gs[i] = gs[i];
            if (ts[i] == 0) reduced *= cshape[i]; else kept *= cshape[i];
        }
        hipLaunchKernelGGL((reduce_generic_kernel<MODE>), dim3(nk_stream_grid((size_t)kept, 64)), dim3(64), 0,
                           dev->compute, d, g, o, q, p, kept, reduced, assign);
        NK_LAUNCH_CHECK();
        return NK_OK;
    }
}

__global__ void fill_kernel(float* __restrict__ p, size_t n, float v) {
    const size_t n4 = n / 4;
    nk_span_walk<4>(n4, [](size_t) { return 0; }, [&](size_t i, int) { nk_store_stream(reinterpret_cast<float4*>(p) + i, make_float4(v, v, v, v)); });
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] = v;
}

template <bool VEC>
__global__ void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    if (VEC) {
        const size_t n4 = n / 4;
        nk_span_walk<4>(n4, [&](size_t i) { return reinterpret_cast<const float4*>(x)[i]; }, [&](size_t i, float4 v) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            nk_store_stream(reinterpret_cast<float4*>(y) + i, v);
        });
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = fmaxf(x[n4 * 4 + threadIdx.x], 0.f);
    } else {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            y[i] = fmaxf(x[i], 0.f);
    }
}

template <bool VEC>
__global__ void relu_bwd_kernel(float* __restrict__ dx, const float* __restrict__ g, const float* __restrict__ x, size_t n,
                                int assign) {  // assign: dx is a freshly zeroed gradient -> write without reading it
    if (VEC) {
        const size_t n4 = n / 4;
        // bit 1 of `assign`: `nt` loads (operands beyond the Infinity Cache, nk_common.h)
        struct R { float4 x, g, d; };
        nk_span_walk<4>(n4, [&](size_t i) {
            R r;
            r.x = nk_load_stream(reinterpret_cast<const float4*>(x) + i, assign & 2);
            r.g = nk_load_stream(reinterpret_cast<const float4*>(g) + i, assign & 2);
            r.d = (assign & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : nk_load_stream(reinterpret_cast<const float4*>(dx) + i, assign & 2);
            return r;
        }, [&](size_t i, const R& r) {
            float4 d = r.d;
            d.x += r.x.x > 0.f ? r.g.x : 0.f * r.g.x; d.y += r.x.y > 0.f ? r.g.y : 0.f * r.g.y;
            d.z += r.x.z > 0.f ? r.g.z : 0.f * r.g.z; d.w += r.x.w > 0.f ? r.g.w : 0.f * r.g.w;
            nk_store_stream(reinterpret_cast<float4*>(dx) + i, d);
        });
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
            const size_t i = n4 * 4 + threadIdx.x;
            dx[i] = ((assign & 1) ? 0.f : dx[i]) + (x[i] > 0.f ? g[i] : 0.f * g[i]);
        }
    } else {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            dx[i] = ((assign & 1) ? 0.f : dx[i]) + (x[i] > 0.f ? g[i] : 0.f * g[i]);
    }
}

// g = ((y > 0) as f32) * g in place (one pointer for source and destination: no __restrict__ pair to alias)
template <bool VEC>
__global__ void relu_mask_inplace_kernel(float* g, const float* __restrict__ y, size_t n, bool nt = false) {
    if (VEC) {
        const size_t n4 = n / 4;
        struct R { float4 y, g; };
        nk_span_walk<4>(n4, [&](size_t i) {
            R r;
            r.y = nk_load_stream(reinterpret_cast<const float4*>(y) + i, nt);
            r.g = nk_load_stream(reinterpret_cast<const float4*>(g) + i, nt);
            return r;
        }, [&](size_t i, const R& r) {
            float4 v = r.g;
            v.x = r.y.x > 0.f ? v.x : 0.f * v.x; v.y = r.y.y > 0.f ? v.y : 0.f * v.y;
            v.z = r.y.z > 0.f ? v.z : 0.f * v.z; v.w = r.y.w > 0.f ? v.w : 0.f * v.w;
            reinterpret_cast<float4*>(g)[i] = v;
        });
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
            const size_t i = n4 * 4 + threadIdx.x;
            g[i] = y[i] > 0.f ? g[i] : 0.f * g[i];
        }
    } else {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            g[i] = y[i] > 0.f ? g[i] : 0.f * g[i];
    }
}

// ---- pointwise unary nodes (node/{negation,exp,logn,sqrt,sigmoid,tanh,softplus,leaky_relu,power}) --
__device__ __forceinline__ float powi_dev(float b, int e) {  // Rust `f32::powi`
    float r = 1.f;
    int k = e < 0 ? -e : e;
    while (k) { if (k & 1) r *= b; b *= b; k >>= 1; }
    return e < 0 ? 1.f / r : r;
}
template <int OP>
__device__ __forceinline__ float unary_f(float x, int e) {
    switch (OP) {
        case NK_NEG: return -x;
        case NK_EXP: return expf(x);
        case NK_LN: return logf(x);
        case NK_SQRT: return sqrtf(x);
        case NK_SIGMOID: return 1.f / (1.f + expf(-x));
        case NK_TANH: return tanhf(x);
        case NK_SOFTPLUS: return logf(1.f + expf(x));
        case NK_LEAKY_RELU: return (x > 0.f ? 1.f : 0.f) * x + (x <= 0.f ? 1.f : 0.f) * (0.01f * x);
        default: return powi_dev(x, e);
    }
}
// local gradient term added to dx; r = the buffer the reference node keeps (input or output)
template <int OP>
__device__ __forceinline__ float unary_df(float g, float r, int e) {
    switch (OP) {
        case NK_NEG: return -g;
        case NK_EXP: return g * r;
        case NK_LN: return g / r;
        case NK_SQRT: return g / (r * 2.f);
        case NK_SIGMOID: return g * r * (1.f - r);
        case NK_TANH: return g * (1.f - r * r);
        case NK_SOFTPLUS: return g / (1.f + expf(-r));
        case NK_LEAKY_RELU: return (r > 0.f ? 1.f : 0.f) * g + (r <= 0.f ? 1.f : 0.f) * 0.01f;
        default: return g * powi_dev(r, e - 1) * (float)e;
    }
}
template <int OP>
__global__ void unary_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int e) {
    const size_t n4 = n / 4;
    nk_span_walk<4>(n4, [&](size_t i) { return reinterpret_cast<const float4*>(x)[i]; }, [&](size_t i, float4 v) {
        v.x = unary_f<OP>(v.x, e); v.y = unary_f<OP>(v.y, e); v.z = unary_f<OP>(v.z, e); v.w = unary_f<OP>(v.w, e);
        nk_store_stream(reinterpret_cast<float4*>(y) + i, v);
    });
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = unary_f<OP>(x[n4 * 4 + threadIdx.x], e);
}
template <int OP>
__global__ void unary_bwd_kernel(float* __restrict__ dx, const float* __restrict__ g, const float* __restrict__ r,
                                 size_t n, int e, int assign) {
    const size_t n4 = n / 4;
    struct R { float4 d, g, r; };
    nk_span_walk<4>(n4, [&](size_t i) {
        R q;
        q.d = assign ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(dx)[i];
        q.g = reinterpret_cast<const float4*>(g)[i];
        q.r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (OP != NK_NEG) q.r = reinterpret_cast<const float4*>(r)[i];
        return q;
    }, [&](size_t i, const R& q) {
        float4 d = q.d;
        d.x += unary_df<OP>(q.g.x, q.r.x, e); d.y += unary_df<OP>(q.g.y, q.r.y, e);
        d.z += unary_df<OP>(q.g.z, q.r.z, e); d.w += unary_df<OP>(q.g.w, q.r.w, e);
        nk_store_stream(reinterpret_cast<float4*>(dx) + i, d);
    });
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        dx[i] = (assign ? 0.f : dx[i]) + unary_df<OP>(g[i], OP != NK_NEG ? r[i] : 0.f, e);
    }
}

// ---- optimizers (neuronika-optim): one pass over the parameter, its gradient and the state ------
// grad += penalize(w)  (penalty.rs:63-79; Rust signum: +-0 -> +-1)
__device__ __forceinline__ float penalized(float w, float g, float l1, float l2) {
    if (l1 != 0.f) g += l1 * copysignf(1.f, w);
    if (l2 != 0.f) g += 2.f * l2 * w;
    return g;
}

// one element of `SGDParam::optimize` (sgd/mod.rs:186-236); shared by the one-parameter and the multi-parameter kernel so
// that both make the same contraction choices: their results are bit-identical (test_optimizer_steps)
__device__ __forceinline__ float sgd_one(float wi, float& gi, float* vel, float lr, float momentum, float dampening, int nesterov,
                                         float l1, float l2) {
    gi = penalized(wi, gi, l1, l2);
    if (vel == nullptr) return wi - gi * lr;
    const float v = *vel * momentum + gi * (1.f - dampening);
    *vel = v;
    return wi - (nesterov ? (gi + v * momentum) * lr : v * lr);
}
// `count` <= SGD_MULTI_MAX parameters in one launch: the blocks walk the concatenation of the parameters in chunks of
// SGD_CHUNK elements (a chunk never straddles two parameters); 16-byte accesses where a parameter's three pointers allow.
constexpr int SGD_MULTI_MAX = 8;
constexpr int SGD_CHUNK = 4096;
struct SgdMulti {
    float* w[SGD_MULTI_MAX];
    float* g[SGD_MULTI_MAX];
    float* v[SGD_MULTI_MAX];
    size_t n[SGD_MULTI_MAX];
    unsigned first_chunk[SGD_MULTI_MAX + 1];  // prefix sums of ceil(n / SGD_CHUNK)
    int count;
};
__global__ void sgd_multi_kernel(SgdMulti a, float lr, float momentum, float dampening, int nesterov, float l1, float l2) {
    const bool pen = l1 != 0.f || l2 != 0.f;
    const unsigned total = a.first_chunk[a.count];
    // span walk over the chunks (nk_span_walk's order: a block takes CONSECUTIVE chunks, nk_common.h); a whole chunk = four trips of
    // the 256 lanes, all of its loads issued before the first use
    const unsigned per = (total + gridDim.x - 1) / gridDim.x, c0 = blockIdx.x * per, c1 = c0 + per < total ? c0 + per : total;
    for (unsigned ch = c0; ch < c1; ++ch) {
        int t = 0;
#pragma unroll
        for (int k = 1; k < SGD_MULTI_MAX; ++k) t += (k < a.count && ch >= a.first_chunk[k]) ? 1 : 0;
        float* __restrict__ w = a.w[t];
        float* __restrict__ g = a.g[t];
        float* __restrict__ v = a.v[t];
        const size_t n = a.n[t], base = (size_t)(ch - a.first_chunk[t]) * SGD_CHUNK;
        const size_t end = base + SGD_CHUNK < n ? base + SGD_CHUNK : n;
        const bool vec = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
        if (vec && end - base == SGD_CHUNK) {
            constexpr int TRIPS = SGD_CHUNK / (4 * 256);  // the launch uses 256 threads
            float4 wv[TRIPS], gv[TRIPS], vv[TRIPS];
#pragma unroll
            for (int u = 0; u < TRIPS; ++u) {
                const size_t i = base + 4 * threadIdx.x + (size_t)u * 1024;
                wv[u] = *reinterpret_cast<const float4*>(w + i);
                gv[u] = *reinterpret_cast<const float4*>(g + i);
                vv[u] = v ? *reinterpret_cast<const float4*>(v + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < TRIPS; ++u) {
                const size_t i = base + 4 * threadIdx.x + (size_t)u * 1024;
                wv[u].x = sgd_one(wv[u].x, gv[u].x, v ? &vv[u].x : nullptr, lr, momentum, dampening, nesterov, l1, l2);
                wv[u].y = sgd_one(wv[u].y, gv[u].y, v ? &vv[u].y : nullptr, lr, momentum, dampening, nesterov, l1, l2);
                wv[u].z = sgd_one(wv[u].z, gv[u].z, v ? &vv[u].z : nullptr, lr, momentum, dampening, nesterov, l1, l2);
                wv[u].w = sgd_one(wv[u].w, gv[u].w, v ? &vv[u].w : nullptr, lr, momentum, dampening, nesterov, l1, l2);
                if (pen) *reinterpret_cast<float4*>(g + i) = gv[u];
                if (v) *reinterpret_cast<float4*>(v + i) = vv[u];
                *reinterpret_cast<float4*>(w + i) = wv[u];
            }
        } else {
            for (size_t i = base + threadIdx.x; i < end; i += blockDim.x) {
                float gi = g[i];
                const float wi = sgd_one(w[i], gi, v ? v + i : nullptr, lr, momentum, dampening, nesterov, l1, l2);
                if (pen) g[i] = gi;
                w[i] = wi;
            }
        }
    }
}

__global__ void adam_kernel(float* __restrict__ w, float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ vmax, size_t n, float lr, float beta1, float beta2, float eps,
                            float bc1, float bc2, float l1, float l2) {
    const bool pen = l1 != 0.f || l2 != 0.f;
    const float sbc2 = sqrtf(bc2), step_size = lr / bc1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float wi = w[i];
        const float gi = penalized(wi, grad[i], l1, l2);
        if (pen) grad[i] = gi;
        const float mi = m[i] * beta1 + gi * (1.f - beta1);
        const float vi = v[i] * beta2 + gi * gi * (1.f - beta2);
        m[i] = mi; v[i] = vi;
        float den = vi;
        if (vmax != nullptr) { den = fmaxf(vmax[i], vi); vmax[i] = den; }
        wi -= mi / ((sqrtf(den) / sbc2) + eps) * step_size;
        w[i] = wi;
    }
}

__global__ void adagrad_kernel(float* __restrict__ w, float* __restrict__ grad, float* __restrict__ gsq, size_t n,
                               float clr, float eps, float l1, float l2) {
    const bool pen = l1 != 0.f || l2 != 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float wi = w[i];
        const float gi = penalized(wi, grad[i], l1, l2);
        if (pen) grad[i] = gi;
        const float s = gsq[i] + gi * gi;
        gsq[i] = s;
        w[i] = wi - gi / (sqrtf(s) + eps) * clr;
    }
}

__global__ void rmsprop_kernel(float* __restrict__ w, float* __restrict__ grad, float* __restrict__ sq,
                               float* __restrict__ gavg, float* __restrict__ buf, size_t n, float lr, float alpha,
                               float eps, float momentum, float l1, float l2) {
    const bool pen = l1 != 0.f || l2 != 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float wi = w[i];
        const float gi = penalized(wi, grad[i], l1, l2);
        if (pen) grad[i] = gi;
        const float s = sq[i] * alpha + gi * gi * (1.f - alpha);
        sq[i] = s;
        float den;
        if (gavg != nullptr) {  // centered
            const float a = gavg[i] * alpha + gi * (1.f - alpha);
            gavg[i] = a;
            den = sqrtf(s + (-a * a)) + eps;
        } else {
            den = sqrtf(s) + eps;
        }
        if (buf != nullptr) {
            const float b = buf[i] * momentum + gi / den;
            buf[i] = b;
            wi -= b * lr;
        } else {
            wi -= gi / den * lr;
        }
        w[i] = wi;
    }
}

static float powi_f32(float b, int e) {  // Rust `f32::powi`: repeated squaring in f32
    float r = 1.f;
    int k = e < 0 ? -e : e;
    while (k) { if (k & 1) r *= b; b *= b; k >>= 1; }
    return e < 0 ? 1.f / r : r;
}

}  // namespace

extern "C" {

int nk_fill(nk_device* dev, float* ptr, size_t n, float value) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(ptr != nullptr, "null pointer in nk_fill");
    if (value == 0.f) {
        NK_HIP(hipMemsetAsync(ptr, 0, n * sizeof(float), dev->compute));
        return NK_OK;
    }
    NK_CHECK(al16(ptr), "nk_fill: pointer must be 16-byte aligned");
    hipLaunchKernelGGL(fill_kernel, dim3(nk_stream_grid(n / 4 + 1, 256)), dim3(256), 0, dev->compute, ptr, n, value);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

int nk_binary_fwd(nk_device* dev, int op, float* out, const int* out_shape, int out_nd, const float* l,
                  const int* l_shape, int l_nd, const float* r, const int* r_shape, int r_nd) {
    NK_USE(dev);
    NK_CHECK(op >= NK_ADD && op <= NK_DIV, "unknown binary op %d", op);
    NK_CHECK(out && l && r, "null operand in nk_binary_fwd");
    int rc = check_bcast(out_shape, out_nd, l_shape, l_nd, "left");
    if (rc) return rc;
    rc = check_bcast(out_shape, out_nd, r_shape, r_nd, "right");
    if (rc) return rc;
    for (int i = 0; i < out_nd; ++i) {  // out must be exactly cobroadcast(l, r)
        const int jl = i - (out_nd - l_nd), jr = i - (out_nd - r_nd);
        const int dl = jl >= 0 ? l_shape[jl] : 1, dr = jr >= 0 ? r_shape[jr] : 1;
        NK_CHECK(out_shape[i] == (dl > dr ? dl : dr), "output axis %d has extent %d, broadcast gives %d", i, out_shape[i], dl > dr ? dl : dr);
    }
    const long long total = (long long)nk_numel(out_shape, out_nd);
    if (total == 0) return NK_OK;
    Bcast b{};
    long long ls[NK_MAX_DIMS], rs[NK_MAX_DIMS];
    bstrides(l_shape, l_nd, out_shape, out_nd, ls);
    bstrides(r_shape, r_nd, out_shape, out_nd, rs);
    long long* vecs[2] = {ls, rs};
    b.nd = collapse(out_shape, out_nd, vecs, 2, b.shape);
    for (int i = 0; i < b.nd; ++i) { b.ls[i] = ls[i]; b.rs[i] = rs[i]; }
    const bool vec = (b.shape[b.nd - 1] % 4 == 0) && al16(out) && al16(l) && al16(r) && b.ls[b.nd - 1] <= 1 &&
                     b.rs[b.nd - 1] <= 1;
    const int grid = nk_stream_grid((size_t)(total / (vec ? 4 : 1)), 256);
#define NK_BIN_CASE(OP)                                                                                            \
    case OP:                                                                                                       \
        if (vec) hipLaunchKernelGGL((binary_fwd_kernel<OP, true>), dim3(grid), dim3(256), 0, dev->compute, out, l, r, b, total); \
        else hipLaunchKernelGGL((binary_fwd_kernel<OP, false>), dim3(grid), dim3(256), 0, dev->compute, out, l, r, b, total);    \
        break;
    switch (op) { NK_BIN_CASE(NK_ADD) NK_BIN_CASE(NK_SUB) NK_BIN_CASE(NK_MUL) NK_BIN_CASE(NK_DIV) }
#undef NK_BIN_CASE
    NK_LAUNCH_CHECK();
    return NK_OK;
}

static int binary_bwd_left(nk_device* dev, int op, float* d_left, const int* l_shape, int l_nd, const float* g,
                           const int* g_shape, int g_nd, const float* r, const int* r_shape, int r_nd, int assign) {
    switch (op) {
        case NK_ADD:
        case NK_SUB: return bwd_dispatch<0>(dev, d_left, l_shape, l_nd, g, g_shape, g_nd, nullptr, nullptr, 0, nullptr, nullptr, 0, assign);
        case NK_MUL: return bwd_dispatch<2>(dev, d_left, l_shape, l_nd, g, g_shape, g_nd, r, r_shape, r_nd, nullptr, nullptr, 0, assign);
        case NK_DIV: return bwd_dispatch<3>(dev, d_left, l_shape, l_nd, g, g_shape, g_nd, r, r_shape, r_nd, nullptr, nullptr, 0, assign);
    }
    nk_set_error("unknown binary op %d", op);
    return NK_ERR_INVALID;
}
static int binary_bwd_right(nk_device* dev, int op, float* d_right, const int* r_shape, int r_nd, const float* g,
                            const int* g_shape, int g_nd, const float* l, const int* l_shape, int l_nd,
                            const float* r, int assign) {
    switch (op) {
        case NK_ADD: return bwd_dispatch<0>(dev, d_right, r_shape, r_nd, g, g_shape, g_nd, nullptr, nullptr, 0, nullptr, nullptr, 0, assign);
        case NK_SUB: return bwd_dispatch<1>(dev, d_right, r_shape, r_nd, g, g_shape, g_nd, nullptr, nullptr, 0, nullptr, nullptr, 0, assign);
        case NK_MUL: return bwd_dispatch<2>(dev, d_right, r_shape, r_nd, g, g_shape, g_nd, l, l_shape, l_nd, nullptr, nullptr, 0, assign);
        case NK_DIV: return bwd_dispatch<4>(dev, d_right, r_shape, r_nd, g, g_shape, g_nd, l, l_shape, l_nd, r, r_shape, r_nd, assign);
    }
    nk_set_error("unknown binary op %d", op);
    return NK_ERR_INVALID;
}
int nk_binary_bwd_left(nk_device* dev, int op, float* d_left, const int* l_shape, int l_nd, const float* g,
                       const int* g_shape, int g_nd, const float* r, const int* r_shape, int r_nd) {
    return binary_bwd_left(dev, op, d_left, l_shape, l_nd, g, g_shape, g_nd, r, r_shape, r_nd, 0);
}
int nk_binary_bwd_left_assign(nk_device* dev, int op, float* d_left, const int* l_shape, int l_nd, const float* g,
                              const int* g_shape, int g_nd, const float* r, const int* r_shape, int r_nd) {
    return binary_bwd_left(dev, op, d_left, l_shape, l_nd, g, g_shape, g_nd, r, r_shape, r_nd, 1);
}
int nk_binary_bwd_right(nk_device* dev, int op, float* d_right, const int* r_shape, int r_nd, const float* g,
                        const int* g_shape, int g_nd, const float* l, const int* l_shape, int l_nd,
                        const float* r) {
    return binary_bwd_right(dev, op, d_right, r_shape, r_nd, g, g_shape, g_nd, l, l_shape, l_nd, r, 0);
}
int nk_binary_bwd_right_assign(nk_device* dev, int op, float* d_right, const int* r_shape, int r_nd, const float* g,
                               const int* g_shape, int g_nd, const float* l, const int* l_shape, int l_nd,
                               const float* r) {
    return binary_bwd_right(dev, op, d_right, r_shape, r_nd, g, g_shape, g_nd, l, l_shape, l_nd, r, 1);
}

int nk_unbroadcast_add(nk_device* dev, float* dst, const int* dst_shape, int dst_nd, const float* src,
                       const int* src_shape, int src_nd) {
    return bwd_dispatch<0>(dev, dst, dst_shape, dst_nd, src, src_shape, src_nd, nullptr, nullptr, 0, nullptr, nullptr, 0);
}
int nk_unbroadcast_assign(nk_device* dev, float* dst, const int* dst_shape, int dst_nd, const float* src,
                          const int* src_shape, int src_nd) {
    return bwd_dispatch<0>(dev, dst, dst_shape, dst_nd, src, src_shape, src_nd, nullptr, nullptr, 0, nullptr, nullptr, 0, 1);
}

int nk_unary_fwd(nk_device* dev, int op, const float* x, float* y, size_t n, int iparam) {
    NK_USE(dev);
    NK_CHECK(op >= NK_NEG && op <= NK_POW, "unknown unary op %d", op);
    if (n == 0) return NK_OK;
    NK_CHECK(x && y && al16(x) && al16(y), "nk_unary_fwd: null or unaligned pointer");
    const dim3 grid(nk_stream_grid(n / 4 + 1, 256)), block(256);
#define NK_U(OP) case OP: hipLaunchKernelGGL((unary_fwd_kernel<OP>), grid, block, 0, dev->compute, x, y, n, iparam); break;
    switch (op) { NK_U(NK_NEG) NK_U(NK_EXP) NK_U(NK_LN) NK_U(NK_SQRT) NK_U(NK_SIGMOID) NK_U(NK_TANH) NK_U(NK_SOFTPLUS) NK_U(NK_LEAKY_RELU) NK_U(NK_POW) }
#undef NK_U
    NK_LAUNCH_CHECK();
    return NK_OK;
}

static int unary_bwd(nk_device* dev, int op, float* dx, const float* g, const float* ref, size_t n, int iparam, int assign) {
    NK_USE(dev);
    NK_CHECK(op >= NK_NEG && op <= NK_POW, "unknown unary op %d", op);
    if (n == 0) return NK_OK;
    NK_CHECK(dx && g && al16(dx) && al16(g), "nk_unary_bwd: null or unaligned pointer");
    NK_CHECK(op == NK_NEG || (ref && al16(ref)), "nk_unary_bwd: the node's kept buffer is required");
    const dim3 grid(nk_stream_grid(n / 4 + 1, 256)), block(256);
#define NK_U(OP) case OP: hipLaunchKernelGGL((unary_bwd_kernel<OP>), grid, block, 0, dev->compute, dx, g, ref, n, iparam, assign); break;
    switch (op) { NK_U(NK_NEG) NK_U(NK_EXP) NK_U(NK_LN) NK_U(NK_SQRT) NK_U(NK_SIGMOID) NK_U(NK_TANH) NK_U(NK_SOFTPLUS) NK_U(NK_LEAKY_RELU) NK_U(NK_POW) }
#undef NK_U
    NK_LAUNCH_CHECK();
    return NK_OK;
}

int nk_unary_bwd(nk_device* dev, int op, float* dx, const float* g, const float* ref, size_t n, int iparam) {
    return unary_bwd(dev, op, dx, g, ref, n, iparam, 0);
}
int nk_unary_bwd_assign(nk_device* dev, int op, float* dx, const float* g, const float* ref, size_t n, int iparam) {
    return unary_bwd(dev, op, dx, g, ref, n, iparam, 1);
}

int nk_relu_fwd(nk_device* dev, const float* x, float* y, size_t n) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(x && y, "null pointer in nk_relu_fwd");
    const bool vec = al16(x) && al16(y);
    if (vec) hipLaunchKernelGGL((relu_fwd_kernel<true>), dim3(nk_stream_grid(n / 4 + 1, 256)), dim3(256), 0, dev->compute, x, y, n);
    else hipLaunchKernelGGL((relu_fwd_kernel<false>), dim3(nk_stream_grid(n, 256)), dim3(256), 0, dev->compute, x, y, n);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

static int relu_bwd(nk_device* dev, float* dx, const float* g, const float* x, size_t n, int assign) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(dx && g && x, "null pointer in nk_relu_bwd");
    const bool vec = al16(x) && al16(g) && al16(dx);
    assign = (assign ? 1 : 0) | (nk_streams_past_cache(n * (assign ? 12 : 16)) ? 2 : 0);
    if (vec) hipLaunchKernelGGL((relu_bwd_kernel<true>), dim3(nk_stream_grid(n / 4 + 1, 256)), dim3(256), 0, dev->compute, dx, g, x, n, assign);
    else hipLaunchKernelGGL((relu_bwd_kernel<false>), dim3(nk_stream_grid(n, 256)), dim3(256), 0, dev->compute, dx, g, x, n, assign);
    NK_LAUNCH_CHECK();
    return NK_OK;
}
int nk_relu_bwd(nk_device* dev, float* dx, const float* g, const float* x, size_t n) { return relu_bwd(dev, dx, g, x, n, 0); }
int nk_relu_bwd_assign(nk_device* dev, float* dx, const float* g, const float* x, size_t n) { return relu_bwd(dev, dx, g, x, n, 1); }

int nk_sgd_step_multi(nk_device* dev, int count, float* const* w, float* const* grad, float* const* velocity, const size_t* n,
                      float lr, float momentum, float dampening, int nesterov, float l1, float l2);
int nk_sgd_step(nk_device* dev, float* w, float* grad, float* velocity, size_t n, float lr, float momentum,
                float dampening, int nesterov, float l1, float l2) {
    // one parameter through the multi-parameter kernel: 16-byte accesses (6.4 against 5.3 TB/s of the scalar walk on a 256 MB
    // parameter), the same `sgd_one` per element
    NK_CHECK(n == 0 || (w && grad), "null pointer in nk_sgd_step");
    return nk_sgd_step_multi(dev, 1, &w, &grad, velocity ? &velocity : nullptr, &n, lr, momentum, dampening, nesterov, l1, l2);
}

int nk_sgd_step_multi(nk_device* dev, int count, float* const* w, float* const* grad, float* const* velocity, const size_t* n,
                      float lr, float momentum, float dampening, int nesterov, float l1, float l2) {
    NK_USE(dev);
    NK_CHECK(count >= 0 && (count == 0 || (w && grad && n)), "null table in nk_sgd_step_multi");
    for (int i = 0; i < count;) {
        SgdMulti a{};
        a.first_chunk[0] = 0;
        int c = 0;
        for (; i < count && c < SGD_MULTI_MAX; ++i) {  // the next (up to) SGD_MULTI_MAX non-empty parameters
            if (n[i] == 0) continue;
            NK_CHECK(w[i] && grad[i], "null pointer in nk_sgd_step_multi");
            NK_CHECK((n[i] + SGD_CHUNK - 1) / SGD_CHUNK < (1u << 30), "parameter too large for nk_sgd_step_multi");
            a.w[c] = w[i]; a.g[c] = grad[i]; a.v[c] = velocity ? velocity[i] : nullptr; a.n[c] = n[i];
            a.first_chunk[c + 1] = a.first_chunk[c] + (unsigned)((n[i] + SGD_CHUNK - 1) / SGD_CHUNK);
            ++c;
        }
        a.count = c;
        if (c == 0) break;
        for (int k = c + 1; k <= SGD_MULTI_MAX; ++k) a.first_chunk[k] = a.first_chunk[c];
        const unsigned total = a.first_chunk[c];
        const unsigned grid = total < 8192u ? total : 8192u;
        hipLaunchKernelGGL(sgd_multi_kernel, dim3(grid), dim3(256), 0, dev->compute, a, lr, momentum, dampening, nesterov, l1, l2);
        NK_LAUNCH_CHECK();
    }
    return NK_OK;
}

int nk_relu_mask_inplace(nk_device* dev, float* g, const float* y, size_t n) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(g && y, "null pointer in nk_relu_mask_inplace");
    const bool vec = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (vec) hipLaunchKernelGGL((relu_mask_inplace_kernel<true>), dim3(nk_stream_grid(n / 4 + 1, 256)), dim3(256), 0, dev->compute, g, y, n,
                                nk_streams_past_cache(n * 12));
    else hipLaunchKernelGGL((relu_mask_inplace_kernel<false>), dim3(nk_stream_grid(n, 256)), dim3(256), 0, dev->compute, g, y, n);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

static int refuse_capture(nk_device* dev, const char* what) {   // nk_common.h: why
    return nk_refuse_capture(dev, what, "issue this optimizer step outside the captured region");
}

int nk_adam_step(nk_device* dev, float* w, float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                 size_t n, float lr, float beta1, float beta2, float eps, int step, float l1, float l2) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(w && grad && exp_avg && exp_avg_sq, "null pointer in nk_adam_step");
    NK_CHECK(step >= 1, "step must be >= 1");
    if (int rc = refuse_capture(dev, "nk_adam_step: the bias corrections 1 - beta^step")) return rc;
    const float bc1 = 1.f - powi_f32(beta1, step), bc2 = 1.f - powi_f32(beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(nk_stream_grid(n, 256)), dim3(256), 0, dev->compute, w, grad, exp_avg, exp_avg_sq,
                       max_exp_avg_sq, n, lr, beta1, beta2, eps, bc1, bc2, l1, l2);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

int nk_adagrad_step(nk_device* dev, float* w, float* grad, float* grad_sq, size_t n, float lr, float lr_decay,
                    float eps, int step, float l1, float l2) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(w && grad && grad_sq, "null pointer in nk_adagrad_step");
    NK_CHECK(step >= 1, "step must be >= 1");
    if (lr_decay != 0.f)
        if (int rc = refuse_capture(dev, "nk_adagrad_step: the decayed learning rate lr / (1 + (step-1) * lr_decay)")) return rc;
    const float clr = lr / (1.f + (float)(step - 1) * lr_decay);
    hipLaunchKernelGGL(adagrad_kernel, dim3(nk_stream_grid(n, 256)), dim3(256), 0, dev->compute, w, grad, grad_sq, n, clr,
                       eps, l1, l2);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

int nk_rmsprop_step(nk_device* dev, float* w, float* grad, float* square_avg, float* grad_avg, float* buffer, size_t n,
                    float lr, float alpha, float eps, float momentum, float l1, float l2) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(w && grad && square_avg, "null pointer in nk_rmsprop_step");
    hipLaunchKernelGGL(rmsprop_kernel, dim3(nk_stream_grid(n, 256)), dim3(256), 0, dev->compute, w, grad, square_avg,
                       grad_avg, buffer, n, lr, alpha, eps, momentum, l1, l2);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

}  // extern "C"
