// N-d (1/2/3-d) strided, dilated, grouped cross-correlation as IMPLICIT GEMM on the f32 MFMA
// core (nk_mma.h).  Replaces node/convolution/mod.rs:
//   convolution                 :85-123   Y[n]  = Wflat . cols[n]^T                (beta 0)
//   convolution_backward_input  :146-189  dX   += col2im(Wflat^T . G[n])           (gather form)
//   convolution_backward_kernel :191-226  dW[c]+= G[:,c,:] . cols
//   grouped wrappers            :125-144, 256-294 (channel chunks, run in grid.z here)
// The reference materialises the im2col matrix (N x L x K floats, 925 MB at the C3 config,
// twice) and a K x L buffer per sample for the backward-input; here the columns are gathered
// on the fly while staging tiles into LDS, and the backward-input is written as a gather over
// (co, kernel offset) — deterministic, no atomics, no col2im scatter.
//
//   forward     : M = Cout/g   cols = (n, out pos)   k = (ci, kernel idx)
//   bwd-input   : M = Cin/g    cols = (n, in pos)    k = (co, kernel idx)   [W pre-transposed]
//   bwd-kernel  : M = Cout/g   cols = (ci, kernel idx)   k = (n, out pos)   [split over k]
#include "nk_mma.h"

using namespace nkmma;

namespace {

struct ConvGeom {
    int N, Cin, Cout, groups, Cg, Mg;  // Cg = Cin/groups, Mg = Cout/groups
    int in[3], out[3], k[3], stride[3], dil[3];  // padded in front with 1s to 3 spatial dims
    int inplane, L, KK;                           // prod(in), prod(out), prod(k)
};

// ---- tables (tiny pre-kernels into the device workspace) ---------------------------------------
// koff[k], k = ci*KK + kidx : input offset of kernel element k relative to the window origin
__global__ void conv_koff_kernel(int* __restrict__ koff, ConvGeom g) {
    const int K = g.Cg * g.KK;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int ci = k / g.KK;
        int rem = k % g.KK;
        const int k2 = rem % g.k[2]; rem /= g.k[2];
        const int k1 = rem % g.k[1];
        const int k0 = rem / g.k[1];
        koff[k] = ci * g.inplane + (k0 * g.dil[0] * g.in[1] + k1 * g.dil[1]) * g.in[2] + k2 * g.dil[2];
    }
}
// ktab[k'], k' = co*KK + kidx : {co*L, k0*dil0, k1*dil1, k2*dil2}
__global__ void conv_ktab_kernel(int4* __restrict__ ktab, ConvGeom g) {
    const int K = g.Mg * g.KK;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int co = k / g.KK;
        int rem = k % g.KK;
        const int k2 = rem % g.k[2]; rem /= g.k[2];
        const int k1 = rem % g.k[1];
        const int k0 = rem / g.k[1];
        ktab[k] = make_int4(co * g.L, k0 * g.dil[0], k1 * g.dil[1], k2 * g.dil[2]);
    }
}
// Wt[grp][ci][co][kidx] = W[grp*Mg + co][ci][kidx]
__global__ void conv_wt_kernel(float* __restrict__ wt, const float* __restrict__ w, ConvGeom g) {
    const long long total = (long long)g.Cout * g.Cg * g.KK;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int kidx = (int)(i % g.KK);
        long long rem = i / g.KK;
        const int co = (int)(rem % g.Mg); rem /= g.Mg;
        const int ci = (int)(rem % g.Cg);
        const int grp = (int)(rem / g.Cg);
        wt[i] = w[((long long)(grp * g.Mg + co) * g.Cg + ci) * g.KK + kidx];
    }
}

// ---- column helpers -----------------------------------------------------------------------------
// flat output position l -> offset of its window origin inside one input plane
__device__ __forceinline__ int window_origin(const ConvGeom& g, int l) {
    const int o2 = l % g.out[2];
    int rem = l / g.out[2];
    const int o1 = rem % g.out[1];
    const int o0 = rem / g.out[1];
    return (o0 * g.stride[0] * g.in[1] + o1 * g.stride[1]) * g.in[2] + o2 * g.stride[2];
}

// =================================================================================================
// forward
// =================================================================================================
struct FwdArgs {
    ConvGeom g;
    const float* x;
    const float* w;
    float* y;
    const int* koff;
    int tiles_m, tiles_n;
};

template <bool ALIGNED_A>
__global__ __launch_bounds__(NT, 2) void conv_fwd_kernel(FwdArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const ConvGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    const int grp = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = g.Cg * g.KK;
    const long long cols = (long long)g.N * g.L;
    const float* W = p.w + (long long)grp * g.Mg * K;
    const float* X = p.x + (long long)grp * g.Cg * g.inplane;
    const int nt = (K + BK - 1) / BK;

    // this thread gathers columns n0 + 4*cq + {0..3} for k rows (t>>5) + 8*j of every k-tile
    const int cq = t & 31, krow = t >> 5;
    long long b0, b1, b2, b3;
    bool v0, v1, v2, v3;
    {
        const long long c = (long long)n0 + cq * 4;
#define NK_COL(i, B, V)                                                               \
    {                                                                                 \
        const long long cc = c + i;                                                   \
        V = cc < cols;                                                                \
        const int n = V ? (int)(cc / g.L) : 0, l = V ? (int)(cc % g.L) : 0;           \
        B = (long long)n * g.Cin * g.inplane + window_origin(g, l);                   \
    }
        NK_COL(0, b0, v0) NK_COL(1, b1, v1) NK_COL(2, b2, v2) NK_COL(3, b3, v3)
#undef NK_COL
    }
    auto gather = [&](int k0) {
        Stage r;
#define NK_ROW(j, V)                                                                  \
    {                                                                                 \
        const int k = k0 + krow + 8 * j;                                              \
        const bool kv = k < K;                                                        \
        const int off = kv ? p.koff[k] : 0;                                           \
        V = make_float4(kv && v0 ? X[b0 + off] : 0.f, kv && v1 ? X[b1 + off] : 0.f,   \
                        kv && v2 ? X[b2 + off] : 0.f, kv && v3 ? X[b3 + off] : 0.f);  \
    }
        NK_ROW(0, r.v0) NK_ROW(1, r.v1) NK_ROW(2, r.v2) NK_ROW(3, r.v3)
#undef NK_ROW
        return r;
    };

    f32x16 acc[2][2];
    acc_zero(acc);
    TileLoader<true> la;
    la.init(W, K, m0, 0, g.Mg, K, t);
    Stage ra, rb;
    ra = la.template load<ALIGNED_A>(t);
    rb = gather(0);
    stage_store<true>(smem, ra, t);
    stage_store<false>(smem + TILE_FLOATS, rb, t);
    __syncthreads();
    for (int it = 0; it + 1 < nt; ++it) {
        float* cur = smem + (it & 1) * STAGE_FLOATS;
        float* nxt = smem + ((it + 1) & 1) * STAGE_FLOATS;
        ra = la.template load<ALIGNED_A>(t);
        rb = gather((it + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<true, false>(cur, cur + TILE_FLOATS, acc, wr, wc, lane);
        stage_store<true>(nxt, ra, t);
        stage_store<false>(nxt + TILE_FLOATS, rb, t);
        __syncthreads();
    }
    {
        float* cur = smem + ((nt - 1) & 1) * STAGE_FLOATS;
        mma_tile<true, false>(cur, cur + TILE_FLOATS, acc, wr, wc, lane);
    }
    // Y[n][grp*Mg + co][l]
    float* Y = p.y;
    const int Mg = g.Mg, L = g.L, Cout = g.Cout;
    acc_foreach(acc, wr, wc, lane, [&](int r, int c, float v) {
        const int co = m0 + r;
        const long long cc = (long long)n0 + c;
        if (co < Mg && cc < cols) {
            const int n = (int)(cc / L), l = (int)(cc % L);
            Y[((long long)n * Cout + grp * Mg + co) * L + l] = v;
        }
    });
}

// =================================================================================================
// backward w.r.t. the input (gather form)
// =================================================================================================
struct BwdInArgs {
    ConvGeom g;
    float* dx;
    const float* gy;
    const float* wt;     // [groups][Cg][Mg*KK]
    const int4* ktab;
    int tiles_m, tiles_n;
};

template <bool ALIGNED_A, bool UNIT_STRIDE>
__global__ __launch_bounds__(NT, 2) void conv_bwd_input_kernel(BwdInArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const ConvGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    const int grp = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = g.Mg * g.KK;
    const long long cols = (long long)g.N * g.inplane;
    const float* Wt = p.wt + (long long)grp * g.Cg * K;
    const float* G = p.gy + (long long)grp * g.Mg * g.L;
    const int nt = (K + BK - 1) / BK;

    const int cq = t & 31, krow = t >> 5;
    // per column: sample base into G and the input coordinates (p0,p1,p2)
    long long gb0, gb1, gb2, gb3;
    int pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, pc0, pc1, pc2, pc3;
    bool v0, v1, v2, v3;
    {
        const long long c = (long long)n0 + cq * 4;
#define NK_COL(i, GB, PA, PB, PC, V)                                                    \
    {                                                                                   \
        const long long cc = c + i;                                                     \
        V = cc < cols;                                                                  \
        const int n = V ? (int)(cc / g.inplane) : 0;                                    \
        int q = V ? (int)(cc % g.inplane) : 0;                                          \
        PC = q % g.in[2]; q /= g.in[2];                                                 \
        PB = q % g.in[1];                                                               \
        PA = q / g.in[1];                                                               \
        GB = (long long)n * g.Cout * g.L;                                               \
    }
        NK_COL(0, gb0, pa0, pb0, pc0, v0) NK_COL(1, gb1, pa1, pb1, pc1, v1)
        NK_COL(2, gb2, pa2, pb2, pc2, v2) NK_COL(3, gb3, pa3, pb3, pc3, v3)
#undef NK_COL
    }
    auto one = [&](const int4 kt, bool kv, long long gb, int pa, int pb, int pc, bool v) -> float {
        int a = pa - kt.y, b = pb - kt.z, c = pc - kt.w;
        bool ok = kv && v && a >= 0 && b >= 0 && c >= 0;
        if (!UNIT_STRIDE) {
            ok = ok && (a % g.stride[0] == 0) && (b % g.stride[1] == 0) && (c % g.stride[2] == 0);
            a /= g.stride[0]; b /= g.stride[1]; c /= g.stride[2];
        }
        ok = ok && a < g.out[0] && b < g.out[1] && c < g.out[2];
        return ok ? G[gb + kt.x + (a * g.out[1] + b) * g.out[2] + c] : 0.f;
    };
    auto gather = [&](int k0) {
        Stage r;
#define NK_ROW(j, V)                                                                    \
    {                                                                                   \
        const int k = k0 + krow + 8 * j;                                                \
        const bool kv = k < K;                                                          \
        const int4 kt = kv ? p.ktab[k] : make_int4(0, 0, 0, 0);                         \
        V = make_float4(one(kt, kv, gb0, pa0, pb0, pc0, v0), one(kt, kv, gb1, pa1, pb1, pc1, v1), \
                        one(kt, kv, gb2, pa2, pb2, pc2, v2), one(kt, kv, gb3, pa3, pb3, pc3, v3)); \
    }
        NK_ROW(0, r.v0) NK_ROW(1, r.v1) NK_ROW(2, r.v2) NK_ROW(3, r.v3)
#undef NK_ROW
        return r;
    };

    f32x16 acc[2][2];
    acc_zero(acc);
    TileLoader<true> la;
    la.init(Wt, K, m0, 0, g.Cg, K, t);
    Stage ra, rb;
    ra = la.template load<ALIGNED_A>(t);
    rb = gather(0);
    stage_store<true>(smem, ra, t);
    stage_store<false>(smem + TILE_FLOATS, rb, t);
    __syncthreads();
    for (int it = 0; it + 1 < nt; ++it) {
        float* cur = smem + (it & 1) * STAGE_FLOATS;
        float* nxt = smem + ((it + 1) & 1) * STAGE_FLOATS;
        ra = la.template load<ALIGNED_A>(t);
        rb = gather((it + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<true, false>(cur, cur + TILE_FLOATS, acc, wr, wc, lane);
        stage_store<true>(nxt, ra, t);
        stage_store<false>(nxt + TILE_FLOATS, rb, t);
        __syncthreads();
    }
    {
        float* cur = smem + ((nt - 1) & 1) * STAGE_FLOATS;
        mma_tile<true, false>(cur, cur + TILE_FLOATS, acc, wr, wc, lane);
    }
    // dX[n][grp*Cg + ci][pos] += acc
    float* DX = p.dx;
    const int Cg = g.Cg, Cin = g.Cin, inplane = g.inplane;
    acc_foreach(acc, wr, wc, lane, [&](int r, int c, float v) {
        const int ci = m0 + r;
        const long long cc = (long long)n0 + c;
        if (ci < Cg && cc < cols) {
            const int n = (int)(cc / inplane), q = (int)(cc % inplane);
            float* d = &DX[((long long)n * Cin + grp * Cg + ci) * inplane + q];
            *d += v;
        }
    });
}

// =================================================================================================
// backward w.r.t. the kernel (reduction over (n, out pos), split across blockIdx.y)
// =================================================================================================
struct BwdKArgs {
    ConvGeom g;
    const float* gy;
    const float* x;
    const int* koff;
    float* slabs;        // [splits][groups][Mg][Cg*KK]
    int tiles_m, tiles_n;
    long long r_per_split;  // multiple of BK
};

template <bool VEC_G>
__global__ __launch_bounds__(NT, 2) void conv_bwd_kernel_kernel(BwdKArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const ConvGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    const int grp = blockIdx.z, split = blockIdx.y;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Kc = g.Cg * g.KK;  // columns of dW
    const long long R = (long long)g.N * g.L;
    const long long rbeg = split * p.r_per_split;
    const long long rend = rbeg + p.r_per_split < R ? rbeg + p.r_per_split : R;
    const int nt = rend > rbeg ? (int)((rend - rbeg + BK - 1) / BK) : 0;
    const float* G = p.gy + (long long)grp * g.Mg * g.L;
    const float* X = p.x + (long long)grp * g.Cg * g.inplane;

    // KC staging for both operands: idx = t + 256*j -> row = (t>>3) + 32*j, 4 consecutive r
    const int rq = t & 7, row = t >> 3;
    // B: columns n0 + row + 32*j -> koff (fixed over the k loop)
    int ko0, ko1, ko2, ko3;
    bool cv0, cv1, cv2, cv3;
#define NK_KO(j, KO, CV) { const int c = n0 + row + 32 * j; CV = c < Kc; KO = CV ? p.koff[c] : 0; }
    NK_KO(0, ko0, cv0) NK_KO(1, ko1, cv1) NK_KO(2, ko2, cv2) NK_KO(3, ko3, cv3)
#undef NK_KO
    // A: rows (co) m0 + row + 32*j
    const bool av0 = m0 + row < g.Mg, av1 = m0 + row + 32 < g.Mg, av2 = m0 + row + 64 < g.Mg, av3 = m0 + row + 96 < g.Mg;

    Stage ra, rb;
    auto load_both = [&](long long r0) {
        // decompose the 4 consecutive reduction indices r0 + 4*rq + {0..3} -> (n, l)
        long long xo[4], go[4];
        bool rv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long r = r0 + rq * 4 + c;
            rv[c] = r < rend;
            const int n = rv[c] ? (int)(r / g.L) : 0, l = rv[c] ? (int)(r % g.L) : 0;
            xo[c] = (long long)n * g.Cin * g.inplane + window_origin(g, l);
            go[c] = (long long)n * g.Cout * g.L + l;
        }
#define NK_A(j, V, AV)                                                                       \
    {                                                                                        \
        const long long rowoff = (long long)(m0 + row + 32 * j) * g.L;                       \
        if (VEC_G && AV && rv[3]) V = *reinterpret_cast<const float4*>(&G[go[0] + rowoff]);  \
        else V = make_float4(AV && rv[0] ? G[go[0] + rowoff] : 0.f, AV && rv[1] ? G[go[1] + rowoff] : 0.f, \
                             AV && rv[2] ? G[go[2] + rowoff] : 0.f, AV && rv[3] ? G[go[3] + rowoff] : 0.f); \
    }
        NK_A(0, ra.v0, av0) NK_A(1, ra.v1, av1) NK_A(2, ra.v2, av2) NK_A(3, ra.v3, av3)
#undef NK_A
#define NK_B(V, KO, CV)                                                                      \
    V = make_float4(CV && rv[0] ? X[xo[0] + KO] : 0.f, CV && rv[1] ? X[xo[1] + KO] : 0.f,    \
                    CV && rv[2] ? X[xo[2] + KO] : 0.f, CV && rv[3] ? X[xo[3] + KO] : 0.f);
        NK_B(rb.v0, ko0, cv0) NK_B(rb.v1, ko1, cv1) NK_B(rb.v2, ko2, cv2) NK_B(rb.v3, ko3, cv3)
#undef NK_B
    };

    f32x16 acc[2][2];
    acc_zero(acc);
    if (nt > 0) {
        load_both(rbeg);
        stage_store<true>(smem, ra, t);
        stage_store<true>(smem + TILE_FLOATS, rb, t);
    }
    __syncthreads();
    for (int it = 0; it + 1 < nt; ++it) {
        float* cur = smem + (it & 1) * STAGE_FLOATS;
        float* nxt = smem + ((it + 1) & 1) * STAGE_FLOATS;
        load_both(rbeg + (long long)(it + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<true, true>(cur, cur + TILE_FLOATS, acc, wr, wc, lane);
        stage_store<true>(nxt, ra, t);
        stage_store<true>(nxt + TILE_FLOATS, rb, t);
        __syncthreads();
    }
    if (nt > 0) {
        float* cur = smem + ((nt - 1) & 1) * STAGE_FLOATS;
        mma_tile<true, true>(cur, cur + TILE_FLOATS, acc, wr, wc, lane);
    }
    float* S = p.slabs + ((long long)split * g.groups + grp) * (long long)g.Mg * Kc;
    const int Mg = g.Mg;
    acc_foreach(acc, wr, wc, lane, [&](int r, int c, float v) {
        const int co = m0 + r, col = n0 + c;
        if (co < Mg && col < Kc) S[(long long)co * Kc + col] = v;
    });
}

// dW[i] += sum_s slabs[s][i]  (fixed order -> deterministic)
__global__ void conv_dw_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slabs, long long n, int splits) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += slabs[(long long)k * n + i];
        dw[i] += s;
    }
}

// ---- host side ------------------------------------------------------------------------------------
int make_geom(int nd, const int* x_shape, const int* w_shape, const int* stride, const int* dilation, int groups,
              ConvGeom* out) {
    NK_CHECK(nd >= 1 && nd <= 3, "Invalid convolution dimension %d (1, 2 or 3 supported)", nd);
    NK_CHECK(groups >= 1, "groups must be >= 1");
    ConvGeom g{};
    g.N = x_shape[0]; g.Cin = x_shape[1]; g.Cout = w_shape[0]; g.groups = groups;
    NK_CHECK(g.Cin % groups == 0, "In channels %d is not divisible by groups %d", g.Cin, groups);
    NK_CHECK(g.Cout % groups == 0, "Out channels %d is not divisible by groups %d", g.Cout, groups);
    g.Cg = g.Cin / groups; g.Mg = g.Cout / groups;
    NK_CHECK(w_shape[1] == g.Cg, "kernel has %d input channels per group, expected %d", w_shape[1], g.Cg);
    for (int d = 0; d < 3; ++d) { g.in[d] = g.out[d] = g.k[d] = g.stride[d] = g.dil[d] = 1; }
    g.inplane = g.L = g.KK = 1;
    for (int d = 0; d < nd; ++d) {
        const int q = 3 - nd + d;
        g.in[q] = x_shape[2 + d]; g.k[q] = w_shape[2 + d]; g.stride[q] = stride[d]; g.dil[q] = dilation[d];
        NK_CHECK(g.stride[q] >= 1 && g.dil[q] >= 1 && g.k[q] >= 1, "bad stride/dilation/kernel on axis %d", d);
        NK_CHECK(g.in[q] >= (g.k[q] - 1) * g.dil[q] + 1, "The kernel size can't be greater than actual input size.");
        g.out[q] = (g.in[q] - g.dil[q] * (g.k[q] - 1) - 1) / g.stride[q] + 1;
        g.inplane *= g.in[q]; g.L *= g.out[q]; g.KK *= g.k[q];
    }
    NK_CHECK((long long)g.Cin * g.inplane < 0x7fffffffLL && (long long)g.Cout * g.L < 0x7fffffffLL,
             "one sample exceeds 2^31 elements");
    *out = g;
    return NK_OK;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
size_t round256(size_t b) { return (b + 255) & ~size_t(255); }

}  // namespace

extern "C" {

int nk_conv_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, const float* w, const int* w_shape,
                float* y, const int* stride, const int* dilation, int groups) {
    NK_USE(dev);
    ConvGeom g;
    int rc = make_geom(nd, x_shape, w_shape, stride, dilation, groups, &g);
    if (rc) return rc;
    if ((long long)g.N * g.Cout * g.L == 0) return NK_OK;
    NK_CHECK(x && w && y, "null pointer in nk_conv_fwd");
    const int K = g.Cg * g.KK;
    void* ws = nullptr;
    rc = nk_workspace(dev, round256((size_t)K * sizeof(int)), &ws);
    if (rc) return rc;
    hipLaunchKernelGGL(conv_koff_kernel, dim3((K + 255) / 256), dim3(256), 0, dev->compute, (int*)ws, g);
    NK_LAUNCH_CHECK();
    FwdArgs p{};
    p.g = g; p.x = x; p.w = w; p.y = y; p.koff = (const int*)ws;
    p.tiles_m = (g.Mg + BM - 1) / BM;
    const long long cols = (long long)g.N * g.L;
    p.tiles_n = (int)((cols + BN - 1) / BN);
    const bool aligned_a = (g.Mg % BM == 0) && (K % BK == 0) && al16(w);
    dim3 grid(p.tiles_m * p.tiles_n, 1, groups);
    rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
    if (rc) return rc;
    if (aligned_a) hipLaunchKernelGGL((conv_fwd_kernel<true>), grid, dim3(NT), 0, dev->compute, p);
    else hipLaunchKernelGGL((conv_fwd_kernel<false>), grid, dim3(NT), 0, dev->compute, p);
    NK_LAUNCH_CHECK();
    return nk_prof_stop(dev);
}

int nk_conv_bwd_input(nk_device* dev, int nd, float* dx, const int* x_shape, const float* gy, const float* w,
                      const int* w_shape, const int* stride, const int* dilation, int groups) {
    NK_USE(dev);
    ConvGeom g;
    int rc = make_geom(nd, x_shape, w_shape, stride, dilation, groups, &g);
    if (rc) return rc;
    if ((long long)g.N * g.Cin * g.inplane == 0 || (long long)g.Cout * g.L == 0) return NK_OK;
    NK_CHECK(dx && gy && w, "null pointer in nk_conv_bwd_input");
    const int K = g.Mg * g.KK;
    const size_t wt_bytes = round256((size_t)g.Cout * g.Cg * g.KK * sizeof(float));
    const size_t kt_bytes = round256((size_t)K * sizeof(int4));
    void* ws = nullptr;
    rc = nk_workspace(dev, wt_bytes + kt_bytes, &ws);
    if (rc) return rc;
    float* wt = (float*)ws;
    int4* ktab = (int4*)((char*)ws + wt_bytes);
    hipLaunchKernelGGL(conv_wt_kernel, dim3(nk_stream_grid((size_t)g.Cout * g.Cg * g.KK, 256)), dim3(256), 0, dev->compute, wt, w, g);
    NK_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_ktab_kernel, dim3((K + 255) / 256), dim3(256), 0, dev->compute, ktab, g);
    NK_LAUNCH_CHECK();
    BwdInArgs p{};
    p.g = g; p.dx = dx; p.gy = gy; p.wt = wt; p.ktab = ktab;
    p.tiles_m = (g.Cg + BM - 1) / BM;
    const long long cols = (long long)g.N * g.inplane;
    p.tiles_n = (int)((cols + BN - 1) / BN);
    const bool aligned_a = (g.Cg % BM == 0) && (K % BK == 0);
    const bool unit = g.stride[0] == 1 && g.stride[1] == 1 && g.stride[2] == 1;
    dim3 grid(p.tiles_m * p.tiles_n, 1, groups);
    rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
    if (rc) return rc;
    if (aligned_a && unit) hipLaunchKernelGGL((conv_bwd_input_kernel<true, true>), grid, dim3(NT), 0, dev->compute, p);
    else if (aligned_a) hipLaunchKernelGGL((conv_bwd_input_kernel<true, false>), grid, dim3(NT), 0, dev->compute, p);
    else if (unit) hipLaunchKernelGGL((conv_bwd_input_kernel<false, true>), grid, dim3(NT), 0, dev->compute, p);
    else hipLaunchKernelGGL((conv_bwd_input_kernel<false, false>), grid, dim3(NT), 0, dev->compute, p);
    NK_LAUNCH_CHECK();
    return nk_prof_stop(dev);
}

int nk_conv_bwd_kernel(nk_device* dev, int nd, float* dw, const int* w_shape, const float* gy, const float* x,
                       const int* x_shape, const int* stride, const int* dilation, int groups) {
    NK_USE(dev);
    ConvGeom g;
    int rc = make_geom(nd, x_shape, w_shape, stride, dilation, groups, &g);
    if (rc) return rc;
    const long long R = (long long)g.N * g.L;
    const int Kc = g.Cg * g.KK;
    if ((long long)g.Cout * Kc == 0 || R == 0) return NK_OK;
    NK_CHECK(dw && gy && x, "null pointer in nk_conv_bwd_kernel");
    BwdKArgs p{};
    p.g = g; p.gy = gy; p.x = x;
    p.tiles_m = (g.Mg + BM - 1) / BM;
    p.tiles_n = (Kc + BN - 1) / BN;
    const long long tiles = (long long)p.tiles_m * p.tiles_n * groups;
    const long long rtiles = (R + BK - 1) / BK;
    long long splits = (3LL * dev->num_cus + tiles - 1) / tiles;   // ~3 blocks per CU
    if (splits > rtiles) splits = rtiles;
    if (splits > 1024) splits = 1024;
    if (splits < 1) splits = 1;
    long long rts = (rtiles + splits - 1) / splits;
    splits = (rtiles + rts - 1) / rts;
    p.r_per_split = rts * BK;
    const size_t ko_bytes = round256((size_t)Kc * sizeof(int));
    const long long dw_elems = (long long)g.Cout * Kc;
    void* ws = nullptr;
    rc = nk_workspace(dev, ko_bytes + (size_t)splits * dw_elems * sizeof(float), &ws);
    if (rc) return rc;
    int* koff = (int*)ws;
    p.koff = koff;
    p.slabs = (float*)((char*)ws + ko_bytes);
    hipLaunchKernelGGL(conv_koff_kernel, dim3((Kc + 255) / 256), dim3(256), 0, dev->compute, koff, g);
    NK_LAUNCH_CHECK();
    dim3 grid(p.tiles_m * p.tiles_n, (unsigned)splits, groups);
    const bool vec_g = (g.L % 4 == 0) && al16(gy);
    rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
    if (rc) return rc;
    if (vec_g) hipLaunchKernelGGL((conv_bwd_kernel_kernel<true>), grid, dim3(NT), 0, dev->compute, p);
    else hipLaunchKernelGGL((conv_bwd_kernel_kernel<false>), grid, dim3(NT), 0, dev->compute, p);
    NK_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3(nk_stream_grid((size_t)dw_elems, 256)), dim3(256), 0, dev->compute, dw,
                       p.slabs, dw_elems, (int)splits);
    NK_LAUNCH_CHECK();
    return nk_prof_stop(dev);
}

}  // extern "C"
