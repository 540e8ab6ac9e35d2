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
// Three kernel families, chosen per call by the channel counts per group:
//   multiples of 32  -> "fast" kernels: tap-major k (one k-tile = 32 channels of ONE kernel tap), 16-byte gathers of
//                       row-padded column quads, stride phases in the input-gradient pass;
//   other counts     -> "generic" kernels: offset tables fetched one k-tile ahead, scalar / vector gathers;
//   <= 16 both ways  -> direct (non-MFMA, HBM-bound) kernels: depthwise and small grouped convolutions.
// All staging is branch-free (unconditional loads at clamped addresses, masks applied behind the MFMAs): see DESIGN.md
// 4.1 for what a conditional load costs.
#include "nk_mma.h"

using namespace nkmma;

namespace {

struct ConvGeom {
    int N, Cin, Cout, groups, Cg, Mg;  // Cg = Cin/groups, Mg = Cout/groups
    int in[3], out[3], k[3], stride[3], dil[3];  // padded in front with 1s to 3 spatial dims
    int inplane, L, KK;                           // prod(in), prod(out), prod(k)
    const float* bias;                            // forward: optional per-output-channel bias added in the epilogue
    int assign;                                   // backward: write instead of `+=` (destination's zero fill pending)
    // backward-input through a zero Pad node: dX has the UNPADDED extents `uin` and input coordinate q of dX is
    // coordinate q + pad of the (virtual) padded input `in`.  pad = 0, uin = in otherwise.
    int uin[3], pad[3], uinplane;
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

// QUADV: unit stride on the innermost axis and out[2] % 4 == 0 - the four columns a thread stages are neighbours in one
// output row for EVERY thread, so a staged row is one unaligned 16-byte load; otherwise four scalar loads.  Either way the
// staging is branch-free: loads are unconditional at addresses clamped into the tensor, the masks (k beyond K, columns
// beyond the batch) are applied after the MFMAs, and the koff entries of a k-tile are fetched one k-tile ahead so that the
// gathers never wait for their own offsets.
template <bool ALIGNED_A, int TI, bool QUADV>
__global__ __launch_bounds__(NT, 2) void conv_fwd_kernel(FwdArgs p) {
    constexpr int TJ = 2, BM = 64 * TI, BN = 64 * TJ;
    constexpr int TA_FLOATS = tile_floats<true, BM>(), STAGE = TA_FLOATS + tile_floats<false, BN>();
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
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
    int offn0, offn1, offn2, offn3;  // koff of the rows of the k-tile staged NEXT
    auto load_off = [&](int k0) {
        const int k = k0 + krow;
        offn0 = p.koff[min(k, K - 1)]; offn1 = p.koff[min(k + 8, K - 1)];
        offn2 = p.koff[min(k + 16, K - 1)]; offn3 = p.koff[min(k + 24, K - 1)];
    };
    Stage<4> rb;
    int kbase = 0;  // first k of the tile in rb
    auto gather = [&](int k0) {
        kbase = k0;
        const int o0 = offn0, o1 = offn1, o2 = offn2, o3 = offn3;
        if constexpr (QUADV) {
#define NK_LDU(V, O) { const f32x4u q = *reinterpret_cast<const f32x4u*>(X + b0 + O); V = make_float4(q.x, q.y, q.z, q.w); }
            NK_LDU(rb.v0, o0) NK_LDU(rb.v1, o1) NK_LDU(rb.v2, o2) NK_LDU(rb.v3, o3)
#undef NK_LDU
        } else {
#define NK_LDS(V, O) V = make_float4(X[b0 + O], X[b1 + O], X[b2 + O], X[b3 + O]);
            NK_LDS(rb.v0, o0) NK_LDS(rb.v1, o1) NK_LDS(rb.v2, o2) NK_LDS(rb.v3, o3)
#undef NK_LDS
        }
        load_off(k0 + BK);
    };
    auto gather_finish = [&]() {  // after the MFMAs
        pin_regs(rb.v0); pin_regs(rb.v1); pin_regs(rb.v2); pin_regs(rb.v3);
        const int k = kbase + krow;
        auto keep = [&](float4& q, bool kv) {
            q.x = kv && v0 ? q.x : 0.f; q.y = kv && v1 ? q.y : 0.f; q.z = kv && v2 ? q.z : 0.f; q.w = kv && v3 ? q.w : 0.f;
        };
        keep(rb.v0, k < K); keep(rb.v1, k + 8 < K); keep(rb.v2, k + 16 < K); keep(rb.v3, k + 24 < K);
    };

    f32x16 acc[TI][TJ];
    acc_zero<TI, TJ>(acc);
    TileLoader<true, BM> la;
    la.init(W, K, m0, 0, g.Mg, K, t);
    Stage<BM / 32> ra;
    ra = la.template load<ALIGNED_A>(t);
    load_off(0);
    gather(0);
    gather_finish();
    stage_store<true, BM>(smem, ra, t);
    stage_store<false, BN>(smem + TA_FLOATS, rb, t);
    __syncthreads();
    for (int it = 0; it + 1 < nt; ++it) {
        float* cur = smem + (it & 1) * STAGE;
        float* nxt = smem + ((it + 1) & 1) * STAGE;
        ra = la.template load<ALIGNED_A>(t);
        gather((it + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<true, false, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
        gather_finish();
        stage_store<true, BM>(nxt, ra, t);
        stage_store<false, BN>(nxt + TA_FLOATS, rb, t);
        __syncthreads();
    }
    {
        float* cur = smem + ((nt - 1) & 1) * STAGE;
        mma_tile<true, false, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
    }
    // Y[n][grp*Mg + co][l]
    float* Y = p.y;
    const float* bias = g.bias;
    const int Mg = g.Mg, L = g.L, Cout = g.Cout;
    // the bias of the 16*TI rows this lane owns, loaded before the first store (a load between stores waits for them)
    float bv[TI][16];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = m0 + (wr * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            bv[i][e] = (bias && co < Mg) ? bias[grp * Mg + co] : 0.f;
        }
    // ... and added in registers before the (per-element conditional) stores: with loads still pending when the store
    // blocks are entered, each of them gets its own vmcnt(0), which also waits for the PREVIOUS STORE to be acknowledged
    if (bias) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += bv[i][e];
    }
    // one (n, l) decode per owned column instead of one per element
    acc_foreach_cols<TI, TJ>(acc, wr, wc, lane,
        [&](int c) -> long long {
            const long long cc = (long long)n0 + c;
            if (cc >= cols) return -1;
            const long long n = cc / L;
            return (n * Cout + grp * Mg) * L + (cc - n * L);
        },
        [&](int r, long long base, float v) {
            const int co = m0 + r;
            if (co < Mg && base >= 0) Y[base + (long long)co * L] = v;
        });
}

// dX[cbase[j] + ci * inplane] (+)= acc.  `+=`: every old value is loaded and added in registers before the first store (a
// one-walk `*d += v` is 16*TI*TJ serialised load -> store round trips per lane, the store may alias the next load).
#define NK_BWD_INPUT_EPILOGUE                                                                                        \
    if (!assign) {                                                                                                   \
        float old[TI][TJ][16];                                                                                       \
        acc_foreach_idx<TI, TJ>(acc, wr, wc, lane, [&](int i, int j, int e, int r, int, float) {                     \
            const int ci = m0 + r;                                                                                   \
            old[i][j][e] = (ci < Cg && cbase[j] >= 0) ? DX[cbase[j] + (long long)ci * inplane] : 0.f;                \
        });                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                               \
            _Pragma("unroll") for (int j = 0; j < TJ; ++j)                                                           \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[i][j][e] += old[i][j][e];                         \
    }                                                                                                                \
    acc_foreach_idx<TI, TJ>(acc, wr, wc, lane, [&](int, int j, int, int r, int, float v) {                           \
        const int ci = m0 + r;                                                                                       \
        if (ci < Cg && cbase[j] >= 0) DX[cbase[j] + (long long)ci * inplane] = v;                                    \
    });

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

template <bool ALIGNED_A, bool UNIT_STRIDE, int TI>
__global__ __launch_bounds__(NT, 2) void conv_bwd_input_kernel(BwdInArgs p) {
    constexpr int TJ = 2, BM = 64 * TI, BN = 64 * TJ;
    constexpr int TA_FLOATS = tile_floats<true, BM>(), STAGE = TA_FLOATS + tile_floats<false, BN>();
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    const ConvGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    const int grp = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = g.Mg * g.KK;
    const long long cols = (long long)g.N * g.uinplane;
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
        const int n = V ? (int)(cc / g.uinplane) : 0;                                   \
        int q = V ? (int)(cc % g.uinplane) : 0;                                         \
        PC = q % g.uin[2] + g.pad[2]; q /= g.uin[2];                                    \
        PB = q % g.uin[1] + g.pad[1];                                                   \
        PA = q / g.uin[1] + g.pad[0];                                                   \
        GB = (long long)n * g.Cout * g.L;                                               \
    }
        NK_COL(0, gb0, pa0, pb0, pc0, v0) NK_COL(1, gb1, pa1, pb1, pc1, v1)
        NK_COL(2, gb2, pa2, pb2, pc2, v2) NK_COL(3, gb3, pa3, pb3, pc3, v3)
#undef NK_COL
    }
    // Branch-free staging: the ktab entries of a k-tile are fetched one k-tile ahead, the 16 gradient elements a thread
    // stages per k-tile are loaded unconditionally (offset 0 when the (column, tap) pair has no output position) and the
    // validity bits are applied after the MFMAs.
    int4 ktn0, ktn1, ktn2, ktn3;  // ktab rows of the k-tile staged NEXT
    auto load_kt = [&](int k0) {
        const int k = k0 + krow;
        ktn0 = p.ktab[min(k, K - 1)]; ktn1 = p.ktab[min(k + 8, K - 1)];
        ktn2 = p.ktab[min(k + 16, K - 1)]; ktn3 = p.ktab[min(k + 24, K - 1)];
    };
    Stage<4> rb;
    unsigned okbits = 0;  // bit 4*j + i: element (row j, column i) of rb is a real gradient element
    auto elem = [&](const int4 kt, bool kv, long long gb, int pa, int pb, int pc, bool v, bool& ok) -> long long {
        int a = pa - kt.y, b = pb - kt.z, c = pc - kt.w;
        ok = kv && v && a >= 0 && b >= 0 && c >= 0;
        if (!UNIT_STRIDE) {
            ok = ok && (a % g.stride[0] == 0) && (b % g.stride[1] == 0) && (c % g.stride[2] == 0);
            a /= g.stride[0]; b /= g.stride[1]; c /= g.stride[2];
        }
        ok = ok && a < g.out[0] && b < g.out[1] && c < g.out[2];
        return ok ? gb + kt.x + (a * g.out[1] + b) * g.out[2] + c : 0;
    };
    auto gather = [&](int k0) {
        unsigned bits = 0;
#define NK_ROW(j, V, KT)                                                                \
    {                                                                                   \
        const bool kv = k0 + krow + 8 * j < K;                                          \
        bool o0, o1, o2, o3;                                                            \
        const long long e0 = elem(KT, kv, gb0, pa0, pb0, pc0, v0, o0), e1 = elem(KT, kv, gb1, pa1, pb1, pc1, v1, o1), \
                        e2 = elem(KT, kv, gb2, pa2, pb2, pc2, v2, o2), e3 = elem(KT, kv, gb3, pa3, pb3, pc3, v3, o3); \
        V = make_float4(G[e0], G[e1], G[e2], G[e3]);                                    \
        bits |= ((o0 ? 1u : 0u) | (o1 ? 2u : 0u) | (o2 ? 4u : 0u) | (o3 ? 8u : 0u)) << (4 * j); \
    }
        NK_ROW(0, rb.v0, ktn0) NK_ROW(1, rb.v1, ktn1) NK_ROW(2, rb.v2, ktn2) NK_ROW(3, rb.v3, ktn3)
#undef NK_ROW
        okbits = bits;
        load_kt(k0 + BK);
    };
    auto gather_finish = [&]() {  // after the MFMAs
        pin_regs(rb.v0); pin_regs(rb.v1); pin_regs(rb.v2); pin_regs(rb.v3);
        auto keep = [&](float4& q, unsigned m) {
            q.x = (m & 1u) ? q.x : 0.f; q.y = (m & 2u) ? q.y : 0.f; q.z = (m & 4u) ? q.z : 0.f; q.w = (m & 8u) ? q.w : 0.f;
        };
        keep(rb.v0, okbits); keep(rb.v1, okbits >> 4); keep(rb.v2, okbits >> 8); keep(rb.v3, okbits >> 12);
    };

    f32x16 acc[TI][TJ];
    acc_zero<TI, TJ>(acc);
    TileLoader<true, BM> la;
    la.init(Wt, K, m0, 0, g.Cg, K, t);
    Stage<BM / 32> ra;
    ra = la.template load<ALIGNED_A>(t);
    load_kt(0);
    gather(0);
    gather_finish();
    stage_store<true, BM>(smem, ra, t);
    stage_store<false, BN>(smem + TA_FLOATS, rb, t);
    __syncthreads();
    for (int it = 0; it + 1 < nt; ++it) {
        float* cur = smem + (it & 1) * STAGE;
        float* nxt = smem + ((it + 1) & 1) * STAGE;
        ra = la.template load<ALIGNED_A>(t);
        gather((it + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<true, false, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
        gather_finish();
        stage_store<true, BM>(nxt, ra, t);
        stage_store<false, BN>(nxt + TA_FLOATS, rb, t);
        __syncthreads();
    }
    {
        float* cur = smem + ((nt - 1) & 1) * STAGE;
        mma_tile<true, false, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
    }
    // dX[n][grp*Cg + ci][pos] += acc
    float* DX = p.dx;
    const int assign = g.assign;
    const int Cg = g.Cg, Cin = g.Cin, inplane = g.uinplane;
    long long cbase[TJ];  // one (n, pos) decode per owned column
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const long long cc = (long long)n0 + (wc * TJ + j) * 32 + (lane & 31);
        const long long n = cc / inplane;
        cbase[j] = cc < cols ? (n * Cin + grp * Cg) * inplane + (cc - n * inplane) : -1;
    }
    NK_BWD_INPUT_EPILOGUE
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

// QUADR (unit stride on the innermost axis, out[2] >= 4): the reduction runs over (n, o0, o1, c') with the innermost output
// row padded to W4 = a multiple of 4, so the four consecutive reduction indices a thread stages are one output-row quad:
// one (incremental, division-free) decode per k-tile and one 16-byte load per staged row.  A quad that would run past the
// row end (out[2] % 4 != 0) is loaded `dup` elements earlier - for BOTH operands, a reduction does not care where in the
// k-tile an element sits - and its first `dup` elements, already counted by the previous quad, are masked.
template <bool VEC_G, int TI, int TJ, bool QUADR>
__global__ __launch_bounds__(NT, 2) void conv_bwd_kernel_kernel(BwdKArgs p) {
    constexpr int BM = 64 * TI, BN = 64 * TJ;
    constexpr int TA_FLOATS = tile_floats<true, BM>(), STAGE = TA_FLOATS + tile_floats<true, BN>();
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    const ConvGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    // 1-D grid over (split, tile): the tiles of one split read the same G / X slices, so they are
    // made neighbours in the per-XCD chunk order (shared through that XCD's L2)
    const int ntile = p.tiles_m * p.tiles_n;
    int split, tile;
    tile_coords(blockIdx.x, gridDim.x, 1, (int)gridDim.x, tile, split);  // split := XCD-chunked linear id
    tile = split % ntile;
    split /= ntile;
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int grp = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Kc = g.Cg * g.KK;  // columns of dW
    const int W4 = (g.out[2] + 3) & ~3;
    const long long R = QUADR ? (long long)g.N * g.out[0] * g.out[1] * W4 : (long long)g.N * g.L;
    const long long rbeg = split * p.r_per_split;
    const long long rend = rbeg + p.r_per_split < R ? rbeg + p.r_per_split : R;
    const int nt = rend > rbeg ? (int)((rend - rbeg + BK - 1) / BK) : 0;
    const float* G = p.gy + (long long)grp * g.Mg * g.L;
    const float* X = p.x + (long long)grp * g.Cg * g.inplane;

    // KC staging for both operands: idx = t + 256*j -> row = kc_row(t) + 32*j, 4 consecutive r
    const int rq = kc_q(t), row = kc_row(t);
    // B: columns n0 + row + 32*j -> koff (fixed over the k loop)
    int ko0, ko1, ko2 = 0, ko3 = 0;
    bool cv0, cv1, cv2 = false, cv3 = false;
#define NK_KO(j, KO, CV) { const int c = n0 + row + 32 * j; CV = c < Kc; KO = CV ? p.koff[c] : 0; }
    NK_KO(0, ko0, cv0) NK_KO(1, ko1, cv1)
    if constexpr (TJ == 2) { NK_KO(2, ko2, cv2) NK_KO(3, ko3, cv3) }
#undef NK_KO
    // A: rows (co) m0 + row + 32*j
    const bool av0 = m0 + row < g.Mg, av1 = m0 + row + 32 < g.Mg, av2 = m0 + row + 64 < g.Mg, av3 = m0 + row + 96 < g.Mg;

    Stage<BM / 32> ra;
    Stage<BN / 32> rb;
    // QUADR state: (sample, output coordinates) of this thread's first index in the current tile
    int qn = 0, q0 = 0, q1 = 0, q2 = 0;
    if (QUADR) {
        const long long r = rbeg + rq * 4;
        long long rowid = r / W4;
        q2 = (int)(r - rowid * W4);
        q1 = (int)(rowid % g.out[1]); rowid /= g.out[1];
        q0 = (int)(rowid % g.out[0]);
        qn = (int)(rowid / g.out[0]);
    }
    // QUADR: branch-free staging.  Every load is unconditional at an address clamped into the tensor (row / column / quad
    // offsets of masked lanes are 0) and masked lanes select zeros afterwards: conditional loads whose two arms write the
    // same registers made the compiler wait (vmcnt(0)) before each of the eight loads of a k-tile, i.e. eight serialised
    // memory round trips per k-tile instead of one hidden behind the MFMAs.
    const long long aro0 = av0 ? (long long)(m0 + row) * g.L : 0, aro1 = av1 ? (long long)(m0 + row + 32) * g.L : 0,
                    aro2 = av2 ? (long long)(m0 + row + 64) * g.L : 0, aro3 = av3 ? (long long)(m0 + row + 96) * g.L : 0;
    bool qv = false;  // the quad staged last lies inside [rbeg, rend)
    int qdup = 0;     // its first `qdup` elements belong to the previous quad of the row
    auto load_quad = [&](long long r0) {
        const bool v = r0 + rq * 4 < rend;
        qv = v;
        const int cs = min(q2, g.out[2] - 4);  // start clamped so that the quad ends inside the row
        qdup = q2 - cs;
        const long long x0 = v ? (long long)qn * g.Cin * g.inplane + ((q0 * g.stride[0] * g.in[1] + q1 * g.stride[1]) * g.in[2] + cs) : 0;
        const long long g0 = v ? (long long)qn * g.Cout * g.L + ((q0 * g.out[1] + q1) * g.out[2] + cs) : 0;
        q2 += BK;  // next k-tile: 32 positions further along the (row-padded) reduction index
        while (q2 >= W4) { q2 -= W4; ++q1; }
        while (q1 >= g.out[1]) { q1 -= g.out[1]; ++q0; }
        while (q0 >= g.out[0]) { q0 -= g.out[0]; ++qn; }
#define NK_LDU(V, P) { const f32x4u q = *reinterpret_cast<const f32x4u*>(P); V = make_float4(q.x, q.y, q.z, q.w); }
        NK_LDU(ra.v0, G + g0 + aro0) NK_LDU(ra.v1, G + g0 + aro1)
        if constexpr (TI == 2) { NK_LDU(ra.v2, G + g0 + aro2) NK_LDU(ra.v3, G + g0 + aro3) }
        NK_LDU(rb.v0, X + x0 + ko0) NK_LDU(rb.v1, X + x0 + ko1)
        if constexpr (TJ == 2) { NK_LDU(rb.v2, X + x0 + ko2) NK_LDU(rb.v3, X + x0 + ko3) }
#undef NK_LDU
    };
    // applied AFTER the MFMAs of the current k-tile (touching the loaded registers earlier would wait for the loads)
    auto mask_quad = [&]() {
        // component-wise selects: `cond ? vecA : vecB` on the vector CLASS selects between two addresses and sends both
        // through scratch memory
        const bool d0 = qdup <= 0, d1 = qdup <= 1, d2 = qdup <= 2;  // element i is new when i >= qdup (qdup <= 3)
        auto keep = [&](float4& q, bool k) {
            q.x = k && d0 ? q.x : 0.f; q.y = k && d1 ? q.y : 0.f; q.z = k && d2 ? q.z : 0.f; q.w = k ? q.w : 0.f;
        };
        keep(ra.v0, qv && av0); keep(ra.v1, qv && av1);
        if constexpr (TI == 2) { keep(ra.v2, qv && av2); keep(ra.v3, qv && av3); }
        keep(rb.v0, qv && cv0); keep(rb.v1, qv && cv1);
        if constexpr (TJ == 2) { keep(rb.v2, qv && cv2); keep(rb.v3, qv && cv3); }
    };
    // General form (strided innermost axis or rows shorter than 4): per-element decode, scalar gathers - still branch-free
    // (offsets of masked elements are 0, masks applied after the MFMAs).
    int smask = 0;  // bit c: reduction index r0 + 4*rq + c lies inside [rbeg, rend)
    auto load_scalar = [&](long long r0) {
        long long xo[4], go[4];
        int m = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long r = r0 + rq * 4 + c;
            const bool ok = r < rend;
            const int n = ok ? (int)(r / g.L) : 0, l = ok ? (int)(r % g.L) : 0;
            xo[c] = (long long)n * g.Cin * g.inplane + window_origin(g, l);
            go[c] = (long long)n * g.Cout * g.L + l;
            m |= (ok ? 1 : 0) << c;
        }
        smask = m;
#define NK_A(V, ARO)                                                                                        \
    if constexpr (VEC_G) { /* L % 4 == 0: the four indices are one aligned quad of one sample */           \
        V = *reinterpret_cast<const float4*>(G + go[0] + ARO);                                              \
    } else {                                                                                                \
        V = make_float4(G[go[0] + ARO], G[go[1] + ARO], G[go[2] + ARO], G[go[3] + ARO]);                    \
    }
        NK_A(ra.v0, aro0) NK_A(ra.v1, aro1)
        if constexpr (TI == 2) { NK_A(ra.v2, aro2) NK_A(ra.v3, aro3) }
#undef NK_A
#define NK_B(V, KO) V = make_float4(X[xo[0] + KO], X[xo[1] + KO], X[xo[2] + KO], X[xo[3] + KO]);
        NK_B(rb.v0, ko0) NK_B(rb.v1, ko1)
        if constexpr (TJ == 2) { NK_B(rb.v2, ko2) NK_B(rb.v3, ko3) }
#undef NK_B
    };
    auto mask_scalar = [&]() {
        const bool m0_ = smask & 1, m1_ = smask & 2, m2_ = smask & 4, m3_ = smask & 8;
        auto keep = [&](float4& q, bool k) {
            q.x = k && m0_ ? q.x : 0.f; q.y = k && m1_ ? q.y : 0.f; q.z = k && m2_ ? q.z : 0.f; q.w = k && m3_ ? q.w : 0.f;
        };
        keep(ra.v0, av0); keep(ra.v1, av1);
        if constexpr (TI == 2) { keep(ra.v2, av2); keep(ra.v3, av3); }
        keep(rb.v0, cv0); keep(rb.v1, cv1);
        if constexpr (TJ == 2) { keep(rb.v2, cv2); keep(rb.v3, cv3); }
    };
    auto load_both = [&](long long r0) {
        if constexpr (QUADR) load_quad(r0);
        else load_scalar(r0);
    };

    f32x16 acc[TI][TJ];
    acc_zero<TI, TJ>(acc);
    if (nt > 0) {
        load_both(rbeg);
        if constexpr (QUADR) mask_quad();
        else mask_scalar();
        stage_store<true, BM>(smem, ra, t);
        stage_store<true, BN>(smem + TA_FLOATS, rb, t);
    }
    __syncthreads();
    for (int it = 0; it + 1 < nt; ++it) {
        float* cur = smem + (it & 1) * STAGE;
        float* nxt = smem + ((it + 1) & 1) * STAGE;
        load_both(rbeg + (long long)(it + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<true, true, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
        if constexpr (QUADR) mask_quad();
        else mask_scalar();
        stage_store<true, BM>(nxt, ra, t);
        stage_store<true, BN>(nxt + TA_FLOATS, rb, t);
        __syncthreads();
    }
    if (nt > 0) {
        float* cur = smem + ((nt - 1) & 1) * STAGE;
        mma_tile<true, true, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
    }
    float* S = p.slabs + ((long long)split * g.groups + grp) * (long long)g.Mg * Kc;
    const int Mg = g.Mg;
    acc_foreach<TI, TJ>(acc, wr, wc, lane, [&](int r, int c, float v) {
        const int co = m0 + r, col = n0 + c;
        if (co < Mg && col < Kc) S[(long long)co * Kc + col] = v;
    });
}

// dW[i] += sum_s slabs[s][i].  64 elements x 4 split-lanes per block (lane j sums splits j, j+4, ... with two independent
// accumulators), folded through LDS in a fixed order: `splits/4` loads deep instead of `splits` (the serial form took 28 us
// for 30 MB at C3).  Deterministic.
__global__ void conv_dw_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slabs, long long n, int splits, int assign) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int k = lane;
        for (; k + 4 < splits; k += 8) {
            s0 += slabs[(long long)k * n + i];
            s1 += slabs[(long long)(k + 4) * n + i];
        }
        if (k < splits) s0 += slabs[(long long)k * n + i];
    }
    red[lane][col] = s0 + s1;
    __syncthreads();
    if (lane == 0 && i < n) {
        const float s = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        dw[i] = assign ? s : dw[i] + s;
    }
}

// =================================================================================================
// Fast paths (tap-major reduction order).  When the channel count per group is a multiple of
// 32, a k-tile of 32 covers 32 channels of ONE kernel tap, so the tap decode / border test is
// done once per k-tile (wave-uniform, scalar) instead of once per k row, the per-row address is
// `base + row * plane`, and — for unit stride along the innermost axis — the four columns a
// thread stages are one unaligned 16-B load.  The weights are re-ordered once per call by a
// tiny pre-kernel (they are KBs to MBs; the activations are hundreds of MBs).
// =================================================================================================
// tapoff[tap] = input offset of kernel tap `tap` relative to the window origin
__global__ void conv_tapoff_kernel(int* __restrict__ tapoff, int4* __restrict__ tapd, ConvGeom g) {
    for (int tap = blockIdx.x * blockDim.x + threadIdx.x; tap < g.KK; tap += gridDim.x * blockDim.x) {
        int rem = tap;
        const int k2 = rem % g.k[2]; rem /= g.k[2];
        const int k1 = rem % g.k[1];
        const int k0 = rem / g.k[1];
        tapoff[tap] = (k0 * g.dil[0] * g.in[1] + k1 * g.dil[1]) * g.in[2] + k2 * g.dil[2];
        tapd[tap] = make_int4(k0 * g.dil[0], k1 * g.dil[1], k2 * g.dil[2], 0);
    }
}
// Wp[grp][co][tap][ci] = W[grp*Mg + co][ci][tap]   (forward A operand, k = tap*Cg + ci)
__global__ void conv_wp_kernel(float* __restrict__ wp, const float* __restrict__ w, ConvGeom g) {
    const long long total = (long long)g.Cout * g.Cg * g.KK;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % g.Cg);
        long long rem = i / g.Cg;
        const int tap = (int)(rem % g.KK);
        const long long co = rem / g.KK;  // absolute output channel
        wp[i] = w[(co * g.Cg + ci) * g.KK + tap];
    }
}
// ---- backward-input: stride phases ------------------------------------------------------------------------------
// Input coordinate a (in the padded frame) receives kernel tap k only when (a - k*dil) is a multiple of the stride, i.e.
// for the taps with k*dil = a (mod stride).  The input positions therefore fall into prod(stride) residue classes
// ("phases"), each with its own subset of the taps; inside one phase, stepping the input coordinate by `stride` steps
// the output coordinate by 1, so every phase is a UNIT-stride gather over the gradient: out = q + e(tap) with
// q = (a - r)/stride and e = (r - k*dil)/stride.  Unit stride is the one-phase case (all taps, e = -k*dil).
constexpr int MAX_PHASES = 16;
struct BwdInPhase {
    int first[3];   // first UNPADDED input coordinate of the class on each axis
    int count[3];   // number of input coordinates of the class on each axis (0: the class is empty)
    int q0[3];      // class-local coordinate i (input coordinate first + i*stride) <-> output-frame coordinate q0 + i
    int tap_begin, ntaps;  // its taps in the phase-sorted tap table
    int tile_begin;        // first column tile of the phase in the launch
};
struct BwdInPhaseTable { BwdInPhase ph[MAX_PHASES]; };
// by-value table -> device memory (indexing a by-value array with a run-time index would spill it to scratch; a kernel
// instead of a host copy keeps the call capturable in a hipGraph)
__global__ void conv_phase_table_kernel(BwdInPhase* __restrict__ out, BwdInPhaseTable tbl) {
#pragma unroll
    for (int i = 0; i < MAX_PHASES; ++i) out[i] = tbl.ph[i];
}
__device__ __forceinline__ int tap_phase(const ConvGeom& g, int tap, int* kd) {
    int rem = tap;
    kd[2] = (rem % g.k[2]) * g.dil[2]; rem /= g.k[2];
    kd[1] = (rem % g.k[1]) * g.dil[1];
    kd[0] = (rem / g.k[1]) * g.dil[0];
    return ((kd[0] % g.stride[0]) * g.stride[1] + kd[1] % g.stride[1]) * g.stride[2] + kd[2] % g.stride[2];
}
// tapd[position in phase order] = {d0, d1, d2, tap} with out = q - d (d = (k*dil - r)/stride >= 0);
// tappos[tap] = {first position of its phase, taps in its phase, its rank inside the phase, phase id}
__global__ void conv_phase_taps_kernel(int4* __restrict__ tapd, int4* __restrict__ tappos, ConvGeom g) {
    for (int tap = blockIdx.x * blockDim.x + threadIdx.x; tap < g.KK; tap += gridDim.x * blockDim.x) {
        int kd[3], kd2[3];
        const int pid = tap_phase(g, tap, kd);
        int begin = 0, cnt = 0, rank = 0;
        for (int t2 = 0; t2 < g.KK; ++t2) {
            const int pid2 = tap_phase(g, t2, kd2);
            if (pid2 < pid) ++begin;
            else if (pid2 == pid) { ++cnt; if (t2 < tap) ++rank; }
        }
        tapd[begin + rank] = make_int4((kd[0] - kd[0] % g.stride[0]) / g.stride[0], (kd[1] - kd[1] % g.stride[1]) / g.stride[1],
                                       (kd[2] - kd[2] % g.stride[2]) / g.stride[2], tap);
        tappos[tap] = make_int4(begin, cnt, rank, pid);
    }
}
// Wq[grp][ci][phase][chunk][tap in phase][c32] = W[grp*Mg + chunk*32 + c32][ci][tap]   (backward-input A operand).  Per
// phase, k runs over 32-channel chunks of co with the taps INSIDE a chunk: the 32 x (tile + halo) slab of the gradient
// that one chunk needs is then re-read by all taps back to back (L2 hits) instead of once per tap across all of co (PMC:
// 1.7 GB fetched per launch at C3 with the tap-major order, 9x the gradient).
__global__ void conv_wq_kernel(float* __restrict__ wq, const float* __restrict__ w, const int4* __restrict__ tappos, ConvGeom g) {
    const long long total = (long long)g.Cout * g.Cg * g.KK;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {  // i = source index ((grp*Mg + co)*Cg + ci)*KK + tap
        const int tap = (int)(i % g.KK);
        long long rem = i / g.KK;
        const int ci = (int)(rem % g.Cg); rem /= g.Cg;
        const int co = (int)(rem % g.Mg);
        const int grp = (int)(rem / g.Mg);
        const int4 tp = tappos[tap];
        const int chunk = co / BK, c32 = co - chunk * BK;
        wq[((long long)grp * g.Cg + ci) * ((long long)g.Mg * g.KK) + (long long)g.Mg * tp.x + (chunk * tp.y + tp.z) * BK + c32] = w[i];
    }
}

struct FastFwdArgs {
    ConvGeom g;
    const float* x;
    const float* wp;
    float* y;
    const int* tapoff;
    int tiles_m, tiles_n;
    // Tail balancing: the first `full_blocks` tiles (whole waves of resident blocks) are computed by one block each;
    // every remaining tile is split over `tail_splits` blocks of `tail_kts` k-tiles that write partial tiles to
    // `slabs` ([tail tile][split][BM][BN]); conv_tail_reduce_kernel sums them in split order.  tail_splits == 0: off.
    int full_blocks, tail_splits, tail_kts;
    float* slabs;
};

// requires Cg % 32 == 0, stride[2] == 1, out[2] >= 4, per-tensor element counts < 2^31.
// Columns are (n, o0, o1, c') with the innermost output row padded to W4 = a multiple of 4, so the quad a thread stages is
// four consecutive positions of ONE output row = one unaligned 16-byte load per staged row.  RP (out[2] % 4 != 0): the
// last quad of a row is loaded `dup` elements earlier (so that it ends inside the input row) and shifted left by `dup`
// behind the MFMAs; its trailing `dup` columns are dummies whose accumulators are never stored.
template <bool ALIGNED_A, int TI, bool RP>
__global__ __launch_bounds__(NT, 2) void conv_fwd_fast_kernel(FastFwdArgs p) {
    constexpr int TJ = 2, BM = 64 * TI, BN = 64 * TJ;
    constexpr int TA_FLOATS = tile_floats<true, BM>(), STAGE = TA_FLOATS + tile_floats<false, BN>();
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    const ConvGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    int tm, tn;
    const int K = g.Cg * g.KK, tpt = g.Cg / BK;  // k-tiles per tap
    int kt0 = 0, nt = K / BK;
    float* slab = nullptr;
    if (p.tail_splits == 0) {
        tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    } else if ((int)blockIdx.x < p.full_blocks) {
        tile_of_seq(xcd_chunk(blockIdx.x, p.full_blocks), p.tiles_m, p.tiles_n, tm, tn);
    } else {
        const int tb = blockIdx.x - p.full_blocks, tail_tile = tb / p.tail_splits, split = tb - tail_tile * p.tail_splits;
        tile_of_seq(p.full_blocks + tail_tile, p.tiles_m, p.tiles_n, tm, tn);
        kt0 = split * p.tail_kts;
        nt = min(nt - kt0, p.tail_kts);
        slab = p.slabs + ((long long)(blockIdx.z * (p.tiles_m * p.tiles_n - p.full_blocks) + tail_tile) * p.tail_splits + split) * (BM * BN);
    }
    const int grp = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int W4 = (g.out[2] + 3) & ~3, rows_per_n = g.out[0] * g.out[1];
    const int cols = g.N * rows_per_n * W4;  // < 2^31 (checked by the host)
    const float* W = p.wp + (long long)grp * g.Mg * K;
    const float* X = p.x + (long long)grp * g.Cg * g.inplane;

    // this thread stages the column quad n0 + 4*cq .. +3 (one output row) for channel rows
    // (t>>5) + 8*j of every k-tile
    const int cq = t & 31, krow = t >> 5;
    const int c0 = n0 + cq * 4;
    const bool valid = c0 < cols;
    int xb = krow * g.inplane, dup = 0;
    if (valid) {
        const int rowid = c0 / W4, oc = c0 - rowid * W4;
        const int n = rowid / rows_per_n, ab = rowid - n * rows_per_n;
        const int oa = ab / g.out[1], ob = ab - oa * g.out[1];
        const int cs = RP ? min(oc, g.out[2] - 4) : oc;
        dup = oc - cs;
        xb = n * g.Cin * g.inplane + (oa * g.stride[0] * g.in[1] + ob * g.stride[1]) * g.in[2] + cs + krow * g.inplane;
    }
    const int jstep = 8 * g.inplane;
    auto gather = [&](int kt) {
        kt += kt0;
        const int tap = kt / tpt, ci0 = (kt - tap * tpt) * BK;
        const float* src = X + (p.tapoff[tap] + ci0 * g.inplane);
        Stage<4> r;
        // unconditional: a quad beyond the last column (xb = krow * inplane) reads real memory and feeds accumulators that
        // are never stored - no branch, no mask
        const f32x4u q0 = *reinterpret_cast<const f32x4u*>(src + xb);
        const f32x4u q1 = *reinterpret_cast<const f32x4u*>(src + xb + jstep);
        const f32x4u q2 = *reinterpret_cast<const f32x4u*>(src + xb + 2 * jstep);
        const f32x4u q3 = *reinterpret_cast<const f32x4u*>(src + xb + 3 * jstep);
        r.v0 = make_float4(q0.x, q0.y, q0.z, q0.w);
        r.v1 = make_float4(q1.x, q1.y, q1.z, q1.w);
        r.v2 = make_float4(q2.x, q2.y, q2.z, q2.w);
        r.v3 = make_float4(q3.x, q3.y, q3.z, q3.w);
        return r;
    };
    // RP: element i of the quad = element i + dup of the loaded vector (register selects, after the MFMAs)
    const bool l1 = dup & 1, l2 = dup & 2;
    auto shift = [&](Stage<4>& r) {
        auto sh = [&](float4& q) {
            float e0 = q.x, e1 = q.y, e2 = q.z, e3 = q.w;
            e0 = l1 ? e1 : e0; e1 = l1 ? e2 : e1; e2 = l1 ? e3 : e2;
            e0 = l2 ? e2 : e0; e1 = l2 ? e3 : e1;
            q.x = e0; q.y = e1; q.z = e2; q.w = e3;
        };
        sh(r.v0); sh(r.v1); sh(r.v2); sh(r.v3);
    };

    f32x16 acc[TI][TJ];
    acc_zero<TI, TJ>(acc);
    TileLoader<true, BM> la;
    la.init(W, K, m0, kt0 * BK, g.Mg, K, t);
    Stage<BM / 32> ra;
    Stage<4> rb;
    ra = la.template load<ALIGNED_A>(t);
    rb = gather(0);
    if constexpr (RP) shift(rb);
    stage_store<true, BM>(smem, ra, t);
    stage_store<false, BN>(smem + TA_FLOATS, rb, t);
    __syncthreads();
    for (int it = 0; it + 1 < nt; ++it) {
        float* cur = smem + (it & 1) * STAGE;
        float* nxt = smem + ((it + 1) & 1) * STAGE;
        ra = la.template load<ALIGNED_A>(t);
        rb = gather(it + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<true, false, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
        if constexpr (RP) shift(rb);
        stage_store<true, BM>(nxt, ra, t);
        stage_store<false, BN>(nxt + TA_FLOATS, rb, t);
        __syncthreads();
    }
    if (nt > 0) {
        float* cur = smem + ((nt - 1) & 1) * STAGE;
        mma_tile<true, false, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
    }
    if (slab) {  // partial tile of a split tail tile
        acc_foreach<TI, TJ>(acc, wr, wc, lane, [&](int r, int c, float v) { slab[r * BN + c] = v; });
        return;
    }
    float* Y = p.y;
    const float* bias = g.bias;
    const int Mg = g.Mg, L = g.L, Cout = g.Cout;
    // the bias of the 16*TI rows this lane owns, loaded before the first store (a load between stores waits for them)
    float bv[TI][16];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = m0 + (wr * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            bv[i][e] = (bias && co < Mg) ? bias[grp * Mg + co] : 0.f;
        }
    // ... and added in registers before the (per-element conditional) stores: with loads still pending when the store
    // blocks are entered, each of them gets its own vmcnt(0), which also waits for the PREVIOUS STORE to be acknowledged
    if (bias) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += bv[i][e];
    }
    // one (n, l) decode per owned column instead of one per element
    acc_foreach_cols<TI, TJ>(acc, wr, wc, lane,
        [&](int c) -> long long {
            const int cc = n0 + c;
            if (cc >= cols) return -1;
            const int rowid = cc / W4, cpos = cc - rowid * W4;
            if (cpos >= g.out[2]) return -1;  // padding column of the row
            const int n = rowid / rows_per_n;
            return ((long long)n * Cout + grp * Mg) * L + (long long)(rowid - n * rows_per_n) * g.out[2] + cpos;
        },
        [&](int r, long long base, float v) {
            const int co = m0 + r;
            if (co < Mg && base >= 0) Y[base + (long long)co * L] = v;
        });
}

// Y[tail tiles] = sum over splits (fixed order) of the partial tiles (+ bias)
template <int BM>
__global__ void conv_fwd_tail_reduce_kernel(FastFwdArgs p) {
    constexpr int BN = 128;
    const ConvGeom& g = p.g;
    const int ntail = p.tiles_m * p.tiles_n - p.full_blocks;
    const int tail_tile = blockIdx.x, grp = blockIdx.z;
    int tm, tn;
    tile_of_seq(p.full_blocks + tail_tile, p.tiles_m, p.tiles_n, tm, tn);
    const float* base = p.slabs + ((long long)(grp * ntail + tail_tile) * p.tail_splits) * (BM * BN);
    const int W4 = (g.out[2] + 3) & ~3, rows_per_n = g.out[0] * g.out[1];
    const int cols = g.N * rows_per_n * W4;  // row-padded column space of the fast kernel, < 2^31
    // blockIdx.y: a 1024-element slice of the tile (8 rows x 128 columns); consecutive threads = consecutive columns
    const int e = blockIdx.y * 1024 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ee = e + i * 256;
        const int r = ee / BN, c = ee - r * BN;
        const int co = tm * BM + r, cc = tn * BN + c;
        if (co >= g.Mg || cc >= cols) continue;
        float s = 0.f;
        for (int k = 0; k < p.tail_splits; ++k) s += base[(long long)k * (BM * BN) + ee];
        const int rowid = cc / W4, cpos = cc - rowid * W4;
        if (cpos >= g.out[2]) continue;
        const int n = rowid / rows_per_n, l = (rowid - n * rows_per_n) * g.out[2] + cpos;
        p.y[((long long)n * g.Cout + grp * g.Mg + co) * g.L + l] = g.bias ? s + g.bias[grp * g.Mg + co] : s;
    }
}

struct FastBwdInArgs {
    ConvGeom g;
    float* dx;
    const float* gy;
    const float* wq;  // [groups][Cg][KK*Mg], phase-sorted (conv_wq_kernel)
    const int4* tapd;
    const BwdInPhase* phases;
    int nphase;
    int tiles_m, tiles_n;  // tiles_n: column tiles of all phases together
};

// requires Mg % 32 == 0, out[2] >= 4, at most MAX_PHASES stride phases, per-tensor element counts < 2^31
template <bool ALIGNED_A, int TI>
__global__ __launch_bounds__(NT, 2) void conv_bwd_input_fast_kernel(FastBwdInArgs p) {
    constexpr int TJ = 2, BM = 64 * TI, BN = 64 * TJ;
    constexpr int TA_FLOATS = tile_floats<true, BM>(), STAGE = TA_FLOATS + tile_floats<false, BN>();
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    const ConvGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    int pid = 0;  // the stride phase this column tile belongs to (block-uniform)
    for (int i = 1; i < p.nphase; ++i)
        if (tn >= p.phases[i].tile_begin) pid = i;
    const BwdInPhase ph = p.phases[pid];
    tn -= ph.tile_begin;
    const int grp = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = g.Mg * g.KK, nt = g.Mg * ph.ntaps / BK;  // K: row length of Wq; this phase reduces over Mg * ntaps
    // Columns are the phase's input positions (n, i0, i1, i2') with the innermost extent padded to W4 = a multiple of 4,
    // so that the quad a thread stages never straddles two rows: for every tap its four gradient elements are then
    // contiguous in memory and ONE 16-byte load per staged row serves interior and border quads alike (start clamped
    // into the row, elements picked by a shift, outside ones masked) - no divergent scalar path.  Cost: W4/count[2] - 1
    // dummy columns.
    const int W4 = (ph.count[2] + 3) & ~3, rows_per_n = ph.count[0] * ph.count[1];
    const int cols = g.N * rows_per_n * W4;  // < 2^31 (checked by the host)
    const float* Wq = p.wq + (long long)grp * g.Cg * K + (long long)g.Mg * ph.tap_begin;
    const float* G = p.gy + (long long)grp * g.Mg * g.L;
    const int4* tapd = p.tapd + ph.tap_begin;
    const int ntaps = ph.ntaps;

    const int cq = t & 31, krow = t >> 5;
    const int cc0 = n0 + cq * 4;
    const bool valid = cc0 < cols;
    int qa = 0, qb = 0, qc = 0, gbase = 0;
    if (valid) {
        const int rowid = cc0 / W4;
        qc = cc0 - rowid * W4 + ph.q0[2];
        const int n = rowid / rows_per_n, ab = rowid - n * rows_per_n;
        qa = ab / ph.count[1];
        qb = ab - qa * ph.count[1] + ph.q0[1];
        qa += ph.q0[0];
        gbase = n * g.Cout * g.L + krow * g.L;
    }
    const int jstep = 8 * g.L;
    // Two halves: `gather` only ISSUES the four 16-byte loads of the next k-tile (before the MFMAs of the current one);
    // `gather_finish` picks / masks the elements and runs AFTER the MFMAs - touching the loaded registers any earlier
    // makes the wave wait for its loads with nothing to hide them behind.
    Stage<4> rb;
    int g_sh = 0, g_c = 0;
    bool g_ok = false;
    auto gather = [&](int kt) {
        const int chunk = kt / ntaps, tap = kt - chunk * ntaps, co0 = chunk * BK;  // taps inside a 32-channel chunk
        const int4 d = tapd[tap];
        const float* src = G + co0 * g.L;
        const int a = qa - d.x, b = qb - d.y, c = qc - d.z;  // output coordinates of the quad's first element
        g_ok = valid && a >= 0 && a < g.out[0] && b >= 0 && b < g.out[1];
        const int ac = min(max(a, 0), g.out[0] - 1), bc = min(max(b, 0), g.out[1] - 1);
        const int cs = min(max(c, 0), g.out[2] - 4);  // load start clamped into the row
        g_sh = c - cs;                                  // shift of element 0 inside the loaded vector
        g_c = c;
        const float* ptr = src + (gbase + (ac * g.out[1] + bc) * g.out[2] + cs);
#define NK_LDU(V, P) { const f32x4u q = *reinterpret_cast<const f32x4u*>(P); V = make_float4(q.x, q.y, q.z, q.w); }
        NK_LDU(rb.v0, ptr) NK_LDU(rb.v1, ptr + jstep) NK_LDU(rb.v2, ptr + 2 * jstep) NK_LDU(rb.v3, ptr + 3 * jstep)
#undef NK_LDU
    };
    auto gather_finish = [&]() {
        const int sh = g_sh;
        if (sh == 0) {  // interior quad (the common case): the loaded vector is the quad
            auto keep = [](float4& q, bool k) { q.x = k ? q.x : 0.f; q.y = k ? q.y : 0.f; q.z = k ? q.z : 0.f; q.w = k ? q.w : 0.f; };
            keep(rb.v0, g_ok); keep(rb.v1, g_ok); keep(rb.v2, g_ok); keep(rb.v3, g_ok);
        } else {
            const bool in0 = g_ok && g_c >= 0 && g_c < g.out[2], in1 = g_ok && g_c + 1 >= 0 && g_c + 1 < g.out[2],
                       in2 = g_ok && g_c + 2 >= 0 && g_c + 2 < g.out[2], in3 = g_ok && g_c + 3 >= 0 && g_c + 3 < g.out[2];
            // element i of the quad = element i + sh of the loaded vector: a barrel shifter of register selects (a pick by
            // dynamic index makes the compiler index the vector through scratch memory); elements shifted in from outside
            // the vector are always masked by in0..in3
            const int sl = max(sh, 0), sr = max(-sh, 0);
            const bool l1 = sl & 1, l2 = sl & 2, r1 = sr & 1, r2 = sr & 2;
            auto sel = [&](float4& q) {
                float e0 = q.x, e1 = q.y, e2 = q.z, e3 = q.w;
                e0 = l1 ? e1 : e0; e1 = l1 ? e2 : e1; e2 = l1 ? e3 : e2;
                e0 = l2 ? e2 : e0; e1 = l2 ? e3 : e1;
                e3 = r1 ? e2 : e3; e2 = r1 ? e1 : e2; e1 = r1 ? e0 : e1;
                e3 = r2 ? e1 : e3; e2 = r2 ? e0 : e2;
                q.x = in0 ? e0 : 0.f; q.y = in1 ? e1 : 0.f; q.z = in2 ? e2 : 0.f; q.w = in3 ? e3 : 0.f;
            };
            sel(rb.v0); sel(rb.v1); sel(rb.v2); sel(rb.v3);
        }
    };

    f32x16 acc[TI][TJ];
    acc_zero<TI, TJ>(acc);
    TileLoader<true, BM> la;
    la.init(Wq, K, m0, 0, g.Cg, nt * BK, t);
    Stage<BM / 32> ra;
    if (nt > 0) {  // a phase without taps (e.g. a 1x1 kernel with stride 2) only has zeros to write
        ra = la.template load<ALIGNED_A>(t);
        gather(0);
        gather_finish();
        stage_store<true, BM>(smem, ra, t);
        stage_store<false, BN>(smem + TA_FLOATS, rb, t);
    }
    __syncthreads();
    for (int it = 0; it + 1 < nt; ++it) {
        float* cur = smem + (it & 1) * STAGE;
        float* nxt = smem + ((it + 1) & 1) * STAGE;
        ra = la.template load<ALIGNED_A>(t);
        gather(it + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<true, false, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
        gather_finish();
        stage_store<true, BM>(nxt, ra, t);
        stage_store<false, BN>(nxt + TA_FLOATS, rb, t);
        __syncthreads();
    }
    if (nt > 0) {
        float* cur = smem + ((nt - 1) & 1) * STAGE;
        mma_tile<true, false, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
    }
    float* DX = p.dx;
    const int assign = g.assign;
    const int Cg = g.Cg, Cin = g.Cin, inplane = g.uinplane;
    long long cbase[TJ];  // one (n, i0, i1, i2) decode per owned column
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int cc = n0 + (wc * TJ + j) * 32 + (lane & 31);
        const int rowid = cc / W4, cpos = cc - rowid * W4;
        const int n = rowid / rows_per_n, ab = rowid - n * rows_per_n;
        const int i0 = ab / ph.count[1], i1 = ab - i0 * ph.count[1];
        cbase[j] = (cc < cols && cpos < ph.count[2])  // not a padding column of the row
                       ? ((long long)n * Cin + grp * Cg) * inplane +
                             ((long long)(ph.first[0] + i0 * g.stride[0]) * g.uin[1] + ph.first[1] + i1 * g.stride[1]) * g.uin[2] +
                             ph.first[2] + cpos * g.stride[2]
                       : -1;
    }
    NK_BWD_INPUT_EPILOGUE
}

// =================================================================================================
// Direct kernels for FEW channels per group (depthwise and small grouped convolutions).  With Mg or Cg of a handful an
// implicit GEMM fills 4 of the 64 rows of an MFMA tile (measured 2-4 TFLOP/s at 4 channels per group); the work per
// output element is only Cg * prod(k) multiply-adds, so the pass is HBM-bound and one thread per element with the taps in
// registers / L1 is the right shape.  Accumulation order: k = (ci, kernel idx) ascending, the reference's im2col order.
// =================================================================================================
constexpr int DIRECT_MAX_CH = 16;  // both Cin/g and Cout/g at most this many

// Block = 256 positions of ONE (sample, channel) plane, so the channel - and with it every weight address - is
// block-uniform: the weights come through scalar loads, the only vector loads are the activations.
// A thread owns PT positions 256 apart: the tap loops have run-time bounds and do not unroll, so one position per thread is
// one dependent load -> fma chain per iteration (latency-bound: 277 us for 12.8 M outputs x 36 taps); PT independent chains
// per iteration share the scalar weight load and keep PT activations in flight.
// y[n][co][l] = sum_{ci, tap} w[co][ci][tap] * x[n][grp*Cg + ci][origin(l) + tap]  (+ bias[co])
// TK1 x TK2: compile-time extents of the two innermost kernel axes (0 = run-time): the tap loops then unroll and a whole
// channel's TK1*TK2*PT activations are in flight at once instead of PT.
template <int PT, int TK1, int TK2>
__global__ __launch_bounds__(256) void conv_direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ y, ConvGeom g) {
    constexpr int UNROLL1 = TK1 ? TK1 : 1, UNROLL2 = TK2 ? TK2 : 1;  // full unroll for compile-time extents only
    const int K1 = TK1 ? TK1 : g.k[1], K2 = TK2 ? TK2 : g.k[2];
    const int nc = blockIdx.x, co = nc % g.Cout, n = nc / g.Cout, grp = co / g.Mg;
    const int l0 = blockIdx.y * (256 * PT) + threadIdx.x;
    const float* xp = x + ((long long)n * g.Cin + (long long)grp * g.Cg) * g.inplane;
    int org[PT];
    float acc[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int l = l0 + 256 * i;
        org[i] = window_origin(g, l < g.L ? l : 0);  // positions beyond the plane recompute position 0 and are not stored
        acc[i] = 0.f;
    }
    const float* ws = w + (long long)co * g.Cg * g.KK;
    for (int ci = 0; ci < g.Cg; ++ci) {
        const float* xc = xp + (long long)ci * g.inplane;
        const float* wc = ws + ci * g.KK;
        for (int k0 = 0; k0 < g.k[0]; ++k0)
#pragma unroll UNROLL1
            for (int k1 = 0; k1 < K1; ++k1) {
                const int roff = (k0 * g.dil[0] * g.in[1] + k1 * g.dil[1]) * g.in[2];
                const float* wr = wc + (k0 * K1 + k1) * K2;
#pragma unroll UNROLL2
                for (int k2 = 0; k2 < K2; ++k2) {
                    const float wv = wr[k2];
                    const int off = roff + k2 * g.dil[2];
#pragma unroll
                    for (int i = 0; i < PT; ++i) acc[i] = fmaf(wv, xc[org[i] + off], acc[i]);
                }
            }
    }
    const float bv = g.bias ? g.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int l = l0 + 256 * i;
        if (l < g.L) y[(long long)nc * g.L + l] = g.bias ? acc[i] + bv : acc[i];
    }
}

// Row-blocked forward for the common depthwise / small-group case (k[0] == 1, unit stride and dilation on the innermost
// axis, out[2] % 4 == 0, 16-byte aligned y): a thread computes FOUR adjacent outputs of a row, so a kernel row needs one
// 4 + TK2 - 1 element segment of the input row (one unaligned 16-byte load + TK2 - 1 scalars) instead of 4 * TK2 scalar
// loads, and the result goes out as one 16-byte store: a third of the load instructions and half the L1 bytes per output.
template <int TK1, int TK2>
__global__ __launch_bounds__(256) void conv_direct_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   float* __restrict__ y, ConvGeom g) {
    const int nc = blockIdx.x, co = nc % g.Cout, n = nc / g.Cout, grp = co / g.Mg;
    const int qpr = g.out[2] / 4;  // quads per output row
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= g.out[1] * qpr) return;
    const int oh = q / qpr, ow = (q - oh * qpr) * 4;
    const float* xp = x + ((long long)n * g.Cin + (long long)grp * g.Cg) * g.inplane + (oh * g.stride[1]) * g.in[2] + ow;
    const float* ws = w + (long long)co * g.Cg * (TK1 * TK2);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int ci = 0; ci < g.Cg; ++ci) {
        const float* xc = xp + (long long)ci * g.inplane;
        const float* wc = ws + ci * (TK1 * TK2);
#pragma unroll
        for (int k1 = 0; k1 < TK1; ++k1) {
            const float* xr = xc + k1 * g.dil[1] * g.in[2];
            float seg[4 + TK2 - 1];
            const f32x4u v = *reinterpret_cast<const f32x4u*>(xr);
            seg[0] = v.x; seg[1] = v.y; seg[2] = v.z; seg[3] = v.w;
#pragma unroll
            for (int j = 4; j < 4 + TK2 - 1; ++j) seg[j] = xr[j];
#pragma unroll
            for (int k2 = 0; k2 < TK2; ++k2) {  // same (ci, k1, k2) accumulation order as the one-output kernel
                const float wv = wc[k1 * TK2 + k2];
                a0 = fmaf(wv, seg[k2], a0); a1 = fmaf(wv, seg[k2 + 1], a1);
                a2 = fmaf(wv, seg[k2 + 2], a2); a3 = fmaf(wv, seg[k2 + 3], a3);
            }
        }
    }
    if (g.bias) { const float bv = g.bias[co]; a0 += bv; a1 += bv; a2 += bv; a3 += bv; }
    *reinterpret_cast<float4*>(y + (long long)nc * g.L + oh * g.out[2] + ow) = make_float4(a0, a1, a2, a3);
}

// dx[n][grp*Cg + ci][pos] (+)= sum_{co in group, tap} w[co][ci][tap] * gy[n][co][(pos + pad - tap*dil) / stride]
template <bool UNIT_STRIDE, int PT, int TK1, int TK2>
__global__ __launch_bounds__(256) void conv_direct_bwd_input_kernel(float* __restrict__ dx, const float* __restrict__ gy,
                                                                    const float* __restrict__ w, ConvGeom g) {
    constexpr int UNROLL1 = TK1 ? TK1 : 1, UNROLL2 = TK2 ? TK2 : 1;  // full unroll for compile-time extents only
    const int K1 = TK1 ? TK1 : g.k[1], K2 = TK2 ? TK2 : g.k[2];
    const int nc = blockIdx.x, cabs = nc % g.Cin, n = nc / g.Cin, grp = cabs / g.Cg, ci = cabs - grp * g.Cg;
    const int p0 = blockIdx.y * (256 * PT) + threadIdx.x;
    int pa[PT], pb[PT], pc[PT];
    float acc[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        int pos = p0 + 256 * i;
        pos = pos < g.uinplane ? pos : 0;
        pc[i] = pos % g.uin[2] + g.pad[2]; pos /= g.uin[2];
        pb[i] = pos % g.uin[1] + g.pad[1];
        pa[i] = pos / g.uin[1] + g.pad[0];
        acc[i] = 0.f;
    }
    const float* gs = gy + ((long long)n * g.Cout + (long long)grp * g.Mg) * g.L;
    for (int m = 0; m < g.Mg; ++m) {
        const float* gc = gs + (long long)m * g.L;
        const float* wc = w + ((long long)(grp * g.Mg + m) * g.Cg + ci) * g.KK;
        for (int k0 = 0; k0 < g.k[0]; ++k0) {
            int ra[PT];  // output coordinate on axis 0, or -1 when this tap row has none for the position
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                int a = pa[i] - k0 * g.dil[0];
                bool ok = a >= 0;
                if (!UNIT_STRIDE) { ok = ok && a % g.stride[0] == 0; a /= g.stride[0]; }
                ra[i] = ok && a < g.out[0] ? a : -1;
            }
#pragma unroll UNROLL1
            for (int k1 = 0; k1 < K1; ++k1) {
                const float* wr = wc + (k0 * K1 + k1) * K2;
                int rbase[PT];  // offset of the gradient row, or -1
#pragma unroll
                for (int i = 0; i < PT; ++i) {
                    int b = pb[i] - k1 * g.dil[1];
                    bool ok = ra[i] >= 0 && b >= 0;
                    if (!UNIT_STRIDE) { ok = ok && b % g.stride[1] == 0; b /= g.stride[1]; }
                    rbase[i] = ok && b < g.out[1] ? (ra[i] * g.out[1] + b) * g.out[2] : -1;
                }
#pragma unroll UNROLL2
                for (int k2 = 0; k2 < K2; ++k2) {
                    const float wv = wr[k2];
#pragma unroll
                    for (int i = 0; i < PT; ++i) {  // branch-free: a clamped (always valid) address, the product masked
                        int c = pc[i] - k2 * g.dil[2];
                        bool ok = rbase[i] >= 0 && c >= 0;
                        if (!UNIT_STRIDE) { ok = ok && c % g.stride[2] == 0; c /= g.stride[2]; }
                        ok = ok && c < g.out[2];
                        const float gv = gc[ok ? rbase[i] + c : 0];
                        acc[i] = fmaf(wv, ok ? gv : 0.f, acc[i]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int pos = p0 + 256 * i;
        if (pos < g.uinplane) {
            const long long o = (long long)nc * g.uinplane + pos;
            dx[o] = g.assign ? acc[i] : dx[o] + acc[i];
        }
    }
}

// Row-blocked backward-input, same idea (unit stride on every axis, unit dilation on the innermost one, k[0] == 1,
// uin[2] % 4 == 0, 16-byte aligned dx): four adjacent input positions share one 4 + TK2 - 1 element segment of each
// gradient row; positions outside the gradient read a clamped address and are masked.  Same (m, k1, k2) accumulation
// order per element as the one-position kernel.
template <int TK1, int TK2>
__global__ __launch_bounds__(256) void conv_direct_bwd_input_rows_kernel(float* __restrict__ dx, const float* __restrict__ gy,
                                                                         const float* __restrict__ w, ConvGeom g) {
    const int nc = blockIdx.x, cabs = nc % g.Cin, n = nc / g.Cin, grp = cabs / g.Cg, ci = cabs - grp * g.Cg;
    const int qpr = g.uin[2] / 4;
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= g.uin[1] * qpr) return;
    const int a = q / qpr, b = (q - a * qpr) * 4;
    const int pa = a + g.pad[1], c0 = b + g.pad[2] - (TK2 - 1);  // gradient column of segment element 0
    const float* gs = gy + ((long long)n * g.Cout + (long long)grp * g.Mg) * g.L;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    for (int m = 0; m < g.Mg; ++m) {
        const float* gc = gs + (long long)m * g.L;
        const float* wc = w + ((long long)(grp * g.Mg + m) * g.Cg + ci) * (TK1 * TK2);
#pragma unroll
        for (int k1 = 0; k1 < TK1; ++k1) {
            const int ra = pa - k1 * g.dil[1];
            const bool rowok = ra >= 0 && ra < g.out[1];
            const float* gr = gc + (rowok ? ra : 0) * g.out[2];
            float seg[4 + TK2 - 1];
#pragma unroll
            for (int j = 0; j < 4 + TK2 - 1; ++j) {
                const int col = c0 + j;
                const bool ok = rowok && col >= 0 && col < g.out[2];
                const float v = gr[ok ? col : 0];
                seg[j] = ok ? v : 0.f;
            }
#pragma unroll
            for (int k2 = 0; k2 < TK2; ++k2) {
                const float wv = wc[k1 * TK2 + k2];
                d0 = fmaf(wv, seg[TK2 - 1 - k2], d0); d1 = fmaf(wv, seg[TK2 - k2], d1);
                d2 = fmaf(wv, seg[TK2 + 1 - k2], d2); d3 = fmaf(wv, seg[TK2 + 2 - k2], d3);
            }
        }
    }
    float4* out = reinterpret_cast<float4*>(dx + (long long)nc * g.uinplane + a * g.uin[2] + b);
    if (g.assign) {
        *out = make_float4(d0, d1, d2, d3);
    } else {
        const float4 o = *out;
        *out = make_float4(o.x + d0, o.y + d1, o.z + d2, o.w + d3);
    }
}

// slab[split][co][ci][tap] = sum over the split's samples and all l of gy[n][co][l] * x[n][grp*Cg + ci][origin(l) + tap]:
// one block per (co, ci, tap); a thread keeps its output positions (one window decode each) and walks the samples;
// fixed-order block reduction, conv_dw_reduce_kernel sums the splits in order.
__global__ __launch_bounds__(256) void conv_direct_bwd_kernel_kernel(float* __restrict__ slabs, const float* __restrict__ gy,
                                                                     const float* __restrict__ x, ConvGeom g, int n_per_split) {
    __shared__ float red[256];
    const int e = blockIdx.x;  // (co*Cg + ci)*KK + tap
    const int tap = e % g.KK, cc = e / g.KK, ci = cc % g.Cg, co = cc / g.Cg, grp = co / g.Mg;
    int rem = tap;
    const int k2 = rem % g.k[2]; rem /= g.k[2];
    const int k1 = rem % g.k[1], k0 = rem / g.k[1];
    const int toff = (k0 * g.dil[0] * g.in[1] + k1 * g.dil[1]) * g.in[2] + k2 * g.dil[2];
    const int nbeg = blockIdx.y * n_per_split, nend = min(g.N, nbeg + n_per_split);
    const long long gstep = (long long)g.Cout * g.L, xstep = (long long)g.Cin * g.inplane;
    float acc = 0.f;
    for (int l = threadIdx.x; l < g.L; l += 256) {
        const float* gp = gy + ((long long)nbeg * g.Cout + co) * g.L + l;
        const float* xp = x + ((long long)nbeg * g.Cin + (long long)grp * g.Cg + ci) * g.inplane + window_origin(g, l) + toff;
#pragma unroll 8
        for (int n = nbeg; n < nend; ++n) {  // unrolled: eight independent load pairs in flight per trip
            acc = fmaf(*gp, *xp, acc);
            gp += gstep; xp += xstep;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) slabs[(long long)blockIdx.y * gridDim.x + e] = red[0];
}

// Same, all TK1*TK2 taps of one (co, ci) pair in one block (k[0] == 1): the gradient element is loaded once per (n, l) and
// the TK1*TK2 activations around it come from L1, instead of one block per tap re-reading both planes.
template <int TK1, int TK2>
__global__ __launch_bounds__(256) void conv_direct_bwd_kernel_taps_kernel(float* __restrict__ slabs, const float* __restrict__ gy,
                                                                          const float* __restrict__ x, ConvGeom g, int n_per_split) {
    constexpr int KK = TK1 * TK2;
    __shared__ float red[256];
    const int cc = blockIdx.x, ci = cc % g.Cg, co = cc / g.Cg, grp = co / g.Mg;
    const int nbeg = blockIdx.y * n_per_split, nend = min(g.N, nbeg + n_per_split);
    const long long gstep = (long long)g.Cout * g.L, xstep = (long long)g.Cin * g.inplane;
    float acc[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) acc[k] = 0.f;
    for (int l = threadIdx.x; l < g.L; l += 256) {
        const float* gp = gy + ((long long)nbeg * g.Cout + co) * g.L + l;
        const float* xp = x + ((long long)nbeg * g.Cin + (long long)grp * g.Cg + ci) * g.inplane + window_origin(g, l);
#pragma unroll 2
        for (int n = nbeg; n < nend; ++n) {
            const float gv = *gp;
#pragma unroll
            for (int k1 = 0; k1 < TK1; ++k1)
#pragma unroll
                for (int k2 = 0; k2 < TK2; ++k2)
                    acc[k1 * TK2 + k2] = fmaf(gv, xp[k1 * g.dil[1] * g.in[2] + k2 * g.dil[2]], acc[k1 * TK2 + k2]);
            gp += gstep; xp += xstep;
        }
    }
#pragma unroll
    for (int k = 0; k < KK; ++k) {  // fixed-order block reduction, one tap at a time
        red[threadIdx.x] = acc[k];
        __syncthreads();
        for (int sft = 128; sft > 0; sft >>= 1) {
            if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
            __syncthreads();
        }
        if (threadIdx.x == 0) slabs[((long long)blockIdx.y * gridDim.x + cc) * KK + k] = red[0];
        __syncthreads();
    }
}

bool use_direct(const ConvGeom& g) {
    return g.Cg <= DIRECT_MAX_CH && g.Mg <= DIRECT_MAX_CH && (long long)g.N * g.Cout < 0x7fffffffLL &&
           (long long)g.N * g.Cin < 0x7fffffffLL && (long long)g.Cout * g.Cg * g.KK < 0x7fffffffLL &&
           g.L / 256 < 65535 && g.uinplane / 256 < 65535;  // grid.y carries the position blocks
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
    for (int d = 0; d < 3; ++d) { g.uin[d] = g.in[d]; g.pad[d] = 0; }
    g.uinplane = g.inplane;
    *out = g;
    return NK_OK;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
size_t round256(size_t b) { return (b + 255) & ~size_t(255); }

int conv_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, const float* w, const int* w_shape,
             const float* bias, float* y, const int* stride, const int* dilation, int groups) {
    NK_USE(dev);
    ConvGeom g;
    int rc = make_geom(nd, x_shape, w_shape, stride, dilation, groups, &g);
    if (rc) return rc;
    g.bias = bias;
    if ((long long)g.N * g.Cout * g.L == 0) return NK_OK;
    NK_CHECK(x && w && y, "null pointer in nk_conv_fwd");
    if (use_direct(g)) {
        rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
        if (rc) return rc;
        const bool rows_ok = g.k[0] == 1 && g.out[0] == 1 && g.stride[2] == 1 && g.dil[2] == 1 && g.out[2] % 4 == 0 && al16(y) &&
                             ((g.k[1] == 3 && g.k[2] == 3) || (g.k[1] == 5 && g.k[2] == 5) || (g.k[1] == 1 && g.k[2] == 3));
        if (rows_ok) {
            const dim3 rgrid((unsigned)(g.N * g.Cout), (unsigned)((g.out[1] * (g.out[2] / 4) + 255) / 256));
            if (g.k[1] == 3) hipLaunchKernelGGL((conv_direct_fwd_rows_kernel<3, 3>), rgrid, dim3(256), 0, dev->compute, x, w, y, g);
            else if (g.k[1] == 5) hipLaunchKernelGGL((conv_direct_fwd_rows_kernel<5, 5>), rgrid, dim3(256), 0, dev->compute, x, w, y, g);
            else hipLaunchKernelGGL((conv_direct_fwd_rows_kernel<1, 3>), rgrid, dim3(256), 0, dev->compute, x, w, y, g);
            NK_LAUNCH_CHECK();
            return nk_prof_stop(dev);
        }
        const int pt = g.L >= 512 ? 4 : 1;
        const dim3 dgrid((unsigned)(g.N * g.Cout), (unsigned)((g.L + 256 * pt - 1) / (256 * pt)));
#define NK_DF(PT_, A, B) hipLaunchKernelGGL((conv_direct_fwd_kernel<PT_, A, B>), dgrid, dim3(256), 0, dev->compute, x, w, y, g)
        if (g.k[1] == 3 && g.k[2] == 3) { if (pt == 4) NK_DF(4, 3, 3); else NK_DF(1, 3, 3); }
        else if (g.k[1] == 5 && g.k[2] == 5) { if (pt == 4) NK_DF(4, 5, 5); else NK_DF(1, 5, 5); }
        else { if (pt == 4) NK_DF(4, 0, 0); else NK_DF(1, 0, 0); }
#undef NK_DF
        NK_LAUNCH_CHECK();
        return nk_prof_stop(dev);
    }
    const int K = g.Cg * g.KK;
    const long long x_elems = (long long)g.N * g.Cin * g.inplane, y_elems = (long long)g.N * g.Cout * g.L;
    const long long fcols = (long long)g.N * g.out[0] * g.out[1] * ((g.out[2] + 3) & ~3);  // row-padded column space
    if (g.Cg % BK == 0 && g.stride[2] == 1 && g.out[2] >= 4 && x_elems < 0x7fffffffLL && y_elems < 0x7fffffffLL && fcols < 0x7fffff00LL) {
        const size_t wp_bytes = round256((size_t)g.Cout * K * sizeof(float));
        const size_t to_bytes = round256((size_t)g.KK * sizeof(int));
        const size_t tables = wp_bytes + to_bytes + round256((size_t)g.KK * sizeof(int4));
        FastFwdArgs fp{};
        const int fti = g.Mg <= 64 || (g.Mg % 128 != 0 && g.Mg % 64 == 0) ? 1 : 2;
        fp.tiles_m = (g.Mg + 64 * fti - 1) / (64 * fti);
        fp.tiles_n = (int)((fcols + 127) / 128);
        const bool al = g.Mg % (64 * fti) == 0;
        // tail balancing: tiles beyond the last whole wave of resident blocks (2 per CU) are split along k
        const int tiles = fp.tiles_m * fp.tiles_n, slots = 2 * dev->num_cus, nkt = K / BK;
        int nblocks = tiles;
        size_t slab_bytes = 0;
        fp.full_blocks = tiles;
#ifndef NK_AB_NO_TAIL
        const int tail = tiles % slots;
        if (groups == 1 && tiles > slots && tail > 0 && tail * 2 <= slots && nkt >= 4) {
            int S = slots / tail;
            if (S > nkt / 2) S = nkt / 2;
            const int kts = (nkt + S - 1) / S;
            S = (nkt + kts - 1) / kts;
            if (S >= 2) {
                fp.full_blocks = tiles - tail;
                fp.tail_splits = S;
                fp.tail_kts = kts;
                nblocks = fp.full_blocks + tail * S;
                slab_bytes = (size_t)tail * S * (64 * fti) * 128 * sizeof(float);
            }
        }
#endif
        void* wsf = nullptr;
        rc = nk_workspace(dev, tables + slab_bytes, &wsf);
        if (rc) return rc;
        float* wp = (float*)wsf;
        int* tapoff = (int*)((char*)wsf + wp_bytes);
        int4* tapd = (int4*)((char*)wsf + wp_bytes + to_bytes);
        if (slab_bytes) fp.slabs = (float*)((char*)wsf + tables);
        hipLaunchKernelGGL(conv_wp_kernel, dim3(nk_stream_grid((size_t)g.Cout * K, 256)), dim3(256), 0, dev->compute, wp, w, g);
        NK_LAUNCH_CHECK();
        hipLaunchKernelGGL(conv_tapoff_kernel, dim3(1), dim3(64), 0, dev->compute, tapoff, tapd, g);
        NK_LAUNCH_CHECK();
        fp.g = g; fp.x = x; fp.wp = wp; fp.y = y; fp.tapoff = tapoff;
        dim3 fgrid(nblocks, 1, groups);
        rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
        if (rc) return rc;
#define NK_LAUNCH_FF(AL, TI_, RP_) hipLaunchKernelGGL((conv_fwd_fast_kernel<AL, TI_, RP_>), fgrid, dim3(NT), 0, dev->compute, fp)
        if (g.out[2] % 4 == 0) {
            if (al && fti == 2) NK_LAUNCH_FF(true, 2, false);
            else if (al) NK_LAUNCH_FF(true, 1, false);
            else if (fti == 2) NK_LAUNCH_FF(false, 2, false);
            else NK_LAUNCH_FF(false, 1, false);
        } else {
            if (al && fti == 2) NK_LAUNCH_FF(true, 2, true);
            else if (al) NK_LAUNCH_FF(true, 1, true);
            else if (fti == 2) NK_LAUNCH_FF(false, 2, true);
            else NK_LAUNCH_FF(false, 1, true);
        }
#undef NK_LAUNCH_FF
        NK_LAUNCH_CHECK();
        if (fp.tail_splits) {
            const dim3 rgrid(tiles - fp.full_blocks, (64 * fti * 128) / 1024, groups);
            if (fti == 2) hipLaunchKernelGGL(conv_fwd_tail_reduce_kernel<128>, rgrid, dim3(256), 0, dev->compute, fp);
            else hipLaunchKernelGGL(conv_fwd_tail_reduce_kernel<64>, rgrid, dim3(256), 0, dev->compute, fp);
            NK_LAUNCH_CHECK();
        }
        return nk_prof_stop(dev);
    }
    void* ws = nullptr;
    rc = nk_workspace(dev, round256((size_t)K * sizeof(int)), &ws);
    if (rc) return rc;
    hipLaunchKernelGGL(conv_koff_kernel, dim3((K + 255) / 256), dim3(256), 0, dev->compute, (int*)ws, g);
    NK_LAUNCH_CHECK();
    FwdArgs p{};
    p.g = g; p.x = x; p.w = w; p.y = y; p.koff = (const int*)ws;
    const int ti = g.Mg <= 64 || (g.Mg % 128 != 0 && g.Mg % 64 == 0) ? 1 : 2;
    const int BM = 64 * ti, BN = 128;
    p.tiles_m = (g.Mg + BM - 1) / BM;
    const long long cols = (long long)g.N * g.L;
    p.tiles_n = (int)((cols + BN - 1) / BN);
    const bool aligned_a = (g.Mg % BM == 0) && (K % BK == 0) && al16(w);
    dim3 grid(p.tiles_m * p.tiles_n, 1, groups);
    rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
    if (rc) return rc;
#define NK_LAUNCH_FG(AL, TI_, QV) hipLaunchKernelGGL((conv_fwd_kernel<AL, TI_, QV>), grid, dim3(NT), 0, dev->compute, p)
    if (g.stride[2] == 1 && g.out[2] % 4 == 0) {  // every staged column quad lies in one output row
        if (aligned_a && ti == 2) NK_LAUNCH_FG(true, 2, true);
        else if (aligned_a) NK_LAUNCH_FG(true, 1, true);
        else if (ti == 2) NK_LAUNCH_FG(false, 2, true);
        else NK_LAUNCH_FG(false, 1, true);
    } else {
        if (aligned_a && ti == 2) NK_LAUNCH_FG(true, 2, false);
        else if (aligned_a) NK_LAUNCH_FG(true, 1, false);
        else if (ti == 2) NK_LAUNCH_FG(false, 2, false);
        else NK_LAUNCH_FG(false, 1, false);
    }
#undef NK_LAUNCH_FG
    NK_LAUNCH_CHECK();
    return nk_prof_stop(dev);
}

// `padding` (may be null): x_shape is the input of a zero Pad node whose output the convolution read; dX gets the centre
// block of the padded input's gradient (PadBackward folded into the gather, the padded gradient is never stored).
int conv_bwd_input(nk_device* dev, int nd, float* dx, const int* x_shape, const int* padding, const float* gy, const float* w,
                   const int* w_shape, const int* stride, const int* dilation, int groups, int assign) {
    NK_USE(dev);
    NK_CHECK(nd >= 1 && nd <= 3, "Invalid convolution dimension %d (1, 2 or 3 supported)", nd);
    ConvGeom g;
    int pshape[5] = {x_shape[0], x_shape[1], 1, 1, 1};
    for (int d = 0; d < nd; ++d) {
        NK_CHECK(!padding || padding[d] >= 0, "negative padding on axis %d", d);
        pshape[2 + d] = x_shape[2 + d] + (padding ? 2 * padding[d] : 0);
    }
    int rc = make_geom(nd, pshape, w_shape, stride, dilation, groups, &g);
    if (rc) return rc;
    g.assign = assign;
    if (padding) {
        g.uinplane = 1;
        for (int d = 0; d < nd; ++d) {
            const int q = 3 - nd + d;
            g.pad[q] = padding[d];
            g.uin[q] = x_shape[2 + d];
            g.uinplane *= g.uin[q];
        }
    }
    if ((long long)g.N * g.Cin * g.uinplane == 0) return NK_OK;
    if ((long long)g.Cout * g.L == 0) {  // no output positions: the gradient is zero
        if (assign) NK_HIP(hipMemsetAsync(dx, 0, (size_t)g.N * g.Cin * g.uinplane * sizeof(float), dev->compute));
        return NK_OK;
    }
    NK_CHECK(dx && gy && w, "null pointer in nk_conv_bwd_input");
    if (use_direct(g)) {
        rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
        if (rc) return rc;
        const bool unit = g.stride[0] == 1 && g.stride[1] == 1 && g.stride[2] == 1;
        const bool rows_ok = unit && g.k[0] == 1 && g.uin[0] == 1 && g.pad[0] == 0 && g.dil[2] == 1 && g.uin[2] % 4 == 0 && al16(dx) &&
                             ((g.k[1] == 3 && g.k[2] == 3) || (g.k[1] == 5 && g.k[2] == 5) || (g.k[1] == 1 && g.k[2] == 3));
        if (rows_ok) {
            const dim3 rgrid((unsigned)(g.N * g.Cin), (unsigned)((g.uin[1] * (g.uin[2] / 4) + 255) / 256));
            if (g.k[1] == 3) hipLaunchKernelGGL((conv_direct_bwd_input_rows_kernel<3, 3>), rgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            else if (g.k[1] == 5) hipLaunchKernelGGL((conv_direct_bwd_input_rows_kernel<5, 5>), rgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            else hipLaunchKernelGGL((conv_direct_bwd_input_rows_kernel<1, 3>), rgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            NK_LAUNCH_CHECK();
            return nk_prof_stop(dev);
        }
        const int pt = g.uinplane >= 512 ? 4 : 1;
        const dim3 dgrid((unsigned)(g.N * g.Cin), (unsigned)((g.uinplane + 256 * pt - 1) / (256 * pt)));
#define NK_DI(U, PT_, A, B) hipLaunchKernelGGL((conv_direct_bwd_input_kernel<U, PT_, A, B>), dgrid, dim3(256), 0, dev->compute, dx, gy, w, g)
        if (g.k[1] == 3 && g.k[2] == 3) {
            if (unit) { if (pt == 4) NK_DI(true, 4, 3, 3); else NK_DI(true, 1, 3, 3); }
            else { if (pt == 4) NK_DI(false, 4, 3, 3); else NK_DI(false, 1, 3, 3); }
        } else {
            if (unit) { if (pt == 4) NK_DI(true, 4, 0, 0); else NK_DI(true, 1, 0, 0); }
            else { if (pt == 4) NK_DI(false, 4, 0, 0); else NK_DI(false, 1, 0, 0); }
        }
#undef NK_DI
        NK_LAUNCH_CHECK();
        return nk_prof_stop(dev);
    }
    const int K = g.Mg * g.KK;
    {
        const long long x_elems = (long long)g.N * g.Cin * g.inplane, y_elems = (long long)g.N * g.Cout * g.L;
        // stride phases (see BwdInPhase): residue classes of the padded input coordinate, each with its taps and columns
        const int nphase = g.stride[0] * g.stride[1] * g.stride[2];
        BwdInPhaseTable tbl{};
        long long tiles_n = 0, max_cols = 0;
        bool fits = nphase <= MAX_PHASES && g.Mg % BK == 0 && g.out[2] >= 4 && x_elems < 0x7fffffffLL && y_elems < 0x7fffffffLL;
        if (fits) {
            int tap_begin = 0;
            for (int pid = 0; pid < nphase; ++pid) {
                BwdInPhase& ph = tbl.ph[pid];
                const int r[3] = {pid / (g.stride[1] * g.stride[2]), (pid / g.stride[2]) % g.stride[1], pid % g.stride[2]};
                long long cols = g.N;
                for (int d = 0; d < 3; ++d) {
                    const int sd = g.stride[d];
                    ph.first[d] = ((r[d] - g.pad[d]) % sd + sd) % sd;
                    ph.count[d] = ph.first[d] < g.uin[d] ? (g.uin[d] - ph.first[d] + sd - 1) / sd : 0;
                    ph.q0[d] = (ph.first[d] + g.pad[d] - r[d]) / sd;
                    cols *= d == 2 ? (ph.count[d] + 3) & ~3 : ph.count[d];
                }
                int ntaps = 0;
                for (int k0 = 0; k0 < g.k[0]; ++k0)
                    for (int k1 = 0; k1 < g.k[1]; ++k1)
                        for (int k2 = 0; k2 < g.k[2]; ++k2)
                            ntaps += (k0 * g.dil[0]) % g.stride[0] == r[0] && (k1 * g.dil[1]) % g.stride[1] == r[1] &&
                                     (k2 * g.dil[2]) % g.stride[2] == r[2];
                ph.tap_begin = tap_begin; ph.ntaps = ntaps;
                tap_begin += ntaps;
                ph.tile_begin = (int)tiles_n;
                tiles_n += (cols + 127) / 128;
                if (cols > max_cols) max_cols = cols;
            }
            for (int pid = nphase; pid < MAX_PHASES; ++pid) tbl.ph[pid].tile_begin = 0x7fffffff;
            fits = max_cols < 0x7fffff00LL && tiles_n < 0x7fffffffLL;
        }
        if (fits) {
            const size_t wq_bytes = round256((size_t)g.Cout * g.Cg * g.KK * sizeof(float));
            const size_t td_bytes = round256((size_t)g.KK * sizeof(int4));
            const size_t ph_bytes = round256(sizeof(BwdInPhaseTable));
            void* wsf = nullptr;
            rc = nk_workspace(dev, wq_bytes + 2 * td_bytes + ph_bytes, &wsf);
            if (rc) return rc;
            float* wq = (float*)wsf;
            int4* tapd = (int4*)((char*)wsf + wq_bytes);
            int4* tappos = (int4*)((char*)wsf + wq_bytes + td_bytes);
            BwdInPhase* phases = (BwdInPhase*)((char*)wsf + wq_bytes + 2 * td_bytes);
            hipLaunchKernelGGL(conv_phase_taps_kernel, dim3((g.KK + 63) / 64), dim3(64), 0, dev->compute, tapd, tappos, g);
            NK_LAUNCH_CHECK();
            hipLaunchKernelGGL(conv_phase_table_kernel, dim3(1), dim3(1), 0, dev->compute, phases, tbl);
            NK_LAUNCH_CHECK();
            hipLaunchKernelGGL(conv_wq_kernel, dim3(nk_stream_grid((size_t)g.Cout * g.Cg * g.KK, 256)), dim3(256), 0, dev->compute, wq, w, tappos, g);
            NK_LAUNCH_CHECK();
            FastBwdInArgs fp{};
            fp.g = g; fp.dx = dx; fp.gy = gy; fp.wq = wq; fp.tapd = tapd; fp.phases = phases; fp.nphase = nphase;
            const int fti = g.Cg <= 64 || (g.Cg % 128 != 0 && g.Cg % 64 == 0) ? 1 : 2;
            fp.tiles_m = (g.Cg + 64 * fti - 1) / (64 * fti);
            fp.tiles_n = (int)tiles_n;
            const bool al = g.Cg % (64 * fti) == 0;
            dim3 fgrid(fp.tiles_m * fp.tiles_n, 1, groups);
            rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
            if (rc) return rc;
            if (al && fti == 2) hipLaunchKernelGGL((conv_bwd_input_fast_kernel<true, 2>), fgrid, dim3(NT), 0, dev->compute, fp);
            else if (al) hipLaunchKernelGGL((conv_bwd_input_fast_kernel<true, 1>), fgrid, dim3(NT), 0, dev->compute, fp);
            else if (fti == 2) hipLaunchKernelGGL((conv_bwd_input_fast_kernel<false, 2>), fgrid, dim3(NT), 0, dev->compute, fp);
            else hipLaunchKernelGGL((conv_bwd_input_fast_kernel<false, 1>), fgrid, dim3(NT), 0, dev->compute, fp);
            NK_LAUNCH_CHECK();
            return nk_prof_stop(dev);
        }
    }
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
    const int ti = g.Cg <= 64 || (g.Cg % 128 != 0 && g.Cg % 64 == 0) ? 1 : 2;
    const int BM = 64 * ti, BN = 128;
    p.tiles_m = (g.Cg + BM - 1) / BM;
    const long long cols = (long long)g.N * g.uinplane;
    p.tiles_n = (int)((cols + BN - 1) / BN);
    const bool aligned_a = (g.Cg % BM == 0) && (K % BK == 0);
    const bool unit = g.stride[0] == 1 && g.stride[1] == 1 && g.stride[2] == 1;
    dim3 grid(p.tiles_m * p.tiles_n, 1, groups);
    rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
    if (rc) return rc;
#define NK_LAUNCH_BWI(AL, UN, TI_) hipLaunchKernelGGL((conv_bwd_input_kernel<AL, UN, TI_>), grid, dim3(NT), 0, dev->compute, p)
    if (ti == 2) {
        if (aligned_a && unit) NK_LAUNCH_BWI(true, true, 2);
        else if (aligned_a) NK_LAUNCH_BWI(true, false, 2);
        else if (unit) NK_LAUNCH_BWI(false, true, 2);
        else NK_LAUNCH_BWI(false, false, 2);
    } else {
        if (aligned_a && unit) NK_LAUNCH_BWI(true, true, 1);
        else if (aligned_a) NK_LAUNCH_BWI(true, false, 1);
        else if (unit) NK_LAUNCH_BWI(false, true, 1);
        else NK_LAUNCH_BWI(false, false, 1);
    }
#undef NK_LAUNCH_BWI
    NK_LAUNCH_CHECK();
    return nk_prof_stop(dev);
}

int conv_bwd_kernel(nk_device* dev, int nd, float* dw, const int* w_shape, const float* gy, const float* x,
                    const int* x_shape, const int* stride, const int* dilation, int groups, int assign) {
    NK_USE(dev);
    ConvGeom g;
    int rc = make_geom(nd, x_shape, w_shape, stride, dilation, groups, &g);
    if (rc) return rc;
    const bool quadr = g.stride[2] == 1 && g.out[2] >= 4;  // row-padded quad staging (see the kernel)
    const long long R = quadr ? (long long)g.N * g.out[0] * g.out[1] * ((g.out[2] + 3) & ~3) : (long long)g.N * g.L;
    const int Kc = g.Cg * g.KK;
    if ((long long)g.Cout * Kc == 0) return NK_OK;
    NK_CHECK(dw && gy && x, "null pointer in nk_conv_bwd_kernel");
    if (R == 0) {  // empty batch: the gradient is zero
        if (assign) NK_HIP(hipMemsetAsync(dw, 0, (size_t)g.Cout * Kc * sizeof(float), dev->compute));
        return NK_OK;
    }
    if (use_direct(g)) {  // few channels per group: one block per (co, ci, tap) dot product, split over (n, l)
        const long long dw_n = (long long)g.Cout * Kc;
        const bool taps_in_regs = g.k[0] == 1 && ((g.k[1] == 3 && g.k[2] == 3) || (g.k[1] == 5 && g.k[2] == 5));
        const long long dblocks = taps_in_regs ? (long long)g.Cout * g.Cg : dw_n;
        long long dsplits = (4096 + dblocks - 1) / dblocks;  // ~4096 blocks, split over the samples
        if (dsplits > g.N) dsplits = g.N;
        if (dsplits < 1) dsplits = 1;
        const int rps = (int)((g.N + dsplits - 1) / dsplits);  // samples per split
        dsplits = (g.N + rps - 1) / rps;
        void* wsd = nullptr;
        rc = nk_workspace(dev, (size_t)dsplits * dw_n * sizeof(float), &wsd);
        if (rc) return rc;
        rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
        if (rc) return rc;
        const dim3 tgrid((unsigned)(g.Cout * g.Cg), (unsigned)dsplits);  // all taps of a (co, ci) pair per block
        if (g.k[0] == 1 && g.k[1] == 3 && g.k[2] == 3)
            hipLaunchKernelGGL((conv_direct_bwd_kernel_taps_kernel<3, 3>), tgrid, dim3(256), 0, dev->compute, (float*)wsd, gy, x, g, rps);
        else if (g.k[0] == 1 && g.k[1] == 5 && g.k[2] == 5)
            hipLaunchKernelGGL((conv_direct_bwd_kernel_taps_kernel<5, 5>), tgrid, dim3(256), 0, dev->compute, (float*)wsd, gy, x, g, rps);
        else
            hipLaunchKernelGGL(conv_direct_bwd_kernel_kernel, dim3((unsigned)dw_n, (unsigned)dsplits), dim3(256), 0, dev->compute,
                               (float*)wsd, gy, x, g, rps);
        NK_LAUNCH_CHECK();
        hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3((unsigned)((dw_n + 63) / 64)), dim3(256), 0, dev->compute, dw, (const float*)wsd,
                           dw_n, (int)dsplits, assign);
        NK_LAUNCH_CHECK();
        return nk_prof_stop(dev);
    }
    BwdKArgs p{};
    p.g = g; p.gy = gy; p.x = x;
    const int ti = g.Mg <= 64 || (g.Mg % 128 != 0 && g.Mg % 64 == 0) ? 1 : 2;
    const int tj = Kc <= 64 ? 1 : 2;
    const int BM = 64 * ti, BN = 64 * tj;
    p.tiles_m = (g.Mg + BM - 1) / BM;
    p.tiles_n = (Kc + BN - 1) / BN;
    const long long tiles = (long long)p.tiles_m * p.tiles_n * groups;
    const long long rtiles = (R + BK - 1) / BK;
    // whole "waves" of blocks: the kernel keeps 2 blocks per CU resident, so the grid is sized to
    // (a multiple of) 2 * CUs blocks — a ragged second wave would idle most of the chip
    const long long slots = 2LL * dev->num_cus;
    long long waves = (tiles * ((rtiles + 127) / 128) + slots - 1) / slots;  // <= ~128 k-tiles per block ...
    if (waves < 1) waves = 1;
    long long splits = slots * waves / tiles;             // ... in full waves
    if (splits < 1) splits = 1;
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
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n * splits), 1, groups);
    const bool vec_g = (g.L % 4 == 0) && al16(gy);
    rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
    if (rc) return rc;
#define NK_LAUNCH_BWK(VG, TI_, TJ_, Q) hipLaunchKernelGGL((conv_bwd_kernel_kernel<VG, TI_, TJ_, Q>), grid, dim3(NT), 0, dev->compute, p)
    if (quadr) {
        if (ti == 2 && tj == 2) NK_LAUNCH_BWK(true, 2, 2, true);
        else if (ti == 2) NK_LAUNCH_BWK(true, 2, 1, true);
        else if (tj == 2) NK_LAUNCH_BWK(true, 1, 2, true);
        else NK_LAUNCH_BWK(true, 1, 1, true);
    } else if (vec_g) {
        if (ti == 2 && tj == 2) NK_LAUNCH_BWK(true, 2, 2, false);
        else if (ti == 2) NK_LAUNCH_BWK(true, 2, 1, false);
        else if (tj == 2) NK_LAUNCH_BWK(true, 1, 2, false);
        else NK_LAUNCH_BWK(true, 1, 1, false);
    } else {
        if (ti == 2 && tj == 2) NK_LAUNCH_BWK(false, 2, 2, false);
        else if (ti == 2) NK_LAUNCH_BWK(false, 2, 1, false);
        else if (tj == 2) NK_LAUNCH_BWK(false, 1, 2, false);
        else NK_LAUNCH_BWK(false, 1, 1, false);
    }
#undef NK_LAUNCH_BWK
    NK_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3((unsigned)((dw_elems + 63) / 64)), dim3(256), 0, dev->compute, dw,
                       p.slabs, dw_elems, (int)splits, assign);
    NK_LAUNCH_CHECK();
    return nk_prof_stop(dev);
}

}  // namespace

extern "C" {

int nk_conv_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, const float* w, const int* w_shape,
                float* y, const int* stride, const int* dilation, int groups) {
    return conv_fwd(dev, nd, x, x_shape, w, w_shape, nullptr, y, stride, dilation, groups);
}
int nk_conv_bias_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, const float* w, const int* w_shape,
                     const float* bias, float* y, const int* stride, const int* dilation, int groups) {
    NK_CHECK(bias != nullptr, "null bias");
    return conv_fwd(dev, nd, x, x_shape, w, w_shape, bias, y, stride, dilation, groups);
}
int nk_conv_bwd_input(nk_device* dev, int nd, float* dx, const int* x_shape, const float* gy, const float* w,
                      const int* w_shape, const int* stride, const int* dilation, int groups) {
    return conv_bwd_input(dev, nd, dx, x_shape, nullptr, gy, w, w_shape, stride, dilation, groups, 0);
}
int nk_conv_bwd_input_assign(nk_device* dev, int nd, float* dx, const int* x_shape, const float* gy, const float* w,
                             const int* w_shape, const int* stride, const int* dilation, int groups) {
    return conv_bwd_input(dev, nd, dx, x_shape, nullptr, gy, w, w_shape, stride, dilation, groups, 1);
}
int nk_conv_bwd_input_padded(nk_device* dev, int nd, float* dx, const int* x_shape, const int* padding, const float* gy,
                             const float* w, const int* w_shape, const int* stride, const int* dilation, int groups) {
    NK_CHECK(padding != nullptr, "null padding");
    return conv_bwd_input(dev, nd, dx, x_shape, padding, gy, w, w_shape, stride, dilation, groups, 0);
}
int nk_conv_bwd_input_padded_assign(nk_device* dev, int nd, float* dx, const int* x_shape, const int* padding, const float* gy,
                                    const float* w, const int* w_shape, const int* stride, const int* dilation, int groups) {
    NK_CHECK(padding != nullptr, "null padding");
    return conv_bwd_input(dev, nd, dx, x_shape, padding, gy, w, w_shape, stride, dilation, groups, 1);
}
int nk_conv_bwd_kernel(nk_device* dev, int nd, float* dw, const int* w_shape, const float* gy, const float* x,
                       const int* x_shape, const int* stride, const int* dilation, int groups) {
    return conv_bwd_kernel(dev, nd, dw, w_shape, gy, x, x_shape, stride, dilation, groups, 0);
}
int nk_conv_bwd_kernel_assign(nk_device* dev, int nd, float* dw, const int* w_shape, const float* gy, const float* x,
                              const int* x_shape, const int* stride, const int* dilation, int groups) {
    return conv_bwd_kernel(dev, nd, dw, w_shape, gy, x, x_shape, stride, dilation, groups, 1);
}

}  // extern "C"
