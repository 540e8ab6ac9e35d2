// Generic implicit-GEMM convolution kernels (any channel count) and the kernel-gradient pass  -  part of the convolution translation unit (included by nk_conv.hip inside its anonymous
// namespace; not a stand-alone header).
#pragma once

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
    float* slabs;        // [splits][groups][Mg][Cg*KK]
    float* bias_slabs;   // optional [splits][groups][Mg]: per-split sums of G over the reduction range (the conv module's bias gradient)
    int tiles_m, tiles_n;
    long long r_per_split;  // multiple of BK
    // conv_bwd_kernel_mixed_kernel: the last column tile is 64 wide and split `narrow_splits` (< the others' count) ways
    int narrow_splits;
    long long r_per_split_narrow;
};

// QUADR (unit stride on the innermost axis, out[2] >= 4): the reduction runs over (n, o0, o1, c') with the innermost output
// row padded to W4 = a multiple of 4, so the four consecutive reduction indices a thread stages are one output-row quad:
// one (incremental, division-free) decode per k-tile and one 16-byte load per staged row.  A quad that would run past the
// row end (out[2] % 4 != 0) is loaded `dup` elements earlier - for BOTH operands, a reduction does not care where in the
// k-tile an element sits - and its first `dup` elements, already counted by the previous quad, are masked.
// SW (QUADR only): stride on the innermost axis, 1 or 2 - with 2 the X quad of four consecutive output positions is input positions
// 0, 2, 4, 6 from its origin: two unaligned 16-byte loads per staged row (at +0 and +3, see conv_fwd_fast_kernel), G is unit-stride.
template <bool VEC_G, int TI, int TJ, bool QUADR, int SW = 1>
__global__ __launch_bounds__(NT, 2) void conv_bwd_kernel_kernel(BwdKArgs p) {
    // 1-D grid over (split, tile): the tiles of one split read the same G / X slices, so they are
    // made neighbours in the per-XCD chunk order (shared through that XCD's L2)
#define NK_BWK_DECODE                                                                                       \
    const int ntile = p.tiles_m * p.tiles_n;                                                                \
    int split, tile;                                                                                        \
    tile_coords(blockIdx.x, gridDim.x, 1, (int)gridDim.x, tile, split); /* split := XCD-chunked linear id */ \
    tile = split % ntile;                                                                                   \
    split /= ntile;                                                                                         \
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
#define NK_BWK_SMEM __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
#define NK_BWK_RPS p.r_per_split
#include "nk_conv_bwdk_body.h"
#undef NK_BWK_DECODE
#undef NK_BWK_SMEM
#undef NK_BWK_RPS
}

// Column counts that end in HALF a tile (Cin/g * taps = 128 q + 64: 3 x 3 on 64 channels is 576 = 4.5 tiles) - one launch in which
// the whole column tiles run the 128-wide body and the last one the 64-wide body (wave tiles 64 x 32: half the MFMAs and half the
// X gathers per k-tile, none of them on padding columns), with the reduction split per tile SHAPE so that every block takes
// about the same time: the narrow tile is cut into fewer, longer ranges (`narrow_splits` < `splits`, `r_per_split_narrow`).
// Blocks [0, narrow_splits * tiles): split-major over all tiles; the rest: split-major over the wide tiles only.
// TJ = 2: a whole column tile; TJ = 1: the narrow one = column tile 2 * (tiles_n - 1) of the 64-wide tiling (n0 = tn * 64)
template <bool VEC_G, int TI, int TJ, bool QUADR, int SW>
__device__ __forceinline__ void conv_bwd_kernel_mixed_body(const BwdKArgs& p, float* smem_shared, int tile, int split_) {
#define NK_BWK_DECODE   \
    int split = split_; \
    const int tm = tile / p.tiles_n, tn = TJ == 1 ? 2 * (p.tiles_n - 1) : tile % p.tiles_n;
#define NK_BWK_SMEM float* const smem = smem_shared;
#define NK_BWK_RPS (TJ == 1 ? p.r_per_split_narrow : p.r_per_split)
#include "nk_conv_bwdk_body.h"
#undef NK_BWK_DECODE
#undef NK_BWK_SMEM
#undef NK_BWK_RPS
}
template <bool VEC_G, int TI, bool QUADR, int SW>
__global__ __launch_bounds__(NT, 2) void conv_bwd_kernel_mixed_kernel(BwdKArgs p) {
    __shared__ __attribute__((aligned(16))) float smem_shared[2 * (tile_floats<true, 64 * TI>() + tile_floats<true, 128>())];
    const int ntile = p.tiles_m * p.tiles_n, nwide = p.tiles_m * (p.tiles_n - 1);
    int id, unused;
    tile_coords(blockIdx.x, gridDim.x, 1, (int)gridDim.x, unused, id);  // XCD-chunked linear id
    int split, tile;
    const int both = p.narrow_splits * ntile;
    if (id < both) {
        split = id / ntile;
        tile = id - split * ntile;
    } else {
        const int r = id - both, s2 = r / nwide, w = r - s2 * nwide;
        split = p.narrow_splits + s2;
        tile = (w / (p.tiles_n - 1)) * p.tiles_n + w % (p.tiles_n - 1);
    }
    if (tile % p.tiles_n == p.tiles_n - 1) conv_bwd_kernel_mixed_body<VEC_G, TI, 1, QUADR, SW>(p, smem_shared, tile, split);
    else conv_bwd_kernel_mixed_body<VEC_G, TI, 2, QUADR, SW>(p, smem_shared, tile, split);
}

// dW[i] += sum_s slabs[s][i].  64 elements x 4 split-lanes per block (lane j sums splits j, j+4, ... with two independent
// accumulators), folded through LDS in a fixed order: `splits/4` loads deep instead of `splits` (the serial form took 28 us
// for 30 MB at C3).  Deterministic.
// `db` (optional): the bias gradient's per-split sums reduced by the blocks behind the dW ones, in the same launch
// (`Kc`, `narrow_col0`, `narrow_splits`: the mixed launch - columns from narrow_col0 on have only narrow_splits slabs; Kc = 0: off)
__global__ void conv_dw_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slabs, long long n, int splits, int assign,
                                      float* __restrict__ db = nullptr, const float* __restrict__ bias_slabs = nullptr, long long nb = 0,
                                      int assign_b = 0, int Kc = 0, int narrow_col0 = 0, int narrow_splits = 0) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, lane = threadIdx.x >> 6;
    long long blk = blockIdx.x;
    const long long dw_blocks = (n + 63) / 64;
    if (blk >= dw_blocks) {  // block-uniform: this block belongs to the bias gradient
        blk -= dw_blocks; dw = db; slabs = bias_slabs; n = nb; assign = assign_b; Kc = 0;
    }
    const long long i = blk * 64 + col;
    if (Kc > 0 && (int)(i % Kc) >= narrow_col0) splits = narrow_splits;
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

