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
    static_assert(SW == 1 || QUADR, "strided quads: the row-padded form only");
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
#define NK_KO(j, KO, CV) { const int c = n0 + row + 32 * j; CV = c < Kc; KO = CV ? conv_koff(g, c) : 0; }  // (once per thread: no table)
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
        const long long x0 = v ? (long long)qn * g.Cin * g.inplane + ((q0 * g.stride[0] * g.in[1] + q1 * g.stride[1]) * g.in[2] + cs * SW) : 0;
        const long long g0 = v ? (long long)qn * g.Cout * g.L + ((q0 * g.out[1] + q1) * g.out[2] + cs) : 0;
        q2 += BK;  // next k-tile: 32 positions further along the (row-padded) reduction index
        while (q2 >= W4) { q2 -= W4; ++q1; }
        while (q1 >= g.out[1]) { q1 -= g.out[1]; ++q0; }
        while (q0 >= g.out[0]) { q0 -= g.out[0]; ++qn; }
#define NK_LDU(V, P) { const f32x4u q = *reinterpret_cast<const f32x4u*>(P); V = make_float4(q.x, q.y, q.z, q.w); }
        NK_LDU(ra.v0, G + g0 + aro0) NK_LDU(ra.v1, G + g0 + aro1)
        if constexpr (TI == 2) { NK_LDU(ra.v2, G + g0 + aro2) NK_LDU(ra.v3, G + g0 + aro3) }
        if constexpr (SW == 2) {
#define NK_LDS2(V, P) { const f32x4u lo = *reinterpret_cast<const f32x4u*>(P); const f32x4u hi = *reinterpret_cast<const f32x4u*>((P) + 3); \
                        V = make_float4(lo.x, lo.z, hi.y, hi.w); }
            NK_LDS2(rb.v0, X + x0 + ko0) NK_LDS2(rb.v1, X + x0 + ko1)
            if constexpr (TJ == 2) { NK_LDS2(rb.v2, X + x0 + ko2) NK_LDS2(rb.v3, X + x0 + ko3) }
#undef NK_LDS2
        } else {
            NK_LDU(rb.v0, X + x0 + ko0) NK_LDU(rb.v1, X + x0 + ko1)
            if constexpr (TJ == 2) { NK_LDU(rb.v2, X + x0 + ko2) NK_LDU(rb.v3, X + x0 + ko3) }
        }
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

    // Bias gradient of the conv module (sum of G over samples and positions per output channel), for free: the masked A
    // operand IS G, every thread adds its staged quads of its rows (16 adds per k-tile, no branch in the loop - a wave-uniform
    // `tn == 0` test there costs more than the adds, section 4.2 of DESIGN.md); the column-tile-0 blocks write the sums.
    float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;
    auto bias_acc = [&]() {
        bs0 += (ra.v0.x + ra.v0.y) + (ra.v0.z + ra.v0.w);
        bs1 += (ra.v1.x + ra.v1.y) + (ra.v1.z + ra.v1.w);
        if constexpr (TI == 2) {
            bs2 += (ra.v2.x + ra.v2.y) + (ra.v2.z + ra.v2.w);
            bs3 += (ra.v3.x + ra.v3.y) + (ra.v3.z + ra.v3.w);
        }
    };
    f32x16 acc[TI][TJ];
    acc_zero<TI, TJ>(acc);
    if (nt > 0) {
        load_both(rbeg);
        if constexpr (QUADR) { mask_quad(); bias_acc(); }
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
        if constexpr (QUADR) { mask_quad(); bias_acc(); }
        else mask_scalar();
        stage_store<true, BM>(nxt, ra, t);
        stage_store<true, BN>(nxt + TA_FLOATS, rb, t);
        __syncthreads();
    }
    if (nt > 0) {
        float* cur = smem + ((nt - 1) & 1) * STAGE;
        mma_tile<true, true, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
    }
    if (QUADR && p.bias_slabs && tn == 0) {  // the 8 lanes that staged one row (k-quads 0..7) are neighbours: fold, lane 0 writes
        float* Bsl = p.bias_slabs + ((long long)split * g.groups + grp) * g.Mg;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            bs0 += __shfl_xor(bs0, o, 64); bs1 += __shfl_xor(bs1, o, 64);
            bs2 += __shfl_xor(bs2, o, 64); bs3 += __shfl_xor(bs3, o, 64);
        }
        if (rq == 0) {
            if (av0) Bsl[m0 + row] = bs0;
            if (av1) Bsl[m0 + row + 32] = bs1;
            if (TI == 2 && av2) Bsl[m0 + row + 64] = bs2;
            if (TI == 2 && av3) Bsl[m0 + row + 96] = bs3;
        }
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
// `db` (optional): the bias gradient's per-split sums reduced by the blocks behind the dW ones, in the same launch
__global__ void conv_dw_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slabs, long long n, int splits, int assign,
                                      float* __restrict__ db = nullptr, const float* __restrict__ bias_slabs = nullptr, long long nb = 0,
                                      int assign_b = 0) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, lane = threadIdx.x >> 6;
    long long blk = blockIdx.x;
    const long long dw_blocks = (n + 63) / 64;
    if (blk >= dw_blocks) {  // block-uniform: this block belongs to the bias gradient
        blk -= dw_blocks; dw = db; slabs = bias_slabs; n = nb; assign = assign_b;
    }
    const long long i = blk * 64 + col;
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

