// The block program of conv_bwd_kernel_kernel, textually shared with conv_bwd_kernel_mixed_kernel's wide and narrow bodies (included
// by nk_conv_generic.h inside both; not a stand-alone header).  The single-shape kernel stays the source the compiler saw before
// the mixed kernel existed (see nk_gemm_body.h for why).  The including scope provides
//   NK_BWK_DECODE   statements that define `const int tm, tn` (this block's tile) and `int split`
//   NK_BWK_SMEM     the declaration of `smem` (2 * STAGE floats of LDS)
//   NK_BWK_RPS      reduction indices per split for this block's tile (a multiple of BK)
// and the template parameters VEC_G, TI, TJ, QUADR, SW plus `BwdKArgs p`.
    static_assert(SW == 1 || QUADR, "strided quads: the row-padded form only");
    constexpr int BM = 64 * TI, BN = 64 * TJ;
    constexpr int TA_FLOATS = tile_floats<true, BM>(), STAGE = TA_FLOATS + tile_floats<true, BN>();
    NK_BWK_SMEM
    const ConvGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, wr = wid >> 1, wc = wid & 1;
    NK_BWK_DECODE
    const int grp = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Kc = g.Cg * g.KK;  // columns of dW
    const int W4 = (g.out[2] + 3) & ~3;
    const long long R = QUADR ? (long long)g.N * g.out[0] * g.out[1] * W4 : (long long)g.N * g.L;
    const long long rbeg = split * NK_BWK_RPS;
    const long long rend = rbeg + NK_BWK_RPS < R ? rbeg + NK_BWK_RPS : R;
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
