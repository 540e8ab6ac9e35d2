// The block program of sgemm_kernel, textually shared with sgemm_pair_kernel's two problem bodies (nk_gemm.hip includes this
// file inside both): the single-problem kernel must stay the SAME source the compiler saw before the pair kernel existed - its
// k-loop schedule and register allocation move by a percent or more when the surrounding code changes shape (a refactoring
// into a function of (block id, blocks) cost NT 4096^3 1.2 % and all layouts 0.9 % at 2048^3, same box) - so the only thing
// that differs between the two uses is where a block's position in the tile sequence comes from:
//   NK_GEMM_BX / NK_GEMM_NBX   this block's index among the blocks of ITS problem / their number
//   NK_GEMM_SPLIT              which range of the reduction it takes (blockIdx.y, except in sgemm_tail_kernel's second problem)
// In scope: template parameters TA, TB, ALIGNED, TI, TJ, KG, EPX; `const GemmArgs& p`; `float* smem_all` (the block's LDS).
// grid.z = batch in every use.  Not a stand-alone header.
    constexpr int BM = 64 * TI, BN = 64 * TJ;
    constexpr bool AKC = !TA;  // A (M x K): k-contiguous unless stored transposed
    constexpr bool BKC = TB;   // B (K x N) stored as N x K when transposed -> k-contiguous
    constexpr int TA_FLOATS = tile_floats<AKC, BM>(), STAGE = TA_FLOATS + tile_floats<BKC, BN>();
    static_assert(KG == 1 || (ALIGNED && 2 * 2 * STAGE >= BM * BN), "k-pair: aligned problems; a group's images hold half a C tile");
    static_assert(KG * 2 * STAGE * sizeof(float) <= 160 * 1024, "the block's LDS images must fit the 160 KB of a gfx950 CU");
    const int grp = KG == 2 ? (int)(threadIdx.x >> 8) : 0;  // NT == 256
    // (the mask is a no-op for the 256-thread blocks, but it tells the compiler the index has 8 bits - launch bounds do not: the
    // address code of every instantiation came out 100 - 450 instructions shorter, TN 4096^3 137.8 -> 139.4 TFLOP/s in three
    // alternating same-box runs, the other layouts and sizes within +-0.3 %; the same mask in the conv kernels LOSES 1.2 - 1.4 % on
    // the forward and kernel-gradient passes and does nothing for the attention kernels: GEMM only)
    const int t = (int)(threadIdx.x & (NT - 1)), lane = t & 63, wid = t >> 6;
    float* const smem = smem_all + grp * 2 * STAGE;
    const int wr = wid >> 1, wc = wid & 1;
    // this block's tiles: positions [seq, seq_end) of the tile sequence (each XCD gets a contiguous range of chunks)
    int seq = xcd_chunk(NK_GEMM_BX, NK_GEMM_NBX) * p.chunk;
    const int seq_end = min(p.tiles_m * p.tiles_n, seq + p.chunk);
    int tm, tn;
    tile_of_seq(seq, p.tiles_m, p.tiles_n, tm, tn, p.group_m);
    int m0 = tm * BM, n0 = tn * BN;
    const int batch = blockIdx.z, split = NK_GEMM_SPLIT;
    const int bo = batch / p.batch_inner, bi = batch % p.batch_inner;
    const float* A = p.A + bo * p.sAo + bi * p.sAi;
    const float* B = p.B + bo * p.sBo + bi * p.sBi;

    int kbeg = split * p.k_per_split;
    int kend = min(p.K, kbeg + p.k_per_split);
    if (KG == 2) {  // group g takes the g-th half; the host only asks for the pair when that is a whole number of k-tiles
        const int half = (kend - kbeg) >> 1;
        if (grp) kbeg += half; else kend = kbeg + half;
    }
    const int nt = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[TI][TJ];
    acc_zero<TI, TJ>(acc);
    TileLoader<AKC, BM> la;
    TileLoader<BKC, BN> lb;
    // Two-k-tile look-ahead: aligned problems only (the guarded loader's state does not fit next to P and Q), every
    // layout, from the per-layout / per-tile k-tile threshold the host passes in `pf2_min` (rules and their same-box
    // sweeps: gemm_impl).  That loop handles exactly ONE tile: gemm_impl sets chunk = 1 whenever nt >= pf2_min.
    constexpr bool PF2 = ALIGNED;
    const bool skew = KG == 2 && grp != 0 && p.kskew != 0;  // wave-uniform
    la.init(A, p.lda, m0, kbeg, p.M, kend, t);
    lb.init(B, p.ldb, n0, kbeg, p.N, kend, t);
    if (PF2 && (KG == 2 || nt >= p.pf2_min)) {
        gemm_loop_lookahead2<AKC, BKC, ALIGNED, TI, TJ, KG>(la, lb, acc, smem, nt, skew, t, wr, wc, lane);
    } else if constexpr (KG == 1) {
    // One k-tile of look-ahead, over the block's whole chunk of tiles: the loads issued in front of the LAST MFMA block
    // of a tile are the first k-tile of the NEXT tile.
    Stage<BM / 32> ra;
    Stage<BN / 32> rb;
    if (nt > 0) {
        ra = la.template load<ALIGNED>(t);
        rb = lb.template load<ALIGNED>(t);
        stage_store<AKC, BM>(smem, ra, t);
        stage_store<BKC, BN>(smem + TA_FLOATS, rb, t);
    }
    __syncthreads();
    int par = 0;  // LDS buffer that holds the current k-tile
    for (;;) {
        for (int it = 0; it + 1 < nt; ++it) {
            float* cur = smem + par * STAGE;
            float* nxt = smem + (par ^ 1) * STAGE;
            // issue the next tile's HBM/L2 loads before the MFMAs (their latency hides under them),
            // write them to the other LDS buffer behind the MFMAs: one barrier per k-tile
            ra = la.template load<ALIGNED>(t);
            rb = lb.template load<ALIGNED>(t);
            __builtin_amdgcn_sched_barrier(0);
            mma_tile<AKC, BKC, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
            stage_store<AKC, BM>(nxt, ra, t);
            stage_store<BKC, BN>(nxt + TA_FLOATS, rb, t);
            __syncthreads();
            par ^= 1;
        }
        const bool more = seq + 1 < seq_end && nt > 0;
        int tm2 = 0, tn2 = 0;
        if (more) {
            tile_of_seq(seq + 1, p.tiles_m, p.tiles_n, tm2, tn2, p.group_m);
            la.init(A, p.lda, tm2 * BM, kbeg, p.M, kend, t);
            lb.init(B, p.ldb, tn2 * BN, kbeg, p.N, kend, t);
            ra = la.template load<ALIGNED>(t);
            rb = lb.template load<ALIGNED>(t);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nt > 0) {
            float* cur = smem + par * STAGE;
            mma_tile<AKC, BKC, TI, TJ>(cur, cur + TA_FLOATS, acc, wr, wc, lane);
        }
        if (more) {
            float* nxt = smem + (par ^ 1) * STAGE;
            stage_store<AKC, BM>(nxt, ra, t);
            stage_store<BKC, BN>(nxt + TA_FLOATS, rb, t);
        }
        gemm_epilogue<ALIGNED, TI, TJ, EPX>(p, acc, m0, n0, bo, bi, split, batch, wr, wc, lane);
        if (!more) return;
        __syncthreads();  // the next tile's first k-tile is in LDS, everybody is done with the current buffer
        par ^= 1;
        acc_zero<TI, TJ>(acc);
        ++seq;
        m0 = tm2 * BM; n0 = tn2 * BN;
    }
    }
    if constexpr (KG == 2) {
        // acc(group 0: first half of the reduction) + acc(group 1: second half).  The groups SWAP halves of the tile through
        // LDS - group g keeps its MFMA tile row g, sends the other row - so that all eight waves share the epilogue (half the
        // old-C loads and C stores per lane; the sum is the same bits in either operand order).  Each group writes into its
        // own images: [wave][column tile][quad][lane] float4, lane-contiguous 16-byte slots.
        if constexpr (TI == 1) {
            // 64-wide tiles (one MFMA tile per wave and column tile): group 1 hands its accumulators over, group 0 adds them
            // to its own - first half + second half, split-K 2's order - and stores the tile.
            __syncthreads();  // every wave has read its last k-tile
            float4* const slot = reinterpret_cast<float4*>(smem_all + 2 * STAGE) + wid * (TJ * 4 * 64) + lane;  // group 1's images
            if (grp == 1) {
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        slot[(j * 4 + q4) * 64] = make_float4(acc[0][j][4 * q4], acc[0][j][4 * q4 + 1], acc[0][j][4 * q4 + 2], acc[0][j][4 * q4 + 3]);
            }
            __syncthreads();
            if (grp == 1) return;
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 v = slot[(j * 4 + q4) * 64];
                    acc[0][j][4 * q4] += v.x; acc[0][j][4 * q4 + 1] += v.y; acc[0][j][4 * q4 + 2] += v.z; acc[0][j][4 * q4 + 3] += v.w;
                }
            gemm_epilogue<ALIGNED, TI, TJ, EPX>(p, acc, m0, n0, bo, bi, split, batch, wr, wc, lane);
            return;
        } else {
        static_assert(TI == 2, "k-pair: 128-row tiles swap halves");
        __syncthreads();  // every wave has read its last k-tile
        float4* const mine_out = reinterpret_cast<float4*>(smem) + wid * (TJ * 4 * 64) + lane;
        const float4* const theirs_in = reinterpret_cast<const float4*>(smem_all + (grp ^ 1) * 2 * STAGE) + wid * (TJ * 4 * 64) + lane;
        f32x16 keep[1][TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const f32x16 send = grp ? acc[0][j] : acc[1][j];
            keep[0][j] = grp ? acc[1][j] : acc[0][j];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                mine_out[(j * 4 + q4) * 64] = make_float4(send[4 * q4], send[4 * q4 + 1], send[4 * q4 + 2], send[4 * q4 + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = theirs_in[(j * 4 + q4) * 64];
                keep[0][j][4 * q4] += v.x; keep[0][j][4 * q4 + 1] += v.y; keep[0][j][4 * q4 + 2] += v.z; keep[0][j][4 * q4 + 3] += v.w;
            }
        gemm_epilogue<ALIGNED, 1, TJ, EPX>(p, keep, m0, n0, bo, bi, split, batch, wr * 2 + grp, wc, lane);
        return;
        }
    }
    gemm_epilogue<ALIGNED, TI, TJ, EPX>(p, acc, m0, n0, bo, bi, split, batch, wr, wc, lane);
