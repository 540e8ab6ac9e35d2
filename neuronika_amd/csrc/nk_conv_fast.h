// Fast (tap-major) implicit-GEMM convolution kernels: channel counts per group that are multiples of 32  -  part of the convolution translation unit (included by nk_conv.hip inside its anonymous
// namespace; not a stand-alone header).
#pragma once

// =================================================================================================
// Fast paths (tap-major reduction order).  When the channel count per group is a multiple of
// 32, a k-tile of 32 covers 32 channels of ONE kernel tap, so the tap decode / border test is
// done once per k-tile (wave-uniform, scalar) instead of once per k row, the per-row address is
// `base + row * plane`, and — for unit stride along the innermost axis — the four columns a
// thread stages are one unaligned 16-B load.  The weights are re-ordered once per call by a
// tiny pre-kernel (they are KBs to MBs; the activations are hundreds of MBs).
// =================================================================================================
// Wp[grp][co][tap][ci] = W[grp*Mg + co][ci][tap]   (forward A operand, k = tap*Cg + ci); the first block also writes the tap
// tables (one launch in front of every forward pass): tapoff[tap] = input offset of kernel tap `tap` relative to the window
// origin, tapd[tap] = its (dilated) coordinates
__global__ void conv_wp_kernel(float* __restrict__ wp, const float* __restrict__ w, ConvGeom g, int* __restrict__ tapoff = nullptr,
                               int4* __restrict__ tapd = nullptr) {
    if (tapoff && blockIdx.x == 0) {
        for (int tap = threadIdx.x; tap < g.KK; tap += blockDim.x) {
            int rem = tap;
            const int k2 = rem % g.k[2]; rem /= g.k[2];
            const int k1 = rem % g.k[1];
            const int k0 = rem / g.k[1];
            tapoff[tap] = (k0 * g.dil[0] * g.in[1] + k1 * g.dil[1]) * g.in[2] + k2 * g.dil[2];
            tapd[tap] = make_int4(k0 * g.dil[0], k1 * g.dil[1], k2 * g.dil[2], 0);
        }
    }
    const long long total = (long long)g.Cout * g.Cg * g.KK;
    if (total < 0x7fffffffLL) {  // 32-bit index arithmetic (a 64-bit division by a run-time value is a few hundred instructions)
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
            const unsigned rem = i / (unsigned)g.Cg, ci = i - rem * (unsigned)g.Cg, co = rem / (unsigned)g.KK, tap = rem - co * (unsigned)g.KK;
            wp[i] = w[(co * (unsigned)g.Cg + ci) * (unsigned)g.KK + tap];
        }
        return;
    }
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
// (the by-value table goes to device memory in conv_wq_tables_kernel: indexing a by-value array with a run-time index would
// spill it to scratch; a kernel instead of a host copy keeps the call capturable in a hipGraph)
__device__ __forceinline__ int tap_phase(const ConvGeom& g, int tap, int* kd) {
    int rem = tap;
    kd[2] = (rem % g.k[2]) * g.dil[2]; rem /= g.k[2];
    kd[1] = (rem % g.k[1]) * g.dil[1];
    kd[0] = (rem / g.k[1]) * g.dil[0];
    return ((kd[0] % g.stride[0]) * g.stride[1] + kd[1] % g.stride[1]) * g.stride[2] + kd[2] % g.stride[2];
}
// Wq[grp][ci][phase][chunk][tap in phase][c32] = W[grp*Mg + chunk*32 + c32][ci][tap]   (backward-input A operand).  Per
// phase, k runs over 32-channel chunks of co with the taps INSIDE a chunk: the 32 x (tile + halo) slab of the gradient
// that one chunk needs is then re-read by all taps back to back (L2 hits) instead of once per tap across all of co (PMC:
// 1.7 GB fetched per launch at C3 with the tap-major order, 9x the gradient).
// (begin, count, rank, phase) of a tap among the taps sorted by phase
__device__ __forceinline__ int4 tap_position(const ConvGeom& g, int tap, int* kd) {
    int kd2[3];
    const int pid = tap_phase(g, tap, kd);
    int begin = 0, cnt = 0, rank = 0;
    for (int t2 = 0; t2 < g.KK; ++t2) {
        const int pid2 = tap_phase(g, t2, kd2);
        if (pid2 < pid) ++begin;
        else if (pid2 == pid) { ++cnt; if (t2 < tap) ++rank; }
    }
    return make_int4(begin, cnt, rank, pid);
}
// One launch in front of the input-gradient pass: the first block writes the tap table - tapd[position in phase order] =
// {d0, d1, d2, tap} with out = q - d (d = (k*dil - r)/stride >= 0) - and the phase table, every block re-lays its share of the weights.
__global__ void conv_wq_tables_kernel(float* __restrict__ wq, const float* __restrict__ w, int4* __restrict__ tapd,
                                      BwdInPhase* __restrict__ phases, BwdInPhaseTable tbl, ConvGeom g) {
    // Every block first works out the KK taps' positions among the phases into LDS - (begin, count, rank, phase) per tap and the
    // inverse, position in phase order -> tap - in two steps: each tap's phase (one decode per thread), then its rank among the
    // taps of that phase (a loop over the KK phases in LDS: a few instructions per trip).  tap_position's loop of KK decodes per
    // thread (3 divisions by run-time values each) was the whole launch: ~5 us per block, twice in block 0 - 10.7 - 13.5 us at C3
    // for 9 taps and 73,728 weights against 4.8 for the forward's re-ordering of the same tensor.  A block then walks its share of
    // the DESTINATION: consecutive threads write consecutive floats of wq and gather from w (295 KB at C3: L2 hits).
    constexpr int TAPS_IN_LDS = 512;
    __shared__ int4 tp_s[TAPS_IN_LDS];
    __shared__ int tap_at[TAPS_IN_LDS];
    __shared__ int pid_s[TAPS_IN_LDS];
    const bool in_lds = g.KK <= TAPS_IN_LDS;
    if (blockIdx.x == 0) {
        if (!in_lds) {
            for (int tap = threadIdx.x; tap < g.KK; tap += blockDim.x) {
                int kd[3];
                const int4 tp = tap_position(g, tap, kd);
                tapd[tp.x + tp.z] = make_int4((kd[0] - kd[0] % g.stride[0]) / g.stride[0], (kd[1] - kd[1] % g.stride[1]) / g.stride[1],
                                              (kd[2] - kd[2] % g.stride[2]) / g.stride[2], tap);
            }
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < MAX_PHASES; ++i) phases[i] = tbl.ph[i];
        }
    }
    const long long total = (long long)g.Cout * g.Cg * g.KK;
    if (in_lds) {
        for (int tap = threadIdx.x; tap < g.KK; tap += blockDim.x) {
            int kd[3];
            pid_s[tap] = tap_phase(g, tap, kd);
        }
        __syncthreads();
        for (int tap = threadIdx.x; tap < g.KK; tap += blockDim.x) {
            const int pid = pid_s[tap];
            int begin = 0, cnt = 0, rank = 0;
            for (int t2 = 0; t2 < g.KK; ++t2) {
                const int pid2 = pid_s[t2];
                begin += pid2 < pid;
                cnt += pid2 == pid;
                rank += pid2 == pid && t2 < tap;
            }
            tp_s[tap] = make_int4(begin, cnt, rank, pid);
            tap_at[begin + rank] = tap;
            if (blockIdx.x == 0) {
                int kd[3];
                tap_phase(g, tap, kd);
                tapd[begin + rank] = make_int4((kd[0] - kd[0] % g.stride[0]) / g.stride[0], (kd[1] - kd[1] % g.stride[1]) / g.stride[1],
                                               (kd[2] - kd[2] % g.stride[2]) / g.stride[2], tap);
            }
        }
        __syncthreads();
        const int row = g.Mg * g.KK;  // one (group, input channel) row of wq
        auto element = [&](long long i, int r, int ci, int grp) {
            const int4 ph = tp_s[tap_at[r / g.Mg]];  // the phase region r lies in: positions [begin, begin + count)
            const int rr = r - g.Mg * ph.x, c32 = rr % BK, t = rr / BK;
            const int rank = t % ph.y, chunk = t / ph.y;
            const int tap = tap_at[ph.x + rank];
            wq[i] = w[(((long long)grp * g.Mg + chunk * BK + c32) * g.Cg + ci) * g.KK + tap];
        };
        if (total < 0x7fffffffLL) {  // 32-bit index arithmetic (a 64-bit division by a run-time value is a few hundred instructions)
            for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
                const unsigned gc = i / (unsigned)row, r = i - gc * (unsigned)row, grp = gc / (unsigned)g.Cg;
                element(i, (int)r, (int)(gc - grp * (unsigned)g.Cg), (int)grp);
            }
            return;
        }
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const long long gc = i / row;
            element(i, (int)(i % row), (int)(gc % g.Cg), (int)(gc / g.Cg));
        }
        return;
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {  // i = source index ((grp*Mg + co)*Cg + ci)*KK + tap
        const int tap = (int)(i % g.KK);
        long long rem = i / g.KK;
        const int ci = (int)(rem % g.Cg); rem /= g.Cg;
        const int co = (int)(rem % g.Mg);
        const int grp = (int)(rem / g.Mg);
        int kd[3];
        const int4 tp = tap_position(g, tap, kd);
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

// requires Cg % 32 == 0, stride[2] == 1 or 2 (SW), out[2] >= 4, per-tensor element counts < 2^31.
// Columns are (n, o0, o1, c') with the innermost output row padded to W4 = a multiple of 4, so the quad a thread stages is
// four consecutive positions of ONE output row = one unaligned 16-byte load per staged row.  RP (out[2] % 4 != 0): the
// last quad of a row is loaded `dup` elements earlier (so that it ends inside the input row) and shifted left by `dup`
// behind the MFMAs; its trailing `dup` columns are dummies whose accumulators are never stored.
// SW = stride on the innermost axis, 1 or 2.  SW = 2 (the downsampling convolutions of a ResNet-style stack): the four consecutive
// output positions of a quad read input positions 0, 2, 4, 6 from the quad's origin - two unaligned 16-byte loads per staged row,
// at +0 (elements 0 and 2 are taken) and at +3 (elements 1 and 3, i.e. positions 4 and 6; a load at +4 would read one float past
// the last position the convolution needs, which at the end of the tensor is past the allocation).
template <bool ALIGNED_A, int TI, bool RP, int SW = 1>
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
        xb = n * g.Cin * g.inplane + (oa * g.stride[0] * g.in[1] + ob * g.stride[1]) * g.in[2] + cs * SW + krow * g.inplane;
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
        if constexpr (SW == 2) {
            const f32x4u h0 = *reinterpret_cast<const f32x4u*>(src + xb + 3);
            const f32x4u h1 = *reinterpret_cast<const f32x4u*>(src + xb + jstep + 3);
            const f32x4u h2 = *reinterpret_cast<const f32x4u*>(src + xb + 2 * jstep + 3);
            const f32x4u h3 = *reinterpret_cast<const f32x4u*>(src + xb + 3 * jstep + 3);
            r.v0 = make_float4(q0.x, q0.z, h0.y, h0.w);
            r.v1 = make_float4(q1.x, q1.z, h1.y, h1.w);
            r.v2 = make_float4(q2.x, q2.z, h2.y, h2.w);
            r.v3 = make_float4(q3.x, q3.z, h3.y, h3.w);
            return r;
        }
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

// sum over the splits of one element of a split tail tile, in split order (the SAME order as a plain loop: deterministic and
// bit-identical to it), four loads in flight at a time - the one-load-per-trip loop ran at 1.4 TB/s (17.7 us for 25 MB at C3)
__device__ __forceinline__ float tail_split_sum(const float* __restrict__ e, int splits, int stride) {
    float s = 0.f;
    if (splits <= 8) {  // every load issued before the first add (C3: 6 splits in the forward pass, 12 in the input gradient)
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = e[(long long)min(k, splits - 1) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) s = k < splits ? s + v[k] : s;
        return s;
    }
    if (splits <= 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = e[(long long)min(k, splits - 1) * stride];
#pragma unroll
        for (int k = 0; k < 16; ++k) s = k < splits ? s + v[k] : s;
        return s;
    }
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
        const float a = e[(long long)k * stride], b = e[(long long)(k + 1) * stride], c = e[(long long)(k + 2) * stride],
                    d = e[(long long)(k + 3) * stride];
        s += a; s += b; s += c; s += d;
    }
    for (; k < splits; ++k) s += e[(long long)k * stride];
    return s;
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
    float sums[4];  // the slabs hold whole tiles: the four elements' sums are read unconditionally, all loads in flight together
#pragma unroll
    for (int i = 0; i < 4; ++i) sums[i] = tail_split_sum(base + e + i * 256, p.tail_splits, BM * BN);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ee = e + i * 256;
        const int r = ee / BN, c = ee - r * BN;
        const int co = tm * BM + r, cc = tn * BN + c;
        if (co >= g.Mg || cc >= cols) continue;
        const float s = sums[i];
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
    const float* wq;  // [groups][Cg][KK*Mg], phase-sorted (conv_wq_tables_kernel)
    const int4* tapd;
    const BwdInPhase* phases;
    int nphase;
    int tiles_m, tiles_n;  // tiles_n: column tiles of all phases together
    // Tail balancing (single phase, one group), as in the forward pass: the first `full_blocks` tiles (whole waves of
    // resident blocks) are computed by one block each; every remaining tile is split over `tail_splits` blocks of
    // `tail_kts` k-tiles that write partial tiles to `slabs` ([tail tile][split][BM][BN]);
    // conv_bwd_input_tail_reduce_kernel sums them in split order into dX.  tail_splits == 0: off.
    int full_blocks, tail_splits, tail_kts;
    float* slabs;
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
    int kt0 = 0, tail_nt = -1;
    float* slab = nullptr;
    if (p.tail_splits == 0) {
        tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    } else if ((int)blockIdx.x < p.full_blocks) {
        tile_of_seq(xcd_chunk(blockIdx.x, p.full_blocks), p.tiles_m, p.tiles_n, tm, tn);
    } else {
        const int tb = blockIdx.x - p.full_blocks, tail_tile = tb / p.tail_splits, split = tb - tail_tile * p.tail_splits;
        tile_of_seq(p.full_blocks + tail_tile, p.tiles_m, p.tiles_n, tm, tn);
        kt0 = split * p.tail_kts;
        tail_nt = p.tail_kts;
        slab = p.slabs + ((long long)tail_tile * p.tail_splits + split) * (BM * BN);
    }
    int pid = 0;  // the stride phase this column tile belongs to (block-uniform)
    for (int i = 1; i < p.nphase; ++i)
        if (tn >= p.phases[i].tile_begin) pid = i;
    const BwdInPhase ph = p.phases[pid];
    tn -= ph.tile_begin;
    const int grp = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = g.Mg * g.KK;  // K: row length of Wq; this phase reduces over Mg * ntaps
    const int nt_all = g.Mg * ph.ntaps / BK, nt = tail_nt < 0 ? nt_all : min(tail_nt, nt_all - kt0);
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
    // last element index at which a lane's four 16-byte loads (rows 0, 8, 16, 24 of the k-tile) may start
    const unsigned gy_last_start = (unsigned)((long long)g.N * g.Cout * g.L - 3LL * jstep - 4);
    // Two halves: `gather` only ISSUES the four 16-byte loads of the next k-tile (before the MFMAs of the current one);
    // `gather_finish` picks / masks the elements and runs AFTER the MFMAs - touching the loaded registers any earlier
    // makes the wave wait for its loads with nothing to hide them behind.
    struct GatherState { int sh = 0, c = 0; bool ok = false; };  // what gather_finish needs to know about the loads it completes
    auto gather_into = [&](int kt, Stage<4>& rb, GatherState& st) {
        kt += kt0;
        const int chunk = kt / ntaps, tap = kt - chunk * ntaps, co0 = chunk * BK;  // taps inside a 32-channel chunk
        const int4 d = tapd[tap];
        const float* src = G + co0 * g.L;
        const int a = qa - d.x, b = qb - d.y, c = qc - d.z;  // output coordinates of the quad's first element
        st.ok = valid && a >= 0 && a < g.out[0] && b >= 0 && b < g.out[1];
        const int ac = min(max(a, 0), g.out[0] - 1), bc = min(max(b, 0), g.out[1] - 1);
        // The load starts AT the quad (element i of the vector = element i of the quad, no shift): a quad that sticks out
        // of its gradient row reads the neighbouring row's elements, which the per-element masks zero.  Only a load that
        // would leave the TENSOR (before the first / behind the last element of gy: a handful of lanes per launch) is
        // pulled back into its row and takes the shifting path.
        // (32-bit: every tensor has fewer than 2^31 elements, checked by the host; one unsigned compare covers both ends -
        //  a start before the tensor wraps to a huge value)
        const int e0 = (int)(src - p.gy) + gbase + (ac * g.out[1] + bc) * g.out[2] + c;
        const bool edge = (unsigned)e0 > gy_last_start;
        const int cs = edge ? min(max(c, 0), g.out[2] - 4) : c;
        st.sh = c - cs;                                  // shift of element 0 inside the loaded vector (0 unless `edge`)
        st.c = c;
        const float* ptr = src + (gbase + (ac * g.out[1] + bc) * g.out[2] + cs);
#define NK_LDU(V, P) { const f32x4u q = *reinterpret_cast<const f32x4u*>(P); V = make_float4(q.x, q.y, q.z, q.w); }
        NK_LDU(rb.v0, ptr) NK_LDU(rb.v1, ptr + jstep) NK_LDU(rb.v2, ptr + 2 * jstep) NK_LDU(rb.v3, ptr + 3 * jstep)
#undef NK_LDU
    };
    auto finish_of = [&](Stage<4>& rb, const GatherState& st) {
        const int sh = st.sh, g_c = st.c;
        const bool g_ok = st.ok;
        // element i of the quad lies in its gradient row: one unsigned compare each
        const bool in0 = g_ok && (unsigned)g_c < (unsigned)g.out[2], in1 = g_ok && (unsigned)(g_c + 1) < (unsigned)g.out[2],
                   in2 = g_ok && (unsigned)(g_c + 2) < (unsigned)g.out[2], in3 = g_ok && (unsigned)(g_c + 3) < (unsigned)g.out[2];
        // A wave in which some lane's load was pulled back into its row (sh != 0: the quad would have left the TENSOR - a
        // handful of lanes per launch) shifts first: element i of the quad = element i + sh of the loaded vector, a barrel
        // shifter of register selects (a pick by dynamic index makes the compiler index the vector through scratch
        // memory).  Wave-uniform branch with no memory operation in either arm (the wait-count bookkeeping is the same on
        // both paths): every other wave skips the 44 selects - they were 40 % of the pass's VALU instructions, and f32 MFMA
        // and VALU share the issue slots.
        if (__any(sh != 0)) {
            const int sl = max(sh, 0), sr = max(-sh, 0);
            const bool l1 = sl & 1, l2 = sl & 2, r1 = sr & 1, r2 = sr & 2;
            auto sel = [&](float4& q) {
                float e0 = q.x, e1 = q.y, e2 = q.z, e3 = q.w;
                e0 = l1 ? e1 : e0; e1 = l1 ? e2 : e1; e2 = l1 ? e3 : e2;
                e0 = l2 ? e2 : e0; e1 = l2 ? e3 : e1;
                e3 = r1 ? e2 : e3; e2 = r1 ? e1 : e2; e1 = r1 ? e0 : e1;
                e3 = r2 ? e1 : e3; e2 = r2 ? e0 : e2;
                q.x = e0; q.y = e1; q.z = e2; q.w = e3;
            };
            sel(rb.v0); sel(rb.v1); sel(rb.v2); sel(rb.v3);
        }
        auto keep = [&](float4& q) { q.x = in0 ? q.x : 0.f; q.y = in1 ? q.y : 0.f; q.z = in2 ? q.z : 0.f; q.w = in3 ? q.w : 0.f; };
        keep(rb.v0); keep(rb.v1); keep(rb.v2); keep(rb.v3);
    };

    Stage<4> rb;
    GatherState gs;
    auto gather = [&](int kt) { gather_into(kt, rb, gs); };
    auto gather_finish = [&]() { finish_of(rb, gs); };

    f32x16 acc[TI][TJ];
    acc_zero<TI, TJ>(acc);
    TileLoader<true, BM> la;
    la.init(Wq, K, m0, kt0 * BK, g.Cg, (kt0 + nt) * BK, t);
    Stage<BM / 32> ra;
    // TI == 1 (Cin = 64: a k-tile is 32 MFMAs per wave, ~2000 cycles - less than a load round trip through L2): two k-tiles
    // of look-ahead as in sgemm_kernel.  Tile it+1 (P, gathered during the previous trip) is masked and goes to LDS at the
    // START of a trip, the loads of tile it+2 (Q) are issued in front of it; a trip ends MFMAs -> barrier.
    if (TI == 1 && nt >= 8) {
        float* const buf0 = smem;
        float* const buf1 = smem + STAGE;
        Stage<BM / 32> pa, qa2;
        Stage<4> pb, qb2;
        GatherState ps, qs;
        pa = la.template load<ALIGNED_A>(t);
        gather_into(0, pb, ps);
        finish_of(pb, ps);
        stage_store<true, BM>(buf0, pa, t);
        stage_store<false, BN>(buf0 + TA_FLOATS, pb, t);
        __syncthreads();
        pa = la.template load<ALIGNED_A>(t);  // tile 1 -> P
        gather_into(1, pb, ps);
        int it = 0;  // invariant: tile `it` (even) is in buf0, tile it+1 in P
        for (; it + 3 < nt; it += 2) {
            qa2 = la.template load<ALIGNED_A>(t);  // tile it+2
            gather_into(it + 2, qb2, qs);
            finish_of(pb, ps);
            stage_store<true, BM>(buf1, pa, t);
            stage_store<false, BN>(buf1 + TA_FLOATS, pb, t);
            __builtin_amdgcn_sched_barrier(0);
            mma_tile<true, false, TI, TJ>(buf0, buf0 + TA_FLOATS, acc, wr, wc, lane);
            __syncthreads();
            pa = la.template load<ALIGNED_A>(t);  // tile it+3
            gather_into(it + 3, pb, ps);
            finish_of(qb2, qs);
            stage_store<true, BM>(buf0, qa2, t);
            stage_store<false, BN>(buf0 + TA_FLOATS, qb2, t);
            __builtin_amdgcn_sched_barrier(0);
            mma_tile<true, false, TI, TJ>(buf1, buf1 + TA_FLOATS, acc, wr, wc, lane);
            __syncthreads();
        }
        const int left = nt - it;  // 2 or 3 tiles: `it` in buf0, it+1 in P
        if (left == 3) {
            qa2 = la.template load<ALIGNED_A>(t);
            gather_into(it + 2, qb2, qs);
        }
        finish_of(pb, ps);
        stage_store<true, BM>(buf1, pa, t);
        stage_store<false, BN>(buf1 + TA_FLOATS, pb, t);
        mma_tile<true, false, TI, TJ>(buf0, buf0 + TA_FLOATS, acc, wr, wc, lane);
        __syncthreads();
        if (left == 3) {
            finish_of(qb2, qs);
            stage_store<true, BM>(buf0, qa2, t);
            stage_store<false, BN>(buf0 + TA_FLOATS, qb2, t);
        }
        mma_tile<true, false, TI, TJ>(buf1, buf1 + TA_FLOATS, acc, wr, wc, lane);
        if (left == 3) {
            __syncthreads();
            mma_tile<true, false, TI, TJ>(buf0, buf0 + TA_FLOATS, acc, wr, wc, lane);
        }
    } else {
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
    }
    if (slab) {  // partial tile of a split tail tile
        acc_foreach<TI, TJ>(acc, wr, wc, lane, [&](int r, int c, float v) { slab[r * BN + c] = v; });
        return;
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


// dX[tail tiles] (+)= sum over splits (fixed order) of the partial tiles.  Single phase, one group.
template <int BM>
__global__ void conv_bwd_input_tail_reduce_kernel(FastBwdInArgs p) {
    constexpr int BN = 128;
    const ConvGeom& g = p.g;
    const BwdInPhase ph = p.phases[0];
    const int tail_tile = blockIdx.x;
    int tm, tn;
    tile_of_seq(p.full_blocks + tail_tile, p.tiles_m, p.tiles_n, tm, tn);
    const float* base = p.slabs + ((long long)tail_tile * p.tail_splits) * (BM * BN);
    const int W4 = (ph.count[2] + 3) & ~3, rows_per_n = ph.count[0] * ph.count[1];
    const int cols = g.N * rows_per_n * W4;
    // blockIdx.y: a 1024-element slice of the tile (8 rows x 128 columns); consecutive threads = consecutive columns
    const int e = blockIdx.y * 1024 + threadIdx.x;
    float sums[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sums[i] = tail_split_sum(base + e + i * 256, p.tail_splits, BM * BN);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ee = e + i * 256;
        const int r = ee / BN, c = ee - r * BN;
        const int ci = tm * BM + r, cc = tn * BN + c;
        if (ci >= g.Cg || cc >= cols) continue;
        const int rowid = cc / W4, cpos = cc - rowid * W4;
        if (cpos >= ph.count[2]) continue;  // padding column of the row
        const float s = sums[i];
        const int n = rowid / rows_per_n, ab = rowid - n * rows_per_n;
        const int i0 = ab / ph.count[1], i1 = ab - i0 * ph.count[1];
        float* d = p.dx + ((long long)n * g.Cin + ci) * g.uinplane +
                   ((long long)(ph.first[0] + i0 * g.stride[0]) * g.uin[1] + ph.first[1] + i1 * g.stride[1]) * g.uin[2] + ph.first[2] +
                   cpos * g.stride[2];
        *d = g.assign ? s : *d + s;
    }
}
