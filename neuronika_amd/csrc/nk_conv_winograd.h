// Winograd F(2x2, 3x3) for the 3x3 / stride 1 / dilation 1 / groups 1 convolution: forward (node/convolution/mod.rs:85-123)
// and input gradient (:146-189, the full correlation with the flipped kernel) - part of the convolution translation unit
// (included by nk_conv.hip inside its anonymous namespace; not a stand-alone header).
//
// Why: at C3 the implicit-GEMM passes run at the rate of a DENSE GEMM of their shape (0.75 - 0.77 of the f32 MFMA peak,
// profiles/r04_chunk_conv_shape.txt) - tuning is exhausted, the flops are not: a 2x2 output tile from a 4x4 input patch needs
// 16 multiplies per (co, ci) where the direct form needs 36.  Y = A^T [ (G g G^T) . (B^T d B) ] A with
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
// (Lavin & Gray 2016): the data and output transforms are additions, the kernel transform halves and quarters - exact on
// integer-valued data.  The 16 element-wise products, summed over input channels, are 16 small GEMMs
//   M[xi] (Cm x P) = U[xi] (Cm x Ck) . V[xi] (Ck x P),   xi = 0..15,  P = N * (Hd/2) * (Wd/2) tiles,
// which run on the MFMA core (v_mfma_f32_32x32x2_f32).  Everything stays on chip: a block transforms the patches of 32 tiles into LDS
// (V, two images of 64 KB / 32 KB), each wave holds the 16 accumulator tiles (xi) of its 32 output channels x 32 tiles in registers
// (256 of the 512 a wave has at one wave per SIMD), takes its U fragments straight from L2 in MFMA operand order (the kernel
// transform writes them that way), and applies the output transform to its own registers - V, M never touch HBM.
//   WIDE    four waves x 32 channels = 128 output channels, reduction channels in chunks of 32, one block per CU
//   NARROW  two waves = 64 output channels, chunks of 16, two independent blocks per CU
// either shape for either pass, by channel count (wino_launch).  One source for both passes: the "source" tensor is the input
// (forward) or the output gradient (backward), read at patch origin (2 ty - offy, 2 tx - offx) with zeros outside [0, Hs) x [0, Ws);
// forward: off = 0 on the caller's padded input, backward: off = 2 - pad (the Pad node's padding folded in, as in the direct
// input-gradient kernel).  Design notes and measurements: DESIGN.md section 4.2.1, profiles/r05_winograd_ab.txt.
// Summation order: per xi one fma chain over the reduction channels (two k per MFMA), then the fixed add trees of the output
// transform: deterministic; NOT the direct kernels' order - equal to them to contraction tolerance, exact on integer data.
#pragma once

struct WinoArgs {
    const float* src;   // (N, Ck, Hs, Ws)
    const float* u;     // transformed kernel in fragment order (wino_weights_kernel)
    float* dst;         // (N, Cm, Hd, Wd)
    const float* bias;  // optional, per output channel (forward)
    int N, Ck, Cm;      // Ck: reduction channels, Cm: output channels
    int Hs, Ws, Hd, Wd, offy, offx;
    int TY, TX;         // tiles per image: Hd / 2, Wd / 2
    long long P;        // N * TY * TX
    int nchunk;         // Ck / KC
    int assign;         // 1: dst = value, 0: dst += value
    int src_bytes, u_bytes, dst_bytes;  // the buffer descriptors' extents (all below 2^31)
    int stagger;        // unit of the staggered start in shader clocks (0: none)
    // division of a tile index by TY * TX and of the remainder by TX without the 40-instruction software divide (two per tile block
    // and thread in the patch offsets, two more in the output phase): Granlund - Montgomery multipliers, exact for every 32-bit dividend
    unsigned per_m, tx_m;
    int per_s1, per_s2, tx_s1, tx_s2;
};

// U in MFMA A-operand order.  For chunk ch (KC reduction channels), xi, block of 32 output channels cbt: a wave's fragment is
// KC/8 float4 per lane; lane = r + 32 h supplies output channel 32 cbt + r and reduction channels ch*KC + (KC/2) h + s,
// s = 0 .. KC/2 - 1, float4 j holding s = 4 j .. 4 j + 3:
//   u[((((ch * 16 + xi) * CBT + cbt) * (KC / 8) + j) * 64 + lane) * 4 + (s & 3)]
// flipped = 0: g = w[co][ci][ky][kx] (forward, w is (Cm, Ck, 3, 3)); flipped = 1: g = w[ci][co][2 - ky][2 - kx] (input gradient:
// w is (Ck, Cm, 3, 3), output channels of the pass are the convolution's input channels).
__global__ void wino_weights_kernel(float* __restrict__ u, const float* __restrict__ w, int Cm, int Ck, int KC, int flipped) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cm * Ck) return;
    const int co = idx / Ck, ci = idx % Ck;
    float g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            g[ky][kx] = flipped ? w[((long long)ci * Cm + co) * 9 + (2 - ky) * 3 + (2 - kx)] : w[((long long)co * Ck + ci) * 9 + ky * 3 + kx];
    float gg[4][3];  // G g
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gg[0][c] = g[0][c];
        gg[1][c] = 0.5f * ((g[0][c] + g[1][c]) + g[2][c]);
        gg[2][c] = 0.5f * ((g[0][c] - g[1][c]) + g[2][c]);
        gg[3][c] = g[2][c];
    }
    float uu[4][4];  // (G g) G^T
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uu[r][0] = gg[r][0];
        uu[r][1] = 0.5f * ((gg[r][0] + gg[r][1]) + gg[r][2]);
        uu[r][2] = 0.5f * ((gg[r][0] - gg[r][1]) + gg[r][2]);
        uu[r][3] = gg[r][2];
    }
    const int KH = KC / 2, CBT = Cm / 32;
    const int cbt = co / 32, r = co % 32, ch = ci / KC, kk = ci % KC, h = kk / KH, s = kk % KH, j = s / 4, tq = s % 4, lane = r + 32 * h;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
        u[((((long long)(ch * 16 + xi) * CBT + cbt) * (KC / 8) + j) * 64 + lane) * 4 + tq] = uu[xi >> 2][xi & 3];
}

// q = n / d for the divisor behind (m, s1, s2) = wino_magic(d)
__device__ __forceinline__ unsigned wino_div(unsigned n, unsigned m, int s1, int s2) {
    const unsigned t = __umulhi(m, n);
    return (t + ((n - t) >> s1)) >> s2;
}
inline void wino_magic(unsigned d, unsigned* m, int* s1, int* s2) {
    int l = 0;
    while ((1ull << l) < d) ++l;  // ceil(log2 d)
    *m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    *s1 = l < 1 ? l : 1;
    *s2 = l - 1 > 0 ? l - 1 : 0;
}

typedef unsigned wino_u2 __attribute__((ext_vector_type(2)));
typedef unsigned wino_u4 __attribute__((ext_vector_type(4)));

// One block per CU (128 KB of LDS, 512 registers per lane), PERSISTENT: block b walks the tile blocks b, b + gridDim.x, ... and, inside
// each, the chunks of KC reduction channels.  The stream of (tile block, chunk) items is software-pipelined through two V buffers:
// while the MFMAs of item i read V[i & 1], the same waves load and transform the patches of item i + 1 into V[(i + 1) & 1] and
// prefetch their U fragments three xi ahead from L2.  One barrier per item.  Only the first item's transform and each tile
// block's output transform run with the matrix pipe idle.
//
// Memory instructions carry no address arithmetic: everything goes through buffer descriptors (source, U, destination) with the
// per-lane part of the address computed once per tile block (voffset), the per-item / per-channel part in a scalar register
// (soffset) and the rest in the instruction's immediate.  A patch element outside the source (image border of the folded padding,
// tile past the end) has voffset 0x80000000 - out of the descriptor's range, so the hardware returns 0 for it (and drops such a
// store): no selects in the transform.  The host keeps this path to tensors below 2 GB.
//
// V layout: [xi][channel / 4][tile][channel % 4] - a thread of the transform owns one tile and FOUR consecutive channels (one
// float4 per patch element, the additions of the four channels side by side), stores one ds_write_b128 per xi, and an MFMA group
// (four k-steps of one xi) takes its four B values with one ds_read_b128.
// ODD (round 6): output extents that are not both even (7 x 7, 13 x 13 planes).  TY / TX = ceil(extent / 2); a border tile's second row /
// column does not exist: its row stores go to the out-of-range offset (dropped), its row pair is stored as ONE float instead of a
// float2 (two store instructions per row pair, each lane served by exactly one of them through the same out-of-range trick), and a
// float2 may start on any 4-byte boundary (rows of odd length).  The source side needs nothing: patch elements beyond the source are
// out of range already.  A separate instantiation - the even kernels' output phase, and with it their register allocation, is untouched.
template <int CB, int PB, int KC, int UR, bool ODD = false>
__global__ __launch_bounds__(64 * CB * PB, 1) void wino_kernel(WinoArgs a) {
    constexpr int PT = 32 * PB;        // tiles per tile block
    constexpr int KH = KC / 2;         // MFMA steps per item and xi (two reduction channels per step)
    constexpr int NJ = KC / 8;         // groups of four steps (one float4 of A, one of B per lane) per item and xi
    constexpr int KQ = KC / 4;         // channel quads per item
    constexpr int G = 16 * NJ;         // MFMA groups per item
    constexpr int VBUF = 16 * KC * PT; // floats of one V buffer
    constexpr int LG = G / 2;          // groups that carry the next item's patch loads (spread thin: 282 -> 267 / 309 -> 292 us at C3 against G / 4; G / 8: 295 / 341)
    constexpr int LPG = 64 / LG;       // ... loads per group
    constexpr int TG = G / 32;         // groups per part of the next item's transform (8 parts: 4 column passes, 4 row passes + stores)
    constexpr int T0 = G - 8 * TG - TG;  // first group of part 0
    static_assert((CB * PB == 4 || CB * PB == 2) && KQ * PT == 64 * CB * PB && KC % 8 == 0, "one (tile, channel quad) per thread and item");
    static_assert(16 % UR == 0 && UR >= 2, "the ring of U fragments (UR - 1 xi ahead of the MFMAs) keeps its phase from item to item");
    static_assert(LPG * LG == 64 && TG >= 1 && T0 >= LG, "the slices of the next item's transform fit the item's groups");
    static_assert(2 * VBUF * sizeof(float) + 32 * CB * sizeof(float) <= 160 * 1024, "two V buffers must fit the 160 KB of a gfx950 CU");
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];
    __shared__ float bias_s[32 * CB];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int cb = wid % CB, pb = wid / CB;
    const int c = lane & 31, h = lane >> 5;
    const int plane = a.Hs * a.Ws;
    const int per = a.TY * a.TX;
    const int npb = (int)((a.P + PT - 1) / PT);
    const int pl = t % PT, kq0 = t / PT;  // transform phase: this thread's tile of a tile block and its channel quad
    const int CBT = a.Cm / 32, cbg = blockIdx.y * CB + cb;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, a.u_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, a.dst_bytes, 0x00020000);
    const int ustep16 = CBT * NJ * 64 * 16;                 // bytes from xi to xi + 1
    const unsigned uvoff = (unsigned)(cbg * NJ * 64 + lane) * 16u;
    const int plane4 = plane * 4;

    if (t < 32 * CB) bias_s[t] = a.bias ? a.bias[32 * blockIdx.y * CB + t] : 0.f;

    // byte offset of patch element (i, j) of tile block `pbk`, channel 4 kq0 of the chunk, from channel 0 of the patch's sample;
    // 0x80000000 outside the source
    unsigned poff[16];
    auto patch = [&](int pbk) {
        const unsigned p = (unsigned)pbk * PT + pl;  // (the host keeps P below 2^30)
        const bool pvalid = pbk < npb && p < (unsigned)a.P;
        const unsigned pv = pvalid ? p : 0u;
        const unsigned n = wino_div(pv, a.per_m, a.per_s1, a.per_s2), rem = pv - n * (unsigned)per;
        const unsigned uty = wino_div(rem, a.tx_m, a.tx_s1, a.tx_s2);
        const int ty = (int)uty, tx = (int)(rem - uty * (unsigned)a.TX);
        const int r0 = 2 * ty - a.offy, c0 = 2 * tx - a.offx;
        const int sb = ((int)n * a.Ck + 4 * kq0) * plane + r0 * a.Ws + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = pvalid && (unsigned)(r0 + i) < (unsigned)a.Hs && (unsigned)(c0 + j) < (unsigned)a.Ws;
                poff[4 * i + j] = ok ? (unsigned)(sb + i * a.Ws + j) * 4u : 0x80000000u;
            }
    };
    // patch load number l of an item, in the order the column passes need them: l = 16 j + 4 i + q (column j, row i, channel q)
    float4 d[16];
    auto load = [&](int l, int ch) {
        const int j = l / 16, i = (l / 4) % 4, q = l % 4;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, poff[4 * i + j], (ch * KC + q) * plane4, 0));
        if (q == 0) d[4 * i + j].x = v;
        else if (q == 1) d[4 * i + j].y = v;
        else if (q == 2) d[4 * i + j].z = v;
        else d[4 * i + j].w = v;
    };
    // the eight parts of an item's transform, in place on d (four channels side by side): column pass j = k (k < 4): B^T d;
    // row pass i = k - 4 (k >= 4): (B^T d) B and the stores of xi = 4 i .. 4 i + 3
    auto part = [&](int k, float* vdst) {
        if (k < 4) {
            const int j = k;
            const float4 d0 = d[j], d1 = d[4 + j], d2 = d[8 + j], d3 = d[12 + j];
            d[j] = d0 - d2;
            d[4 + j] = d1 + d2;
            d[8 + j] = d2 - d1;
            d[12 + j] = d1 - d3;
        } else {
            const int i = k - 4;
            const float4 t0 = d[4 * i], t1 = d[4 * i + 1], t2 = d[4 * i + 2], t3 = d[4 * i + 3];
            float4* vp = reinterpret_cast<float4*>(vdst) + ((4 * i) * KQ + kq0) * PT + pl;
            vp[0 * KQ * PT] = t0 - t2;
            vp[1 * KQ * PT] = t1 + t2;
            vp[2 * KQ * PT] = t2 - t1;
            vp[3 * KQ * PT] = t1 - t3;
        }
    };

    // Blocks land on the eight XCDs round robin (block b on XCD b % 8), each XCD with its own L2: the tile blocks are dealt out in
    // eight contiguous ranges, and inside a range the XCD's blocks walk side by side - neighbouring tile blocks share two of
    // their four patch rows, the second reader finds them in its L2.
    const int nx = gridDim.x % 8 == 0 ? 8 : 1, step = gridDim.x / nx;
    const int range = (npb + nx - 1) / nx, lo = (int)(blockIdx.x % nx) * range, hi = lo + range < npb ? lo + range : npb;
    int pbk = lo + (int)(blockIdx.x / nx);
    if (pbk >= hi) return;
    // Blocks all take the same time per tile block, so the whole chip would reach its output phases together: bursts of stores
    // (16 MB at C3) that the next tile block's first U waits sit behind (loads and stores complete in order).  The tile blocks do
    // not divide evenly among the blocks; those with one tile block fewer than the longest walk start one, two or three units of
    // `stagger` shader clocks late - nobody finishes later than the longest walk, and the output phases are spread out.
    if (a.stagger > 0) {
        const int lb = (int)(blockIdx.x / nx), mine = (hi - lo - lb + step - 1) / step, most = (hi - lo + step - 1) / step;
        if (mine < most) {
            const long long until = (long long)a.stagger * (1 + lb % 3), t0 = (long long)__builtin_readcyclecounter();
            while ((long long)__builtin_readcyclecounter() - t0 < until) __builtin_amdgcn_s_sleep(64);
        }
    }
    // ---- prologue: item (pbk, chunk 0) into V[0], the matrix pipe idle
    patch(pbk);
#pragma unroll
    for (int l = 0; l < 64; ++l) load(l, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) part(k, V);
    // U fragments of the first UR - 1 xi of chunk 0: a ring of UR, UR - 1 xi ahead of the MFMAs (loads return in order: the wait
    // for a fragment is also a wait for every patch load issued before it - the distance is the latency the patch loads may have)
    float4 af[UR][NJ];
#pragma unroll
    for (int x = 0; x < UR - 1; ++x)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            af[x][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff + j * 1024, x * ustep16, 0));
    __syncthreads();

    int s = 0;
    nkmma::f32x16 acc[16];
    // one item: the MFMAs of (tile block, chunk ch) on V[s], with the slices of the item after it (chunk nch; `first`: the accumulators
    // start from zero - the first MFMA of each xi takes the constant)
    auto item = [&](auto first, int ch, int nch) {
        const float4* const vcur = reinterpret_cast<const float4*>(V + s * VBUF) + ((KH / 4) * h) * PT + 32 * pb + c;
        float* const vnext = V + (s ^ 1) * VBUF;
        float4 bf[2];
        bf[0] = vcur[0];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            // U fragments UR - 1 xi ahead (past the item's end: the next item's first ones)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                af[(xi + UR - 1) % UR][j] = __builtin_bit_cast(
                    float4, __builtin_amdgcn_raw_buffer_load_b128(
                                urs, uvoff + j * 1024, xi + UR - 1 < 16 ? (ch * 16 + xi + UR - 1) * ustep16 : (nch * 16 + xi + UR - 1 - 16) * ustep16, 0));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int g = xi * NJ + j;  // group number inside the item: its B values are in bf[g & 1]
                if (g + 1 < G) bf[(g + 1) & 1] = vcur[(((g + 1) / NJ) * KQ + (g + 1) % NJ) * PT];
                // ---- this group's slice of the NEXT item's transform
                if (g < LG) {
#pragma unroll
                    for (int l = 0; l < LPG; ++l) load(g * LPG + l, nch);
                }
                if (g >= T0 && g < T0 + 8 * TG && (g - T0) % TG == 0) part((g - T0) / TG, vnext);
                const float4 av = af[xi % UR][j];
                const float4 bv = bf[g & 1];
                if (decltype(first)::value && j == 0) {
                    const nkmma::f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, zero, 0, 0, 0);
                } else {
                    acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[xi], 0, 0, 0);
                }
                acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[xi], 0, 0, 0);
                acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[xi], 0, 0, 0);
                acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[xi], 0, 0, 0);
                // the order inside the group: an MFMA, then a share of everything else the group carries (the wave would only
                // wait for the matrix pipe there: 64 cycles per MFMA) - four times
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // LDS read (the next group's B values)
                    __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);  // buffer loads (patches, U fragments)
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // VALU
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // LDS stores
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // V[s] has been read by everybody, V[s ^ 1] is complete
        s ^= 1;
    };

    for (; pbk < hi; pbk += step) {
        // the item after a tile block's last chunk is chunk 0 of this block's next tile block (none: every load is out of range
        // and the values land in a V buffer nobody reads)
        const int pnext = pbk + step < hi ? pbk + step : npb;
        if (a.nchunk == 1) patch(pnext);
        item(std::true_type{}, 0, a.nchunk == 1 ? 0 : 1);
        for (int ch = 1; ch < a.nchunk; ++ch) {
            const bool last_ch = ch + 1 == a.nchunk;
            if (last_ch) patch(pnext);
            item(std::false_type{}, ch, last_ch ? 0 : ch + 1);
        }

        // ---- output transform on the wave's own registers: lane (c, h) owns tile 32 pb + c and 16 channels (MFMA C layout):
        // e = 4 Q + el is channel 32 cbg + 8 Q + 4 h + el.  No loads after the first store in the assigning form (loads and stores
        // share one in-order counter: a load behind a store waits for every store before it).  What is left of that in the ISA: the
        // run-time `if (!a.assign)` makes the compiler wait for everything outstanding at each quarter's join (three drains of eight
        // stores per tile block).  Both ways around it were built and are SLOWER, because they move the register allocation of a
        // kernel that sits at its 512-register limit: `assign` as a template parameter (69 spilled registers instead of 54: forward
        // 301 -> 341 us, input gradient 309 -> 314), two copies of the phase under one branch (108 spills: 343 / 329).
        const unsigned p = (unsigned)pbk * PT + 32 * pb + c;
        const bool pvalid = p < (unsigned)a.P;
        const unsigned pv = pvalid ? p : 0u;
        const unsigned un = wino_div(pv, a.per_m, a.per_s1, a.per_s2), urem = pv - un * (unsigned)per;
        const unsigned uty = wino_div(urem, a.tx_m, a.tx_s1, a.tx_s2);
        const int n = (int)un, ty = (int)uty, tx = (int)(urem - uty * (unsigned)a.TX);
        const int oplane = a.Hd * a.Wd;
        const unsigned ovoff = pvalid ? (unsigned)((n * a.Cm + 32 * cbg + 4 * h) * oplane + 2 * ty * a.Wd + 2 * tx) * 4u : 0x80000000u;
        // ODD: per row r the offset of its float2 store (lanes whose tile has both columns) and of its float store (lanes whose tile has
        // only the first); a row that does not exist has neither
        unsigned ov2[2], ov1[2];
        if constexpr (ODD) {
            const bool two_cols = 2 * tx + 1 < a.Wd;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool row_ok = pvalid && 2 * ty + r < a.Hd;
                const unsigned o = ovoff + (unsigned)(r * a.Wd) * 4u;
                ov2[r] = row_ok && two_cols ? o : 0x80000000u;
                ov1[r] = row_ok && !two_cols ? o : 0x80000000u;
            }
        }
#pragma unroll
        for (int Q = 0; Q < 4; ++Q) {
            float2 y[4][2];
#pragma unroll
            for (int el = 0; el < 4; ++el) {
                const int e = 4 * Q + el;
                float tm[2][4];  // A^T M
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tm[0][j] = (acc[0 + j][e] + acc[4 + j][e]) + acc[8 + j][e];
                    tm[1][j] = (acc[4 + j][e] - acc[8 + j][e]) - acc[12 + j][e];
                }
                const float bv = bias_s[32 * cb + 8 * Q + 4 * h + el];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    y[el][r].x = ((tm[r][0] + tm[r][1]) + tm[r][2]) + bv;
                    y[el][r].y = ((tm[r][1] - tm[r][2]) - tm[r][3]) + bv;
                }
            }
            if (!a.assign) {
                float2 old[4][2];
#pragma unroll
                for (int el = 0; el < 4; ++el)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if constexpr (ODD)  // (the second element of a one-column tile is read - the next row's first, or 0 past the end - and not stored)
                            old[el][r] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(drs, ov2[r] < ov1[r] ? ov2[r] : ov1[r],  // the one in range, if any
                                                                                                        (8 * Q + el) * oplane * 4, 0));
                        else
                            old[el][r] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(drs, ovoff, ((8 * Q + el) * oplane + r * a.Wd) * 4, 0));
                    }
#pragma unroll
                for (int el = 0; el < 4; ++el)
#pragma unroll
                    for (int r = 0; r < 2; ++r) { y[el][r].x += old[el][r].x; y[el][r].y += old[el][r].y; }
            }
#pragma unroll
            for (int el = 0; el < 4; ++el)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if constexpr (ODD) {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wino_u2, y[el][r]), drs, ov2[r], (8 * Q + el) * oplane * 4, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[el][r].x), drs, ov1[r], (8 * Q + el) * oplane * 4, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wino_u2, y[el][r]), drs, ovoff, ((8 * Q + el) * oplane + r * a.Wd) * 4, 0);
                    }
                }
        }
    }
}

// Host side.  `taken` = false: not a case for this path (the caller goes on to the implicit-GEMM kernels).
//   fwd:  src = x (N, Ck = Cin, Hs, Ws),  dst = y (N, Cm = Cout, Hs - 2, Ws - 2), off = 0
//   bwd:  src = gy (N, Ck = Cout, Hs, Ws), dst = dx (N, Cm = Cin, Hd, Wd), off = 2 - pad per axis, Hs = Hd + 2 pad - 2
// `force`: whenever the shape allows, whatever the knob and the block-count rule say (the entry points with the module's padding
// folded in have no other kernel to fall back on); `dry`: decide only.
int wino_launch(nk_device* dev, bool bwd, const float* src, const float* w, float* dst, const float* bias, int N, int Ck, int Cm, int Hs,
                int Ws, int Hd, int Wd, int offy, int offx, int assign, double flop, bool* taken, bool force = false, bool dry = false) {
    *taken = false;
    const int mode = force ? 1 : dev->tune_conv_winograd;  // -1 rule, 0 never, 1 whenever the shape allows
    if (mode == 0) return NK_OK;
    // Two block shapes, either pass: WIDE = four waves, 128 output channels, reduction chunks of 32 (one block per CU); NARROW = two
    // waves, 64 output channels, chunks of 16 (two independent blocks per CU: a barrier joins two waves instead of four, the
    // transform work per MFMA doubles).  Wide wherever the channel counts allow it - measured (benchmarks/ab_winograd.py 128 shape,
    // profiles/r05_winograd_ab.txt): 309 / 313 us at C3's forward, 163 / 169 and 162 / 171 at 128 -> 128 channels on 28 x 28,
    // 157 / 172 and 157 / 168 at 256 -> 256 on 14 x 14 (implicit GEMM: 494, 306, 327, 317, 323) - narrow for 64 output channels
    // (C3's input gradient: 319 us against 507).
    const bool wide_ok = Cm % 128 == 0 && Ck % 32 == 0, narrow_ok = Cm % 64 == 0 && Ck % 16 == 0;
    const int shape = dev->tune_conv_wino_shape;  // -1 rule, 0 narrow, 1 wide
    const bool wide = shape < 0 ? wide_ok : (shape == 1 ? wide_ok : !narrow_ok && wide_ok);
    if (!wide && !narrow_ok) return NK_OK;
    const int KC = wide ? 32 : 16, CM = wide ? 128 : 64, PT = 32, NW = wide ? 4 : 2;  // reduction chunk, channels / tiles / waves per block
    if (Hd < 2 || Wd < 2) return NK_OK;
    const bool odd = Hd % 2 != 0 || Wd % 2 != 0;  // a border tile row / column is half empty: the ODD instantiation
    if (!al16(dst) || !al16(src)) return NK_OK;
    const int TYc = (Hd + 1) / 2, TXc = (Wd + 1) / 2;
    const long long P = (long long)N * TYc * TXc;
    const long long blocks = (P + PT - 1) / PT * (Cm / CM);
    // buffer descriptors with 32-bit byte offsets, 0x80000000 as the out-of-range mark: every tensor below 2 GB
    const long long src_bytes = (long long)N * Ck * Hs * Ws * 4, dst_bytes = (long long)N * Cm * Hd * Wd * 4, u_bytes = 16LL * Cm * Ck * 4;
    if (P >= (1LL << 30) || src_bytes >= 0x7fffffffLL || dst_bytes >= 0x7fffffffLL || u_bytes >= 0x7fffffffLL) return NK_OK;
    // by rule: from an eighth of the CUs' worth of wide blocks, one CU's worth of narrow ones.  Measured at C3's geometry with small
    // batches (benchmarks/ab_winograd.py N; profiles/r05_winograd_ab.txt section 7): forward (wide) 49 / 52 / 84 / 161 us -> 27 / 29 / 46 /
    // 82 at N = 4 / 8 / 16 / 32 (98 blocks at N = 4), input gradient (narrow) 42 / 63 / 91 / 146 -> 43 / 44 / 53 / 95 (196 blocks at N = 4: a
    // draw).  (The first rule, four rounds of the CUs, dated from the first version of the kernels.)
    if (mode < 0 && blocks < (wide ? dev->num_cus / 8 : dev->num_cus)) return NK_OK;
    if (dry) { *taken = true; return NK_OK; }
    void* ws = nullptr;
    int rc = nk_workspace(dev, (size_t)16 * Cm * Ck * sizeof(float), &ws);
    if (rc) return rc;
    rc = nk_prof_start(dev, NK_KERNEL_CONV, flop);
    if (rc) return rc;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((Cm * Ck + 255) / 256)), dim3(256), 0, dev->compute, (float*)ws, w, Cm, Ck, KC,
                       bwd ? 1 : 0);
    NK_LAUNCH_CHECK();
    WinoArgs a{};
    a.src = src; a.u = (const float*)ws; a.dst = dst; a.bias = bias;
    a.N = N; a.Ck = Ck; a.Cm = Cm;
    a.Hs = Hs; a.Ws = Ws; a.Hd = Hd; a.Wd = Wd; a.offy = offy; a.offx = offx;
    a.TY = TYc; a.TX = TXc; a.P = P;
    a.nchunk = Ck / KC; a.assign = assign;
    // measured at C3 (benchmarks/ab_winograd.py 128 stagger): forward 315 / 313 / 307 / 311 / 313 us at 0 / 4000 / 6000 / 12000 / 16000 clocks,
    // input gradient 321 / 314 / 317 / 325 / 333 - a small offset is all it takes, larger ones only delay the late starters
    a.stagger = dev->tune_conv_wino_stagger < 0 ? 5000 : dev->tune_conv_wino_stagger;
    wino_magic((unsigned)(a.TY * a.TX), &a.per_m, &a.per_s1, &a.per_s2);
    wino_magic((unsigned)a.TX, &a.tx_m, &a.tx_s1, &a.tx_s2);
    a.src_bytes = (int)src_bytes; a.u_bytes = (int)u_bytes; a.dst_bytes = (int)dst_bytes;
    // persistent blocks, one per CU: block b walks the tile blocks b, b + grid.x, ...
    const long long npb = (P + PT - 1) / PT;
    const long long slots = (long long)dev->num_cus * (4 / NW);  // one wave per SIMD
    const long long per_group = slots / (Cm / CM) > 0 ? slots / (Cm / CM) : 1;
    const dim3 grid((unsigned)(npb < per_group ? npb : per_group), (unsigned)(Cm / CM));
    // U ring: 2 xi for the wide blocks (one xi = four MFMA groups ahead: 2 spilled registers and 283 us at C3's forward; a ring of 4
    // - twelve groups ahead - costs 54 spills and 301 us), 4 xi for the narrow ones (one xi there is two groups: a ring of 2 gives 357 us
    // against 309)
    if (wide && odd) hipLaunchKernelGGL((wino_kernel<4, 1, 32, 2, true>), grid, dim3(256), 0, dev->compute, a);
    else if (wide) hipLaunchKernelGGL((wino_kernel<4, 1, 32, 2>), grid, dim3(256), 0, dev->compute, a);
    else if (odd) hipLaunchKernelGGL((wino_kernel<2, 1, 16, 4, true>), grid, dim3(128), 0, dev->compute, a);
    else hipLaunchKernelGGL((wino_kernel<2, 1, 16, 4>), grid, dim3(128), 0, dev->compute, a);
    NK_LAUNCH_CHECK();
    *taken = true;
    ++dev->wino_launches;
    return nk_prof_stop(dev);
}
