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
// which run on the MFMA core (v_mfma_f32_32x32x2_f32).  Everything stays on chip: a block transforms the patches of 32*PB
// tiles into LDS (V, two buffers of 64 KB), each wave holds the 16 accumulator tiles (xi) of its 32 output channels x 32 tiles
// in registers (256 of the 512 a wave has at one wave per SIMD), takes its U fragments straight from L2 in MFMA operand order
// (the kernel transform writes them that way), and applies the output transform to its own registers - V, M never touch HBM.
//   forward        CB = 4 waves x 32 channels = 128 output channels, 32 tiles, the input channels in chunks of 32
//   input gradient CB = 2 x 32 = 64 channels of dX, PB = 2 x 32 tiles, the gradient's channels in chunks of 16
// One source for both: the "source" tensor is the input (forward) or the output gradient (backward), read at patch origin
// (2 ty - offy, 2 tx - offx) with zeros outside [0, Hs) x [0, Ws); forward: off = 0 on the caller's padded input, backward:
// off = 2 - pad (the Pad node's padding folded in, as in the direct input-gradient kernel).
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

// One block per CU (128 KB of LDS, 512 registers per lane), PERSISTENT: block b walks the tile blocks b, b + gridDim.x, ... and, inside
// each, the chunks of KC reduction channels.  The stream of (tile block, chunk) items is software-pipelined through two V buffers:
// while the MFMAs of item i read V[i & 1], the same waves load and transform the patches of item i + 1 into V[(i + 1) & 1] - the loads
// in the first quarter of the item's MFMA groups, the additions and LDS stores of one channel at a time further on, a few
// instructions per group of four MFMAs - and prefetch their U fragments three xi ahead from L2.  One barrier per item.  Only the
// first item's transform and each tile block's output transform run with the matrix pipe idle.
template <int CB, int PB, int KC>
__global__ __launch_bounds__(256, 1) void wino_kernel(WinoArgs a) {
    constexpr int PT = 32 * PB;        // tiles per tile block
    constexpr int KH = KC / 2;         // MFMA steps per item and xi (two reduction channels per step)
    constexpr int NJ = KC / 8;         // groups of four steps (one float4 of A per lane) per item and xi
    constexpr int CSTEP = 256 / PT;    // channels the block's threads cover per transform pass
    constexpr int NQ = KC / CSTEP;     // channels per thread and item
    constexpr int G = 16 * NJ;         // MFMA groups per item
    constexpr int VBUF = 16 * KC * PT; // floats of one V buffer: [xi][channel in chunk][tile]
    constexpr int LPG = 64 / (G / 4);  // patch loads per group while the loads are issued (groups 0 .. G/4 - 1)
    constexpr int TG = G / 8;          // groups one channel's transform is spread over
    constexpr int T0 = G / 4 + G / 8;  // first group of channel 0's transform (its loads left at least G/8 groups earlier)
    constexpr int PPG = 8 / TG;        // transform parts (of 8 per channel: 4 column passes, 4 row passes + stores) per group
    static_assert(CB * PB == 4 && NQ == 4 && KC % 8 == 0 && 16 % 4 == 0, "four waves, four channels per thread and item");
    static_assert(T0 + NQ * TG <= G && LPG * (G / 4) == 16 * NQ && PPG * TG == 8, "the slices of the next item's transform fit the item's groups");
    static_assert(2 * VBUF * sizeof(float) <= 160 * 1024, "two V buffers must fit the 160 KB of a gfx950 CU");
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int cb = wid % CB, pb = wid / CB;
    const int c = lane & 31, h = lane >> 5;
    const int plane = a.Hs * a.Ws;  // (the host checks that the tensors have fewer than 2^31 elements)
    const int per = a.TY * a.TX;
    const int npb = (int)((a.P + PT - 1) / PT);
    const int pl = t % PT, cl0 = t / PT;  // transform phase: this thread's tile of a tile block and its first channel
    const int CBT = a.Cm / 32, cbg = blockIdx.y * CB + cb;
    const long long ustep = (long long)CBT * NJ * 64;  // float4s from xi to xi + 1
    const float4* const ubase = reinterpret_cast<const float4*>(a.u) + (long long)cbg * NJ * 64 + lane;  // + ch * 16 * ustep + xi * ustep + j * 64

    // offset of patch element (i, j) of tile block `pbk` from channel 0 of the patch's sample, -1 outside the source (loads are
    // branch-free: such an element reads element 0 and is replaced by 0)
    int poff[16];
    auto patch = [&](int pbk) {
        const long long p = (long long)pbk * PT + pl;
        const bool pvalid = pbk < npb && p < a.P;
        const int n = pvalid ? (int)(p / per) : 0, rem = pvalid ? (int)(p % per) : 0;
        const int ty = rem / a.TX, tx = rem % a.TX;
        const int r0 = 2 * ty - a.offy, c0 = 2 * tx - a.offx;
        const int sb = n * a.Ck * plane + r0 * a.Ws + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = pvalid && (unsigned)(r0 + i) < (unsigned)a.Hs && (unsigned)(c0 + j) < (unsigned)a.Ws;
                poff[4 * i + j] = ok ? sb + i * a.Ws + j : -1;
            }
    };
    // the pieces of one channel's transform (d: the 16 loaded values, tt: B^T d): column pass j, then row pass i + stores
    auto col_pass = [&](const float (&d)[16], float (&tt)[16], int j) {
        const float d0 = poff[j] >= 0 ? d[j] : 0.f, d1 = poff[4 + j] >= 0 ? d[4 + j] : 0.f;
        const float d2 = poff[8 + j] >= 0 ? d[8 + j] : 0.f, d3 = poff[12 + j] >= 0 ? d[12 + j] : 0.f;
        tt[j] = d0 - d2;
        tt[4 + j] = d1 + d2;
        tt[8 + j] = d2 - d1;
        tt[12 + j] = d1 - d3;
    };
    auto row_pass = [&](const float (&tt)[16], float* vdst, int cl, int i) {
        float* vp = vdst + ((4 * i) * KC + cl) * PT + pl;
        vp[0 * KC * PT] = tt[4 * i] - tt[4 * i + 2];
        vp[1 * KC * PT] = tt[4 * i + 1] + tt[4 * i + 2];
        vp[2 * KC * PT] = tt[4 * i + 2] - tt[4 * i + 1];
        vp[3 * KC * PT] = tt[4 * i + 1] - tt[4 * i + 3];
    };

    int pbk = blockIdx.x;
    if (pbk >= npb) return;
    // ---- prologue: item (pbk, chunk 0) into V[0], the matrix pipe idle
    patch(pbk);
    {
        float d[NQ][16];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int coff = (cl0 + q * CSTEP) * plane;
#pragma unroll
            for (int e = 0; e < 16; ++e) d[q][e] = a.src[poff[e] >= 0 ? poff[e] + coff : 0];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float tt[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) col_pass(d[q], tt, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) row_pass(tt, V, cl0 + q * CSTEP, i);
        }
    }
    // U fragments of xi = 0, 1, 2 of chunk 0: a ring of four, three xi ahead of the MFMAs
    float4 af[4][NJ];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int j = 0; j < NJ; ++j) af[x][j] = ubase[x * ustep + j * 64];
    __syncthreads();

    int s = 0;
    for (; pbk < npb; pbk += gridDim.x) {
        nkmma::f32x16 acc[16];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[xi][e] = 0.f;
        for (int ch = 0; ch < a.nchunk; ++ch) {
            // the item after this one: the next chunk, or chunk 0 of this block's next tile block (none: its loads read element 0
            // and the values land in a V buffer nobody reads)
            const bool last_ch = ch + 1 == a.nchunk;
            const int nch = last_ch ? 0 : ch + 1;
            if (last_ch) patch(pbk + gridDim.x);
            const int ncoff = (nch * KC + cl0) * plane;
            const float* const vcur = V + s * VBUF + (KH * h) * PT + 32 * pb + c;
            float* const vnext = V + (s ^ 1) * VBUF;
            const float4* const ucur = ubase + (long long)ch * 16 * ustep;
            const float4* const unext = ubase + (long long)nch * 16 * ustep;
            float d[NQ][16], tt[16];
            float bf[2][4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) bf[0][q4] = vcur[q4 * PT];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                // U fragments three xi ahead (past the item's end: the next item's first three)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    af[(xi + 3) & 3][j] = xi + 3 < 16 ? ucur[(xi + 3) * ustep + j * 64] : unext[(xi + 3 - 16) * ustep + j * 64];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int g = xi * NJ + j;  // group number inside the item: its B values are in bf[g & 1]
                    if (g + 1 < G) {
                        const int xn = (g + 1) / NJ, jn = (g + 1) % NJ;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) bf[(g + 1) & 1][q4] = vcur[(xn * KC + 4 * jn + q4) * PT];
                    }
                    // ---- this group's slice of the NEXT item's transform
                    if (g < G / 4) {  // patch loads: LPG per group
#pragma unroll
                        for (int l = 0; l < LPG; ++l) {
                            const int q = (g * LPG + l) / 16, e = (g * LPG + l) % 16;
                            d[q][e] = a.src[poff[e] >= 0 ? poff[e] + ncoff + q * CSTEP * plane : 0];
                        }
                    }
                    if (g >= T0 && g < T0 + NQ * TG) {  // channel q's transform, PPG of its 8 parts per group
                        const int q = (g - T0) / TG, k0 = ((g - T0) % TG) * PPG;
#pragma unroll
                        for (int k = k0; k < k0 + PPG; ++k) {
                            if (k < 4) col_pass(d[q], tt, k);
                            else row_pass(tt, vnext, cl0 + q * CSTEP, k - 4);
                        }
                    }
                    const float4 av = af[xi & 3][j];
                    acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[g & 1][0], acc[xi], 0, 0, 0);
                    acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[g & 1][1], acc[xi], 0, 0, 0);
                    acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[g & 1][2], acc[xi], 0, 0, 0);
                    acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[g & 1][3], acc[xi], 0, 0, 0);
                    // the order inside the group: an MFMA, then a share of everything else the group carries (the wave would only
                    // wait for the matrix pipe there: 64 cycles per MFMA) - four times
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // LDS reads (the next group's B values)
                        __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);  // global loads (patches, U fragments)
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // VALU
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // LDS stores
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();  // V[s] has been read by everybody, V[s ^ 1] is complete
            s ^= 1;
        }

        // ---- output transform on the wave's own registers: lane (c, h) owns tile 32 pb + c and 16 channels (MFMA C layout)
        const long long p = (long long)pbk * PT + 32 * pb + c;
        if (p < a.P) {
            const int n = (int)(p / per), rem = (int)(p % per), ty = rem / a.TX, tx = rem % a.TX;
            const long long oplane = (long long)a.Hd * a.Wd;
            float* const obase = a.dst + (long long)n * a.Cm * oplane + (long long)(2 * ty) * a.Wd + 2 * tx;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = 32 * cbg + (e & 3) + 8 * (e >> 2) + 4 * h;
                float tm[2][4];  // A^T M
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tm[0][j] = (acc[0 + j][e] + acc[4 + j][e]) + acc[8 + j][e];
                    tm[1][j] = (acc[4 + j][e] - acc[8 + j][e]) - acc[12 + j][e];
                }
                const float bv = a.bias ? a.bias[co] : 0.f;
                float* o = obase + (long long)co * oplane;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    float y0 = (tm[r][0] + tm[r][1]) + tm[r][2];
                    float y1 = (tm[r][1] - tm[r][2]) - tm[r][3];
                    if (a.bias) { y0 += bv; y1 += bv; }
                    float2* q = reinterpret_cast<float2*>(o + r * a.Wd);
                    if (!a.assign) { const float2 old = *q; y0 += old.x; y1 += old.y; }
                    *q = make_float2(y0, y1);
                }
            }
        }
    }
}

// Host side.  `taken` = false: not a case for this path (the caller goes on to the implicit-GEMM kernels).
//   fwd:  src = x (N, Ck = Cin, Hs, Ws),  dst = y (N, Cm = Cout, Hs - 2, Ws - 2), off = 0
//   bwd:  src = gy (N, Ck = Cout, Hs, Ws), dst = dx (N, Cm = Cin, Hd, Wd), off = 2 - pad per axis, Hs = Hd + 2 pad - 2
int wino_launch(nk_device* dev, bool bwd, const float* src, const float* w, float* dst, const float* bias, int N, int Ck, int Cm, int Hs,
                int Ws, int Hd, int Wd, int offy, int offx, int assign, double flop, bool* taken) {
    *taken = false;
    const int mode = dev->tune_conv_winograd;  // -1 rule, 0 never, 1 whenever the shape allows
    if (mode == 0) return NK_OK;
    const int KC = bwd ? 16 : 32, CM = bwd ? 64 : 128, PT = bwd ? 64 : 32;
    if (Hd < 2 || Wd < 2 || Hd % 2 != 0 || Wd % 2 != 0 || Ck % KC != 0 || Cm % CM != 0) return NK_OK;
    if (!al16(dst) || !al16(src)) return NK_OK;
    const long long P = (long long)N * (Hd / 2) * (Wd / 2);
    const long long blocks = (P + PT - 1) / PT * (Cm / CM);
    if (blocks > 0x7fffffffLL || (long long)N * Ck * Hs * Ws >= 0x7fffffffLL || (long long)N * Cm * Hd * Wd >= 0x7fffffffLL) return NK_OK;
    // by rule: enough blocks for four rounds of the chip's CUs (one block per CU: 128 KB of LDS, 512 registers per lane)
    if (mode < 0 && blocks < 4LL * dev->num_cus) return NK_OK;
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
    a.TY = Hd / 2; a.TX = Wd / 2; a.P = P;
    a.nchunk = Ck / KC; a.assign = assign;
    // persistent blocks, one per CU: block b walks the tile blocks b, b + grid.x, ...
    const long long npb = (P + PT - 1) / PT;
    const long long per_group = dev->num_cus / (Cm / CM) > 0 ? dev->num_cus / (Cm / CM) : 1;
    const dim3 grid((unsigned)(npb < per_group ? npb : per_group), (unsigned)(Cm / CM));
    if (bwd) hipLaunchKernelGGL((wino_kernel<2, 2, 16>), grid, dim3(256), 0, dev->compute, a);
    else hipLaunchKernelGGL((wino_kernel<4, 1, 32>), grid, dim3(256), 0, dev->compute, a);
    NK_LAUNCH_CHECK();
    *taken = true;
    return nk_prof_stop(dev);
}
