// Winograd F(3x3, 2x2) for the KERNEL gradient of the 3x3 / stride 1 / dilation 1 / groups 1 convolution
// (ConvolutionBackwardKernel::backward, node/convolution/mod.rs:191-228: dW[co][ci][ky][kx] = sum over samples and output
// positions of dY[n][co][y][x] * X[n][ci][y + ky][x + kx]) - part of the convolution translation unit (included by nk_conv.hip
// inside its anonymous namespace, after nk_conv_winograd.h).
//
// Per 2x2 tile of dY and the 4x4 patch of X under it the nine sums are a correlation with a 2x2 "kernel": 16 multiplies instead of 36,
//   dW (3x3) += A^T [ (G dy G^T) . (B^T x B) ] A,   A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1],  G = [1 0; 1/2 1/2; 1/2 -1/2; 0 1],
//                                                   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1]
// (the transposed form of the forward's F(2x2, 3x3): same points 0, 1, -1, inf; matrices checked against the direct sums in
// tests/test_gpu_winograd.py).  Both transforms are linear, so the sum over the tiles is taken INSIDE:
//   M[xi] (Co x Ci) = sum over tiles p of DY^[xi][co][p] * X^[xi][p][ci],   xi = 0..15,   dW[co][ci] = A^T M[.][co][ci] A,
// 16 GEMMs with the tiles as the reduction dimension - 26.3 GFLOP at C3 instead of 59.2 - on v_mfma_f32_32x32x2_f32 (two tiles per
// instruction).  Unlike the forward, BOTH operands are transformed on the fly and nothing is pre-transformed in HBM.
//
// Block = four waves = a 64 x 64 block of (co, ci) for all 16 xi: wave (wr, wc) holds the 16 accumulator tiles of 32 co x 32 ci in
// registers (256 of its 512).  grid.y = the (co block, ci block) pairs, grid.x = slices of the tile sequence, one block per CU.
// An ITEM is 8 tiles: every thread transforms the patch of one tile for TWO input channels and the dY tile for TWO output channels
// (float2 = the channel pair) and stores the 16 + 16 xi values with ds_write_b64 into the LDS images
//   X^ / DY^ : [xi][channel / 4][tile (8)][channel % 4] + 4 floats of padding per channel quad (36 floats: a wave's 32 lanes = 32
//   channels read 32 distinct banks),
// two images of each (2 x 72 KB of the 160 KB); the MFMA waves read one A and one B value per instruction (ds_read_b32).  Loads run
// TWO items ahead of the MFMAs (registers), the transform one item ahead (LDS): an item is only 64 MFMAs per wave = 4096 clocks,
// less than an HBM round trip under load.  One barrier per item.
// Every slice block leaves its 16 x 64 x 64 sums (256 KB) in a workspace slab; a second kernel adds the slabs in slice order, applies
// A^T . A and writes (or adds to) dW; the bias gradient (sum of dY per output channel) is summed on the way by the threads that load
// dY and leaves through the same second kernel.  Deterministic: one fma chain per (xi, co, ci) and slice in tile order, slices in
// order, fixed add trees - not the implicit-GEMM kernel's order; equal to it to contraction tolerance, exact on integer data.
#pragma once

struct WinoDwArgs {
    const float* x;    // (N, Ci, Hs, Ws): the convolution's (padded) input
    const float* gy;   // (N, Co, Hd, Wd), Hd = Hs - 2, Wd = Ws - 2
    float* slabs;      // [slice][pair][xi 16][co 64][ci 64]
    float* bslabs;     // [slice][Co] or null
    int N, Ci, Co, Hs, Ws, Hd, Wd, TY, TX;
    unsigned P;        // N * TY * TX tiles
    int items;         // items of 8 tiles per slice (even)
    int cib;           // ci blocks (Ci / 64): pair = cob * cib + cb
    unsigned per_m, tx_m;
    int per_s1, per_s2, tx_s1, tx_s2;
    int x_bytes, gy_bytes;
    // FOLD instantiation (the Conv module's Zero padding folded in: `x` is the UNPADDED input of extents Hx x Wx, the patch of tile
    // (ty, tx) starts at row 2 ty - pady, column 2 tx - padx, zeros outside; pady, padx in {0, 1}); Hs, Ws stay the padded extents
    int Hx, Wx, pady, padx;
};

constexpr int WDW_T = 8;                       // tiles per item
constexpr int WDW_CQ = 36;                     // floats per (xi, channel quad): 8 tiles x 4 channels + 4 of padding
constexpr int WDW_IMG = 16 * 16 * WDW_CQ;      // floats of one operand image: 16 xi x 16 channel quads (64 channels)

// ODD (round 6): output extents that are not both even; TY / TX = ceil(extent / 2).  A border tile's second dY row / column does not
// exist and must contribute nothing: the row is loaded from the out-of-range offset (zeros), the column is zeroed with one select per
// row - and with it the patch elements only that row / column would have met (patch row 3, patch column 3; with folded padding of 1
// also patch column 2 lies beyond the image there): they belong to the next row / channel or lie past the tensor, and a non-finite
// value among them must not reach a tile that does not see it.  Loads may start on any 4-byte boundary (rows of odd length).  Separate
// instantiations: the even kernels are untouched.
template <bool FOLD, bool ODD = false>
__global__ __launch_bounds__(256, 1) void wino_dw_kernel(WinoDwArgs a) {
    __shared__ __attribute__((aligned(16))) float XS[2 * WDW_IMG];
    __shared__ __attribute__((aligned(16))) float YS[2 * WDW_IMG];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wr = wid >> 1, wc = wid & 1;       // the wave's co block / ci block inside the 64 x 64 block
    const int c = lane & 31, h = lane >> 5;
    const int pair = blockIdx.y, cob = pair / a.cib, cb = pair % a.cib;
    const int tl = t & 7, cp = t >> 3;           // transform phase: this thread's tile of an item and its channel pair (0..31)
    const int plane = FOLD ? a.Hx * a.Wx : a.Hs * a.Ws, oplane = a.Hd * a.Wd;  // plane: of the tensor `x` points to
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, 0, a.gy_bytes, 0x00020000);
    // scalar offsets of the twelve loads of an item, made PROVABLY wave-uniform once (readfirstlane): left as expressions of the
    // kernel arguments the compiler kept some of them in vector registers and wrapped those loads in waterfall loops
    int xso[2][4], yso[2][2];
    const unsigned plane4 = (unsigned)__builtin_amdgcn_readfirstlane(plane * 4);
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xso[ch][i] = FOLD ? 0 : __builtin_amdgcn_readfirstlane((ch * plane + i * a.Ws) * 4);
#pragma unroll
        for (int r = 0; r < 2; ++r) yso[ch][r] = __builtin_amdgcn_readfirstlane((ch * oplane + r * a.Wd) * 4);
    }

    // loaded, not yet transformed operands of one item: two register sets (loads run two items ahead)
    wino_u2 xr[2][2][4][2];  // [set][channel of the pair][patch row][columns 0-1 / 2-3]
    wino_u2 yr[2][2][2];     // [set][channel of the pair][tile row]
    float2 bsum = make_float2(0.f, 0.f);  // bias gradient of this thread's two output channels over its tiles
    bool fix_col[2] = {false, false};   // ODD: the tile of the item in each register set has one dY column (the last column of tiles)
    bool fix_x2[2] = {false, false};    // ODD + FOLD: ... and its patch column 2 lies beyond the image as well (padx = 1)
    bool fix_left[2] = {false, false}, fix_right[2] = {false, false};  // FOLD: border flags of the item in each register set
    int fix_shift[2] = {-1, -1};                                        // ... and the patch row loaded from the tensor's first byte (-1: none)

    const unsigned tile0 = (unsigned)blockIdx.x * (unsigned)a.items * WDW_T;
    // the twelve loads of an item (8 patch rows of 16 bytes, 4 tile rows of 8), issued ONE AT A TIME between MFMA groups: as a burst
    // at the top of the item the four waves' 48 instructions fill the address unit's queue and every wave waits for its turn with the
    // matrix pipe idle - 1.1 us of a 3.1 us item
    unsigned xo = 0x80000000u, yo = 0x80000000u;
    unsigned yo1 = 0x80000000u, xo3 = 0x80000000u;  // ODD: dY row 1 and (unfolded) patch row 3 - out of range when the tile has one row
    bool one_col = false, x2_out = false;
    // FOLD: a byte offset per patch row (a row above / below the image is out of range as a whole: zeros), the left-most column of a
    // tile in the first column of tiles and the right-most of one in the last lie outside the image (one select each per row); the
    // single row in the whole tensor whose window would start 4 bytes BEFORE the tensor (sample 0, channel 0, image row 0, first tile)
    // is loaded from its start and shifted
    unsigned xrow[4] = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};
    bool left = false, right = false;
    int shifted = -1;
    auto address = [&](int item) {
        const unsigned p = tile0 + (unsigned)item * WDW_T + tl;
        const bool valid = item < a.items && p < a.P;
        const unsigned pv = valid ? p : 0u;
        const unsigned n = wino_div(pv, a.per_m, a.per_s1, a.per_s2), rem = pv - n * (unsigned)(a.TY * a.TX);
        const unsigned ty = wino_div(rem, a.tx_m, a.tx_s1, a.tx_s2), tx = rem - ty * (unsigned)a.TX;
        if constexpr (FOLD) {
            const int r0 = 2 * (int)ty - a.pady, c0 = 2 * (int)tx - a.padx;
            const int base = ((int)n * a.Ci + 64 * cb + 2 * cp) * plane + c0;
            left = c0 < 0; right = c0 + 3 >= a.Wx;
            shifted = -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = base + (r0 + i) * a.Wx;  // elements; negative only for the one row described above
                const bool rok = valid & ((unsigned)(r0 + i) < (unsigned)a.Hx);
                if (rok & (off < 0)) shifted = i;          // (image row 0 of channel 0 of sample 0, first column of tiles: no other valid row)
                xrow[i] = rok ? (unsigned)(off < 0 ? 0 : off) * 4u : 0x80000000u;
            }
        } else {
            xo = valid ? (unsigned)(((int)n * a.Ci + 64 * cb + 2 * cp) * plane + 2 * (int)ty * a.Ws + 2 * (int)tx) * 4u : 0x80000000u;
        }
        yo = valid ? (unsigned)(((int)n * a.Co + 64 * cob + 2 * cp) * oplane + 2 * (int)ty * a.Wd + 2 * (int)tx) * 4u : 0x80000000u;
        if constexpr (ODD) {
            const bool two_rows = 2 * (int)ty + 1 < a.Hd;
            yo1 = two_rows ? yo : 0x80000000u;
            if constexpr (!FOLD) xo3 = two_rows ? xo : 0x80000000u;   // (FOLD: that row lies below the image, out of range already)
            one_col = valid && 2 * (int)tx + 1 >= a.Wd;
            if constexpr (FOLD) x2_out = valid && 2 * (int)tx - a.padx + 2 >= a.Wx;
        }
    };
    auto load_one = [&](auto set, int k) {  // k = 0..11, compile-time at every call site
        constexpr int S = decltype(set)::value;
        if (k < 8) {
            const int ch = k >> 2, i = k & 3;
            // FOLD: the second channel's plane goes into the VECTOR offset - the descriptor's range check covers vector + immediate
            // offsets only, and the right-most window of the tensor's last row ends 4 bytes past the tensor: checked per dword, that
            // element reads as 0 (it is the folded padding column, `fix_right`) instead of touching memory behind the allocation
            const unsigned vo = FOLD ? xrow[i] + (ch ? plane4 : 0u) : (ODD && i == 3 ? xo3 : xo);
            xr[S][ch][i][0] = __builtin_amdgcn_raw_buffer_load_b64(xrs, vo, xso[ch][i], 0);
            xr[S][ch][i][1] = __builtin_amdgcn_raw_buffer_load_b64(xrs, vo + 8, xso[ch][i], 0);
        } else {
            const int ch = (k - 8) >> 1, r = (k - 8) & 1;
            yr[S][ch][r] = __builtin_amdgcn_raw_buffer_load_b64(yrs, ODD && r == 1 ? yo1 : yo, yso[ch][r], 0);
        }
    };
    auto load = [&](auto set, int item) {
        constexpr int S_ = decltype(set)::value;
        address(item);
        fix_left[S_] = left; fix_right[S_] = right; fix_shift[S_] = shifted;
        fix_col[S_] = one_col; fix_x2[S_] = x2_out;
#pragma unroll
        for (int k = 0; k < 12; ++k) load_one(set, k);
    };
    // transform of a loaded item, in two steps: `compute` (registers only: the 16 + 16 xi values of the thread's tile and channel
    // pair) and the stores, element (xi, channel quad cp / 2, tile, channels 2 (cp % 2), + 1) as one float2 each
    // (round 6, unfolded instantiations: the arithmetic runs on COLUMN PAIRS - a loaded b64 is two adjacent columns of one channel in a register pair, and the
    //  pairs' adds, subtractions and halvings are v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32 with op_sel / neg modifiers: 28 packed
    //  instructions per channel where the element-wise form issued 56; f32 VALU work cannot overlap the f32 MFMAs, every one counts)
    typedef float v2f __attribute__((ext_vector_type(2)));
    float xv[2][16], yv[2][16];
    auto compute = [&](auto set) {
        constexpr int S = decltype(set)::value;
        if constexpr (!FOLD) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                v2f dl[4], dh[4];  // patch row i: columns 0-1, columns 2-3
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dl[i] = __builtin_bit_cast(v2f, xr[S][ch][i][0]);
                    dh[i] = __builtin_bit_cast(v2f, xr[S][ch][i][1]);
                    if constexpr (FOLD) {
                        if (i < 2 && __any(fix_shift[S] == i)) {  // (one wave of one block of the launch; image row 0 is patch row 0 or 1)
                            const bool sh = fix_shift[S] == i;
                            dh[i].y = sh ? dh[i].x : dh[i].y; dh[i].x = sh ? dl[i].y : dh[i].x; dl[i].y = sh ? dl[i].x : dl[i].y;
                        }
                        dl[i].x = fix_left[S] ? 0.f : dl[i].x;
                        dh[i].y = fix_right[S] ? 0.f : dh[i].y;
                        if constexpr (ODD) dh[i].x = fix_x2[S] ? 0.f : dh[i].x;
                    } else if constexpr (ODD) {
                        dh[i].y = fix_col[S] ? 0.f : dh[i].y;
                    }
                }
                // B^T d, both column pairs of a row at once
                const v2f tl[4] = {dl[0] - dl[2], dl[1] + dl[2], dl[2] - dl[1], dl[3] - dl[1]};
                const v2f th[4] = {dh[0] - dh[2], dh[1] + dh[2], dh[2] - dh[1], dh[3] - dh[1]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // (B^T d) B: (t0 - t2, t1 + t2) and (t2 - t1, t3 - t1)
                    const v2f t22 = __builtin_shufflevector(th[i], th[i], 0, 0), t11 = __builtin_shufflevector(tl[i], tl[i], 1, 1);
                    const v2f o01 = tl[i] + t22 * (v2f){-1.f, 1.f};  // (an exact product: the same bits as the subtraction / addition)
                    const v2f o23 = th[i] - t11;
                    xv[ch][4 * i + 0] = o01.x; xv[ch][4 * i + 1] = o01.y;
                    xv[ch][4 * i + 2] = o23.x; xv[ch][4 * i + 3] = o23.y;
                }
                v2f r0 = __builtin_bit_cast(v2f, yr[S][ch][0]), r1 = __builtin_bit_cast(v2f, yr[S][ch][1]);
                if constexpr (ODD) { r0.y = fix_col[S] ? 0.f : r0.y; r1.y = fix_col[S] ? 0.f : r1.y; }
                const v2f g[4] = {r0, 0.5f * (r0 + r1), 0.5f * (r0 - r1), r1};  // G dy, both columns of a row at once
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // (G dy) G^T: g.x, (g.x + g.y) / 2, (g.x - g.y) / 2, g.y
                    const v2f gx = __builtin_shufflevector(g[i], g[i], 0, 0), gy_ = __builtin_shufflevector(g[i], g[i], 1, 1);
                    const v2f mid = 0.5f * (gx + gy_ * (v2f){1.f, -1.f});
                    yv[ch][4 * i + 0] = g[i].x;
                    yv[ch][4 * i + 1] = mid.x;
                    yv[ch][4 * i + 2] = mid.y;
                    yv[ch][4 * i + 3] = g[i].y;
                }
            }
        } else {
            // (the FOLD instantiations keep the element-wise form: with their border fix-ups on single elements of the pairs the packed
            //  form measured 6 % SLOWER - C3 303 -> 323 us, same box, alternating - while the unfolded kernels gain 4 - 7 %)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                float d[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float2 lo = __builtin_bit_cast(float2, xr[S][ch][i][0]), hi = __builtin_bit_cast(float2, xr[S][ch][i][1]);
                    d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
                    if constexpr (FOLD) {
                        if (i < 2 && __any(fix_shift[S] == i)) {  // (one wave of one block of the launch; image row 0 is patch row 0 or 1)
                            const bool sh = fix_shift[S] == i;
                            d[i][3] = sh ? d[i][2] : d[i][3]; d[i][2] = sh ? d[i][1] : d[i][2]; d[i][1] = sh ? d[i][0] : d[i][1];
                        }
                        d[i][0] = fix_left[S] ? 0.f : d[i][0];
                        d[i][3] = fix_right[S] ? 0.f : d[i][3];
                        if constexpr (ODD) d[i][2] = fix_x2[S] ? 0.f : d[i][2];
                    } else if constexpr (ODD) {
                        d[i][3] = fix_col[S] ? 0.f : d[i][3];
                    }
                }
                float tt[4][4];  // B^T d
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tt[0][j] = d[0][j] - d[2][j];
                    tt[1][j] = d[1][j] + d[2][j];
                    tt[2][j] = d[2][j] - d[1][j];
                    tt[3][j] = d[3][j] - d[1][j];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // (B^T d) B
                    xv[ch][4 * i + 0] = tt[i][0] - tt[i][2];
                    xv[ch][4 * i + 1] = tt[i][1] + tt[i][2];
                    xv[ch][4 * i + 2] = tt[i][2] - tt[i][1];
                    xv[ch][4 * i + 3] = tt[i][3] - tt[i][1];
                }
                float2 r0 = __builtin_bit_cast(float2, yr[S][ch][0]), r1 = __builtin_bit_cast(float2, yr[S][ch][1]);
                if constexpr (ODD) { r0.y = fix_col[S] ? 0.f : r0.y; r1.y = fix_col[S] ? 0.f : r1.y; }
                float g[4][2];  // G dy
                g[0][0] = r0.x; g[0][1] = r0.y;
                g[1][0] = 0.5f * (r0.x + r1.x); g[1][1] = 0.5f * (r0.y + r1.y);
                g[2][0] = 0.5f * (r0.x - r1.x); g[2][1] = 0.5f * (r0.y - r1.y);
                g[3][0] = r1.x; g[3][1] = r1.y;
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // (G dy) G^T
                    yv[ch][4 * i + 0] = g[i][0];
                    yv[ch][4 * i + 1] = 0.5f * (g[i][0] + g[i][1]);
                    yv[ch][4 * i + 2] = 0.5f * (g[i][0] - g[i][1]);
                    yv[ch][4 * i + 3] = g[i][1];
                }
            }
        }
        // the bias gradient rides on the transform: element (1, 1) of G dy G^T is a quarter of the tile's sum (exact scaling)
        bsum.x += yv[0][5];
        bsum.y += yv[1][5];
    };
    constexpr int XI = 16 * WDW_CQ / 2;  // float2 step from xi to xi + 1
    const int woff = (cp >> 1) * WDW_CQ + tl * 4 + 2 * (cp & 1);
    auto store = [&](int xi, float* xs, float* ys) {
        reinterpret_cast<float2*>(xs + woff)[xi * XI] = make_float2(xv[0][xi], xv[1][xi]);
        reinterpret_cast<float2*>(ys + woff)[xi * XI] = make_float2(yv[0][xi], yv[1][xi]);
    };

    nkmma::f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xi][e] = 0.f;

    // ---- prologue: item 0 transformed into image 0, item 1 loaded
    load(std::integral_constant<int, 0>{}, 0);
    load(std::integral_constant<int, 1>{}, 1);
    compute(std::integral_constant<int, 0>{});
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) store(xi, XS, YS);
    __syncthreads();

    // the MFMAs of item i on image `cur`, with: the loads of item i + 2 into register set `set` (item i's own, already transformed),
    // the transform of item i + 1 (register set set ^ 1, loaded during item i - 1) into image cur ^ 1
    auto item = [&](auto set, int i) {
        constexpr int S = decltype(set)::value;
        // lane (c, h): A = DY^[xi][co = 32 wr + c][tile 2 s + h], B = X^[xi][tile 2 s + h][ci = 32 wc + c], s = 0..3
        const float* const ya = YS + S * WDW_IMG + ((32 * wr + c) >> 2) * WDW_CQ + (c & 3) + 4 * h;
        const float* const xb = XS + S * WDW_IMG + ((32 * wc + c) >> 2) * WDW_CQ + (c & 3) + 4 * h;
        float* const xn = XS + (S ^ 1) * WDW_IMG;
        float* const yn = YS + (S ^ 1) * WDW_IMG;
        address(i + 2);
        fix_left[S] = left; fix_right[S] = right; fix_shift[S] = shifted;
        fix_col[S] = one_col; fix_x2[S] = x2_out;
        float av[2][4], bv[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { av[0][s] = ya[8 * s]; bv[0][s] = xb[8 * s]; }
        // the next item's transform arithmetic FIRST (its loads are a whole item old): f32 VALU work cannot overlap the f32 MFMAs
        // anyway, here it fills the wait for the item's first fragments after the barrier; the stores follow inside the MFMA groups
        compute(std::integral_constant<int, S ^ 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            if (xi + 1 < 16) {
#pragma unroll
                for (int s = 0; s < 4; ++s) { av[(xi + 1) & 1][s] = ya[(xi + 1) * 16 * WDW_CQ + 8 * s]; bv[(xi + 1) & 1][s] = xb[(xi + 1) * 16 * WDW_CQ + 8 * s]; }
            }
            store(xi, xn, yn);
            if (xi < 12) load_one(set, xi);  // item i + 2 into the register set item i was transformed out of
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[xi & 1][s], bv[xi & 1][s], acc[xi], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // LDS read (the next xi's fragments)
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // LDS write (the next item's images)
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // buffer load
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // VALU
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int i = 0; i < a.items; i += 2) {
        item(std::integral_constant<int, 0>{}, i);
        item(std::integral_constant<int, 1>{}, i + 1);
    }

    // ---- the block's sums to its slab: wave (wr, wc), lane (c, h), register e = row (co) 8 (e >> 2) + 4 h + (e & 3), column (ci) c
    float* const slab = a.slabs + ((size_t)blockIdx.x * gridDim.y + pair) * (16 * 64 * 64);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * wr + 8 * (e >> 2) + 4 * h + (e & 3), col = 32 * wc + c;
            slab[(xi * 64 + row) * 64 + col] = acc[xi][e];
        }
    if (a.bslabs != nullptr && cb == 0) {  // every ci block of this co block saw the same dY: the first one reports
        float2 b = bsum;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { b.x += __shfl_xor(b.x, o, 64); b.y += __shfl_xor(b.y, o, 64); }
        if (tl == 0) {
            float* const bs = a.bslabs + (size_t)blockIdx.x * a.Co + 64 * cob + 2 * cp;
            bs[0] = 4.f * b.x; bs[1] = 4.f * b.y;
        }
    }
}

// slabs -> dW (Co, Ci, 3, 3).  One block of 1024 threads per (output channel, block of 64 input channels): thread (quarter q of the
// slices, xi, four input channels) adds its quarter's slabs in slice order (float4 loads, whole 256-byte rows per 16 lanes, eight in
// flight), the quarters meet in LDS and 64 threads add them in order and apply A^T . A.  (First form: 256 threads = 16 xi x 16 input
// channels, one chain over all slices per thread - 64-byte row pieces and one to eight loads in flight: 36 - 42 us for the 64 MB of
// C3's slabs.)  Blocks beyond Co * Ci / 64 reduce the bias slabs (one thread per output channel).
__global__ __launch_bounds__(1024) void wino_dw_reduce_kernel(float* __restrict__ dw, float* __restrict__ db, const float* __restrict__ slabs,
                                                               const float* __restrict__ bslabs, int Co, int Ci, int slices, int pairs, int cib,
                                                               int assign, int assign_b) {
    __shared__ __attribute__((aligned(16))) float M[4][16][64];
    const int nb = Co * (Ci / 64);
    if ((int)blockIdx.x >= nb) {
        const int co = ((int)blockIdx.x - nb) * 1024 + threadIdx.x;
        if (db == nullptr || co >= Co) return;
        // the slices' sums are large and alike (C3: 128 of ~1570 each, total ~2e5): one f32 chain over them loses ~sqrt(slices) ulps of
        // the TOTAL (0.09 at C3 against 0.017 for a pairwise sum); in f64 the chain is exact to the last f32 bit
        double s = 0.0;
        for (int sl = 0; sl < slices; ++sl) s += (double)bslabs[(size_t)sl * Co + co];
        db[co] = assign_b ? (float)s : db[co] + (float)s;
        return;
    }
    const int co = blockIdx.x / (Ci / 64), cb = blockIdx.x % (Ci / 64);
    const int q = threadIdx.x >> 8, xi = (threadIdx.x >> 4) & 15, c4 = threadIdx.x & 15;
    const int pair = (co / 64) * cib + cb;
    const size_t sstep = (size_t)pairs * (16 * 64 * 64);
    const float* p = slabs + (size_t)pair * (16 * 64 * 64) + ((size_t)xi * 64 + (co % 64)) * 64 + 4 * c4;
    const int per = (slices + 3) / 4, s0 = q * per, s1 = s0 + per < slices ? s0 + per : slices;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int sl = s0;
    for (; sl + 16 <= s1; sl += 16) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(sl + u) * sstep);
#pragma unroll
        for (int u = 0; u < 16; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; sl < s1; ++sl) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)sl * sstep);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(&M[q][xi][4 * c4]) = s;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int l = threadIdx.x, ci = 64 * cb + l;
    float m[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) m[x] = ((M[0][x][l] + M[1][x][l]) + M[2][x][l]) + M[3][x][l];
    float tm[3][4];  // A^T M
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        tm[0][j] = (m[0 + j] + m[4 + j]) + m[8 + j];
        tm[1][j] = m[4 + j] - m[8 + j];
        tm[2][j] = (m[4 + j] + m[8 + j]) + m[12 + j];
    }
    float* o = dw + ((size_t)co * Ci + ci) * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v0 = (tm[k][0] + tm[k][1]) + tm[k][2], v1 = tm[k][1] - tm[k][2], v2 = (tm[k][1] + tm[k][2]) + tm[k][3];
        if (assign) { o[3 * k] = v0; o[3 * k + 1] = v1; o[3 * k + 2] = v2; }
        else { o[3 * k] += v0; o[3 * k + 1] += v1; o[3 * k + 2] += v2; }
    }
}

// Host side.  `taken` = false: not a case for this path (the caller goes on to the implicit-GEMM kernel).
// (Hs, Ws): extents of the convolution's (padded) input.  pady / padx > 0: `x` is the UNPADDED input (Hs - 2 pady) x (Ws - 2 padx) and
// the Zero padding is folded in (the FOLD instantiation; 0 or 1 per axis).  `force` / `dry`: as wino_launch.
int wino_dw_launch(nk_device* dev, const float* gy, const float* x, float* dw, float* db, int N, int Ci, int Co, int Hs, int Ws, int assign,
                   int assign_b, double flop, bool* taken, int pady = 0, int padx = 0, bool force = false, bool dry = false) {
    *taken = false;
    const int mode = force ? 1 : dev->tune_conv_wino_dw;  // -1 rule, 0 never, 1 whenever the shape allows
    if (mode == 0 || (!force && dev->tune_conv_winograd == 0)) return NK_OK;
    if (pady < 0 || pady > 1 || padx < 0 || padx > 1) return NK_OK;
    const bool fold = pady + padx > 0;
    const int Hx = Hs - 2 * pady, Wx = Ws - 2 * padx;
    if (Hx < 1 || Wx < 1) return NK_OK;
    const int Hd = Hs - 2, Wd = Ws - 2;
    if (Hd < 2 || Wd < 2 || Ci % 64 != 0 || Co % 64 != 0) return NK_OK;
    const bool odd = Hd % 2 != 0 || Wd % 2 != 0;  // border tiles with one row / one column: the ODD instantiations
    if (!al16(x) || !al16(gy)) return NK_OK;
    const long long P = (long long)N * ((Hd + 1) / 2) * ((Wd + 1) / 2);
    const long long x_bytes = (long long)N * Ci * Hx * Wx * 4, gy_bytes = (long long)N * Co * Hd * Wd * 4;
    if (P >= (1LL << 30) || x_bytes >= 0x7fffffffLL || gy_bytes >= 0x7fffffffLL) return NK_OK;
    const int pairs = (Co / 64) * (Ci / 64);
    if (pairs > dev->num_cus) return NK_OK;
    int slices = dev->num_cus / pairs;                          // one block per CU
    const long long items_all = (P + WDW_T - 1) / WDW_T;
    if (slices > items_all) slices = (int)items_all;
    long long items = (items_all + slices - 1) / slices;
    items += items & 1;                                         // the item loop is unrolled by two (register sets, LDS images)
    // by rule: at least 16 items per slice (the slabs and the second kernel are a fixed cost: 256 KB per block)
    if (mode < 0 && items < 16) return NK_OK;
    if (dry) { *taken = true; return NK_OK; }
    const size_t slab_bytes = round256((size_t)slices * pairs * 16 * 64 * 64 * sizeof(float));
    void* ws = nullptr;
    int rc = nk_workspace(dev, slab_bytes + (db ? (size_t)slices * Co * sizeof(float) : 0), &ws);
    if (rc) return rc;
    rc = nk_prof_start(dev, NK_KERNEL_CONV, flop);
    if (rc) return rc;
    WinoDwArgs a{};
    a.x = x; a.gy = gy; a.slabs = (float*)ws; a.bslabs = db ? (float*)((char*)ws + slab_bytes) : nullptr;
    a.N = N; a.Ci = Ci; a.Co = Co; a.Hs = Hs; a.Ws = Ws; a.Hd = Hd; a.Wd = Wd; a.TY = (Hd + 1) / 2; a.TX = (Wd + 1) / 2;
    a.P = (unsigned)P; a.items = (int)items; a.cib = Ci / 64;
    wino_magic((unsigned)(a.TY * a.TX), &a.per_m, &a.per_s1, &a.per_s2);
    wino_magic((unsigned)a.TX, &a.tx_m, &a.tx_s1, &a.tx_s2);
    a.x_bytes = (int)x_bytes; a.gy_bytes = (int)gy_bytes;
    a.Hx = Hx; a.Wx = Wx; a.pady = pady; a.padx = padx;
    const dim3 grid((unsigned)slices, (unsigned)pairs);
    if (fold && odd) hipLaunchKernelGGL((wino_dw_kernel<true, true>), grid, dim3(256), 0, dev->compute, a);
    else if (fold) hipLaunchKernelGGL((wino_dw_kernel<true>), grid, dim3(256), 0, dev->compute, a);
    else if (odd) hipLaunchKernelGGL((wino_dw_kernel<false, true>), grid, dim3(256), 0, dev->compute, a);
    else hipLaunchKernelGGL((wino_dw_kernel<false>), grid, dim3(256), 0, dev->compute, a);
    NK_LAUNCH_CHECK();
    const int nb = Co * (Ci / 64) + (db ? (Co + 1023) / 1024 : 0);
    hipLaunchKernelGGL(wino_dw_reduce_kernel, dim3((unsigned)nb), dim3(1024), 0, dev->compute, dw, db, (const float*)a.slabs, (const float*)a.bslabs, Co,
                       Ci, slices, pairs, Ci / 64, assign, assign_b);
    NK_LAUNCH_CHECK();
    *taken = true;
    ++dev->wino_launches;
    return nk_prof_stop(dev);
}
