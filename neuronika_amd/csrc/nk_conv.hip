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
#include "nk_conv_geom.h"
#include "nk_conv_generic.h"
#include "nk_conv_fast.h"
#include "nk_conv_direct.h"

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

#include "nk_conv_winograd.h"
#include "nk_conv_winograd_dw.h"
#include "nk_conv_s2dx.h"
#include "nk_conv_s2fwd.h"

// 3 x 3, stride 1, dilation 1, one group, two spatial dimensions: the shapes wino_launch may take
bool wino_shape(const ConvGeom& g) {
    return g.groups == 1 && g.in[0] == 1 && g.k[0] == 1 && g.k[1] == 3 && g.k[2] == 3 && g.stride[1] == 1 && g.stride[2] == 1 && g.dil[1] == 1 &&
           g.dil[2] == 1;
}

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
    if (wino_shape(g)) {  // Winograd F(2x2, 3x3) on the MFMA core (nk_conv_winograd.h)
        bool taken = false;
        rc = wino_launch(dev, false, x, w, y, bias, g.N, g.Cin, g.Cout, g.in[1], g.in[2], g.out[1], g.out[2], 0, 0, 1,
                         2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK, &taken);
        if (rc || taken) return rc;
    }
    if (g.groups == 1 && g.in[0] == 1 && g.k[0] == 1 && g.k[1] == 3 && g.k[2] == 3 && g.stride[1] == 2 && g.stride[2] == 2 && g.dil[1] == 1 &&
        g.dil[2] == 1) {  // nine tap products on staged tap planes (nk_conv_s2fwd.h)
        bool taken = false;
        rc = s2f_launch(dev, x, w, bias, y, g.N, g.Cin, g.Cout, g.in[1], g.in[2], g.out[1], g.out[2], 0,
                        2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK, &taken);
        if (rc || taken) return rc;
    }
    const int K = g.Cg * g.KK;
    const long long x_elems = (long long)g.N * g.Cin * g.inplane, y_elems = (long long)g.N * g.Cout * g.L;
    const long long fcols = (long long)g.N * g.out[0] * g.out[1] * ((g.out[2] + 3) & ~3);  // row-padded column space
    if (g.Cg % BK == 0 && (g.stride[2] == 1 || g.stride[2] == 2) && g.out[2] >= 4 && x_elems < 0x7fffffffLL && y_elems < 0x7fffffffLL && fcols < 0x7fffff00LL) {
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
        // ... and a grid that cannot even fill half the resident slots (late layers: 3x3 512 -> 512 at 7x7 is 224 tiles with a
        // 144-k-tile reduction each - 0.50 of peak in round 5 with half the chip idle) is split along k as a whole, as the input
        // gradient below has done since round 3: every tile is a "tail" tile
        const int tail = tiles > slots ? tiles % slots : (tiles * 2 <= slots ? tiles : 0);
        if (groups == 1 && tail > 0 && tail * 2 <= slots && nkt >= 4) {
            int S = slots / tail;
            if (S > nkt / 2) S = nkt / 2;
            if (tiles <= slots && S > nkt / 8) S = nkt / 8;  // whole-grid split: keep >= 8 k-tiles per block
            if (S < 1) S = 1;
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
        void* wsf = nullptr;
        rc = nk_workspace(dev, tables + slab_bytes, &wsf);
        if (rc) return rc;
        float* wp = (float*)wsf;
        int* tapoff = (int*)((char*)wsf + wp_bytes);
        int4* tapd = (int4*)((char*)wsf + wp_bytes + to_bytes);
        if (slab_bytes) fp.slabs = (float*)((char*)wsf + tables);
        hipLaunchKernelGGL(conv_wp_kernel, dim3(nk_stream_grid((size_t)g.Cout * K, 256)), dim3(256), 0, dev->compute, wp, w, g, tapoff, tapd);
        NK_LAUNCH_CHECK();
        fp.g = g; fp.x = x; fp.wp = wp; fp.y = y; fp.tapoff = tapoff;
        dim3 fgrid(nblocks, 1, groups);
        rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
        if (rc) return rc;
#define NK_LAUNCH_FF(AL, TI_, RP_)                                                                                          \
    do {                                                                                                                    \
        if (g.stride[2] == 2) hipLaunchKernelGGL((conv_fwd_fast_kernel<AL, TI_, RP_, 2>), fgrid, dim3(NT), 0, dev->compute, fp); \
        else hipLaunchKernelGGL((conv_fwd_fast_kernel<AL, TI_, RP_, 1>), fgrid, dim3(NT), 0, dev->compute, fp);               \
    } while (0)
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
            const bool buf = (long long)g.N * g.Cout * g.L * 4 < 0x7fffffffLL;  // 32-bit byte offsets through a buffer descriptor
            if (g.k[1] == 3 && buf) hipLaunchKernelGGL((conv_direct_bwd_input_rows_kernel<3, 3, true>), rgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            else if (g.k[1] == 5 && buf) hipLaunchKernelGGL((conv_direct_bwd_input_rows_kernel<5, 5, true>), rgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            else if (buf) hipLaunchKernelGGL((conv_direct_bwd_input_rows_kernel<1, 3, true>), rgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            else if (g.k[1] == 3) hipLaunchKernelGGL((conv_direct_bwd_input_rows_kernel<3, 3>), rgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            else if (g.k[1] == 5) hipLaunchKernelGGL((conv_direct_bwd_input_rows_kernel<5, 5>), rgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            else hipLaunchKernelGGL((conv_direct_bwd_input_rows_kernel<1, 3>), rgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            NK_LAUNCH_CHECK();
            return nk_prof_stop(dev);
        }
        const int pt = g.uinplane >= 512 ? 4 : 1;
        const dim3 dgrid((unsigned)(g.N * g.Cin), (unsigned)((g.uinplane + 256 * pt - 1) / (256 * pt)));
        if (!unit && g.dil[0] == 1 && g.dil[1] == 1 && g.dil[2] == 1) {  // a lane walks its own residue class of taps: no division per tap
            if (pt == 4) hipLaunchKernelGGL((conv_direct_bwd_input_strided_kernel<4>), dgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            else hipLaunchKernelGGL((conv_direct_bwd_input_strided_kernel<1>), dgrid, dim3(256), 0, dev->compute, dx, gy, w, g);
            NK_LAUNCH_CHECK();
            return nk_prof_stop(dev);
        }
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
    if (wino_shape(g)) {  // the full correlation with the flipped kernel, the Pad node's padding folded in: patch origin 2 t - (2 - pad)
        bool taken = false;
        rc = wino_launch(dev, true, gy, w, dx, nullptr, g.N, g.Cout, g.Cin, g.out[1], g.out[2], g.uin[1], g.uin[2], 2 - g.pad[1], 2 - g.pad[2],
                         assign, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK, &taken);
        if (rc || taken) return rc;
    }
    if (g.groups == 1 && g.in[0] == 1 && g.k[0] == 1 && g.k[1] == 3 && g.k[2] == 3 && g.stride[1] == 2 && g.stride[2] == 2 && g.dil[1] == 1 &&
        g.dil[2] == 1 && g.pad[0] == 0 && g.pad[1] == g.pad[2]) {  // the four stride phases of a tile in one block walk (nk_conv_s2dx.h)
        bool taken = false;
        rc = s2dx_launch(dev, gy, w, dx, g.N, g.Cout, g.Cin, g.out[1], g.out[2], g.uin[1], g.uin[2], g.pad[1], assign,
                         2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK, &taken);
        if (rc || taken) return rc;
    }
    const int K = g.Mg * g.KK;
    {
        const long long x_elems = (long long)g.N * g.Cin * g.inplane, y_elems = (long long)g.N * g.Cout * g.L;
        // stride phases (see BwdInPhase): residue classes of the padded input coordinate, each with its taps and columns
        const int nphase = g.stride[0] * g.stride[1] * g.stride[2];
        BwdInPhaseTable tbl{};
        long long tiles_n = 0, max_cols = 0;
        bool fits = nphase <= MAX_PHASES && g.Mg % BK == 0 && g.out[2] >= 4 && x_elems < 0x7fffffffLL && y_elems < 0x7fffffffLL;
        bool empty_phase = false;  // a residue class no tap reaches (1 x 1 kernel, stride 2: three of the four): its gradient is zero
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
                if (ntaps == 0 && cols > 0) {  // no tiles: `+=` has nothing to add; a first write zeroes the tensor once (below)
                    empty_phase = true;        // instead of storing zeros through the tile epilogue's interleaved 4-byte stores
                    cols = 0;
                }
                tiles_n += (cols + 127) / 128;
                if (cols > max_cols) max_cols = cols;
            }
            for (int pid = nphase; pid < MAX_PHASES; ++pid) tbl.ph[pid].tile_begin = 0x7fffffff;
            fits = max_cols < 0x7fffff00LL && tiles_n < 0x7fffffffLL && tiles_n > 0;
        }
        if (fits) {
            if (empty_phase && g.assign)  // every position belongs to exactly one phase: the others assign theirs after this
                NK_HIP(hipMemsetAsync(dx, 0, (size_t)g.N * g.Cin * g.uinplane * sizeof(float), dev->compute));
            const size_t wq_bytes = round256((size_t)g.Cout * g.Cg * g.KK * sizeof(float));
            const size_t td_bytes = round256((size_t)g.KK * sizeof(int4));
            const size_t ph_bytes = round256(sizeof(BwdInPhaseTable));
            FastBwdInArgs fp{};
            const int fti = g.Cg <= 64 || (g.Cg % 128 != 0 && g.Cg % 64 == 0) ? 1 : 2;
            fp.tiles_m = (g.Cg + 64 * fti - 1) / (64 * fti);
            fp.tiles_n = (int)tiles_n;
            // tail balancing: tiles beyond the last whole wave of resident blocks (3 per CU for the 64-row tile, 2 for the
            // 128-row one) are split along k (C3: 3136 tiles over 768 slots = 4.08 waves, i.e. a fifth wave 8 % full)
            const int tiles = fp.tiles_m * fp.tiles_n, slots = (fti == 1 ? 3 : 2) * dev->num_cus, nkt = g.Mg * g.KK / BK;
            int nblocks = tiles;
            size_t slab_bytes = 0;
            fp.full_blocks = tiles;
            // ... and a grid that cannot even fill half the resident slots (late layers: 3x3 512 -> 512 at 7x7 is 224 tiles
            // with a 144-k-tile reduction each) is split along k as a whole: every tile is a "tail" tile
            const int tail = tiles > slots ? tiles % slots : (tiles * 2 <= slots ? tiles : 0);
            if (groups == 1 && nphase == 1 && tail > 0 && tail * 2 <= slots && nkt >= 4) {
                int S = slots / tail;
                if (S > nkt / 2) S = nkt / 2;
                if (tiles <= slots && S > nkt / 8) S = nkt / 8;  // whole-grid split: keep >= 8 k-tiles per block
                if (S < 1) S = 1;
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
            void* wsf = nullptr;
            rc = nk_workspace(dev, wq_bytes + td_bytes + ph_bytes + slab_bytes, &wsf);
            if (rc) return rc;
            float* wq = (float*)wsf;
            int4* tapd = (int4*)((char*)wsf + wq_bytes);
            BwdInPhase* phases = (BwdInPhase*)((char*)wsf + wq_bytes + td_bytes);
            if (slab_bytes) fp.slabs = (float*)((char*)wsf + wq_bytes + td_bytes + ph_bytes);
            hipLaunchKernelGGL(conv_wq_tables_kernel, dim3(nk_stream_grid((size_t)g.Cout * g.Cg * g.KK, 256)), dim3(256), 0, dev->compute,
                               wq, w, tapd, phases, tbl, g);
            NK_LAUNCH_CHECK();
            fp.g = g; fp.dx = dx; fp.gy = gy; fp.wq = wq; fp.tapd = tapd; fp.phases = phases; fp.nphase = nphase;
            const bool al = g.Cg % (64 * fti) == 0;
            dim3 fgrid(nblocks, 1, groups);
            rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
            if (rc) return rc;
            if (al && fti == 2) hipLaunchKernelGGL((conv_bwd_input_fast_kernel<true, 2>), fgrid, dim3(NT), 0, dev->compute, fp);
            else if (al) hipLaunchKernelGGL((conv_bwd_input_fast_kernel<true, 1>), fgrid, dim3(NT), 0, dev->compute, fp);
            else if (fti == 2) hipLaunchKernelGGL((conv_bwd_input_fast_kernel<false, 2>), fgrid, dim3(NT), 0, dev->compute, fp);
            else hipLaunchKernelGGL((conv_bwd_input_fast_kernel<false, 1>), fgrid, dim3(NT), 0, dev->compute, fp);
            NK_LAUNCH_CHECK();
            if (fp.tail_splits) {
                const dim3 rgrid(tiles - fp.full_blocks, (64 * fti * 128) / 1024, 1);
                if (fti == 2) hipLaunchKernelGGL((conv_bwd_input_tail_reduce_kernel<128>), rgrid, dim3(256), 0, dev->compute, fp);
                else hipLaunchKernelGGL((conv_bwd_input_tail_reduce_kernel<64>), rgrid, dim3(256), 0, dev->compute, fp);
                NK_LAUNCH_CHECK();
            }
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

// `db` (optional, with `assign_b`): the conv module's bias gradient, db[co] (+)= sum of gy over samples and positions.  The MFMA
// pass with quad staging produces it from the operand it stages anyway; the other paths run the un-broadcast reduction.
int conv_bwd_kernel(nk_device* dev, int nd, float* dw, const int* w_shape, const float* gy, const float* x,
                    const int* x_shape, const int* stride, const int* dilation, int groups, int assign, float* db = nullptr,
                    int assign_b = 0) {
    NK_USE(dev);
    ConvGeom g;
    int rc = make_geom(nd, x_shape, w_shape, stride, dilation, groups, &g);
    if (rc) return rc;
    auto bias_by_reduction = [&]() -> int {  // AdditionBackwardRight of the (Cout,1,..) bias as its own reduction
        if (!db) return NK_OK;
        int gshape[NK_MAX_DIMS], bshape[NK_MAX_DIMS];
        gshape[0] = g.N; gshape[1] = g.Cout; bshape[0] = g.Cout;
        for (int i = 0; i < nd; ++i) { gshape[2 + i] = g.out[3 - nd + i]; bshape[1 + i] = 1; }
        return (assign_b ? nk_unbroadcast_assign : nk_unbroadcast_add)(dev, db, bshape, nd + 1, gy, gshape, nd + 2);
    };
    const bool quadr = (g.stride[2] == 1 || g.stride[2] == 2) && g.out[2] >= 4;  // row-padded quad staging (see the kernel)
    const long long R = quadr ? (long long)g.N * g.out[0] * g.out[1] * ((g.out[2] + 3) & ~3) : (long long)g.N * g.L;
    const int Kc = g.Cg * g.KK;
    if ((long long)g.Cout * Kc == 0) return bias_by_reduction();
    NK_CHECK(dw && gy && x, "null pointer in nk_conv_bwd_kernel");
    if (R == 0) {  // empty batch: the gradient is zero
        if (assign) NK_HIP(hipMemsetAsync(dw, 0, (size_t)g.Cout * Kc * sizeof(float), dev->compute));
        if (db && assign_b) NK_HIP(hipMemsetAsync(db, 0, (size_t)g.Cout * sizeof(float), dev->compute));
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
        rc = nk_prof_stop(dev);
        return rc ? rc : bias_by_reduction();
    }
    if (wino_shape(g)) {  // Winograd F(3x3, 2x2): the tiles as the reduction dimension (nk_conv_winograd_dw.h); dW and db in one call
        bool taken = false;
        rc = wino_dw_launch(dev, gy, x, dw, db, g.N, g.Cin, g.Cout, g.in[1], g.in[2], assign, assign_b,
                            2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK, &taken);
        if (rc || taken) return rc;
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
    // (a multiple of) 2 * CUs blocks — a ragged second wave would idle most of the chip.
    // (Measured and dropped in round 2: running the last column tile of C3 - Kc = 576 = 4 * 128 + 64 - through the 64-wide
    // body, i.e. issuing none of the 11 % of MFMAs that fall on padding columns, did not change the launch time with
    // one block per tile (565 -> 566 us) and LOST 3 % once those blocks were given twice the reduction range to even
    // out the work (584 us): the launch is bound by the per-k-tile chain gathers -> wait -> mask -> LDS store -> barrier
    // (~1 us per k-tile more than the dense GEMM's), not by MFMA issue.  Round 3, with the pass at 0.89 of what its MFMA count
    // allows: the narrow tile's blocks with their waves 4 x 1 on the 64 real columns (half the MFMAs per k-tile), listed last in
    // every XCD's share of the grid, splits sized for 4.5 tile units (112 instead of 102): 527 -> 611 us.  A narrow block still
    // pays the whole chain per k-tile, so it is not half a block, and 560 blocks no longer fit one wave of resident slots.)
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
    // A column count that ends in half a tile (3 x 3 on 64 channels: 576 = 4.5 x 128): the mixed launch - 128-wide bodies for the
    // whole tiles, the 64-wide body for the last one (no MFMA on padding columns), the narrow tile's reduction cut into fewer,
    // longer ranges so that a narrow block takes as long as a wide one.  `narrow_cost` = time of a narrow block's k-tile in
    // percent of a wide block's (the measured rules of nk_common.h; nk_dev_tune NK_TUNE_CONV_NARROW overrides, 0 = the uniform launch).
    bool mixed = false;
    {
        const int narrow_cost = dev->tune_conv_narrow >= 0 ? dev->tune_conv_narrow : (ti == 2 ? NK_CONV_NARROW_128 : NK_CONV_NARROW_64);
        if (narrow_cost > 0 && quadr && g.stride[2] == 1 && tj == 2 && Kc % 128 == 64 && p.tiles_n >= 2) {
            // s_w wide splits, s_n narrow ones: wide tiles * s_w + narrow tiles * s_n <= slots * waves, s_n = s_w * cost / 100
            const long long nw = (long long)p.tiles_m * (p.tiles_n - 1) * groups, nn = (long long)p.tiles_m * groups;
            long long sw = slots * waves * 100 / (nw * 100 + nn * narrow_cost);
            if (sw > rtiles) sw = rtiles;
            if (sw > 1024) sw = 1024;
            long long sn = sw * narrow_cost / 100;
            if (sw >= 2 && sn >= 1) {
                long long rw = (rtiles + sw - 1) / sw;
                sw = (rtiles + rw - 1) / rw;
                long long rn = (rtiles + sn - 1) / sn;
                sn = (rtiles + rn - 1) / rn;
                if (sn < sw && groups == 1) {
                    mixed = true;
                    splits = sw; rts = rw;
                    p.r_per_split = rw * BK;
                    p.narrow_splits = (int)sn;
                    p.r_per_split_narrow = rn * BK;
                }
            }
        }
    }
    const long long dw_elems = (long long)g.Cout * Kc;
    const bool fused_bias = db && quadr;
    const size_t slab_bytes = round256((size_t)splits * dw_elems * sizeof(float));
    void* ws = nullptr;
    rc = nk_workspace(dev, slab_bytes + (fused_bias ? (size_t)splits * g.Cout * sizeof(float) : 0), &ws);
    if (rc) return rc;
    p.slabs = (float*)ws;
    p.bias_slabs = fused_bias ? (float*)((char*)ws + slab_bytes) : nullptr;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n * splits), 1, groups);
    const bool vec_g = (g.L % 4 == 0) && al16(gy);
    rc = nk_prof_start(dev, NK_KERNEL_CONV, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK);
    if (rc) return rc;
    if (mixed) {
        grid.x = (unsigned)(p.tiles_m * (p.tiles_n - 1) * splits + p.tiles_m * p.narrow_splits);
        if (ti == 2) hipLaunchKernelGGL((conv_bwd_kernel_mixed_kernel<true, 2, true, 1>), grid, dim3(NT), 0, dev->compute, p);
        else hipLaunchKernelGGL((conv_bwd_kernel_mixed_kernel<true, 1, true, 1>), grid, dim3(NT), 0, dev->compute, p);
    } else
#define NK_LAUNCH_BWK(VG, TI_, TJ_, Q) hipLaunchKernelGGL((conv_bwd_kernel_kernel<VG, TI_, TJ_, Q>), grid, dim3(NT), 0, dev->compute, p)
    if (quadr && g.stride[2] == 2) {
#define NK_LAUNCH_BWK2(TI_, TJ_) hipLaunchKernelGGL((conv_bwd_kernel_kernel<true, TI_, TJ_, true, 2>), grid, dim3(NT), 0, dev->compute, p)
        if (ti == 2 && tj == 2) NK_LAUNCH_BWK2(2, 2);
        else if (ti == 2) NK_LAUNCH_BWK2(2, 1);
        else if (tj == 2) NK_LAUNCH_BWK2(1, 2);
        else NK_LAUNCH_BWK2(1, 1);
#undef NK_LAUNCH_BWK2
    } else if (quadr) {
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
    // dW (+)= sum over splits (fixed order) of the slabs; with the fused bias gradient db[co] (+)= the per-split sums' sum, by the
    // blocks behind the dW ones in the same launch
    hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3((unsigned)((dw_elems + 63) / 64 + (fused_bias ? (g.Cout + 63) / 64 : 0))), dim3(256), 0,
                       dev->compute, dw, p.slabs, dw_elems, (int)splits, assign, fused_bias ? db : nullptr, p.bias_slabs,
                       (long long)g.Cout, assign_b, mixed ? Kc : 0, Kc - 64, p.narrow_splits);
    NK_LAUNCH_CHECK();
    rc = nk_prof_stop(dev);
    if (rc) return rc;
    return fused_bias ? NK_OK : bias_by_reduction();
}

// ---- the Conv module with its Zero padding folded into the forward and the kernel gradient (Winograd forms only) ----------------
// geometry of pad -> convolution from the UNPADDED input shape; `fold` = the Winograd kernels can take it with the padding folded in
// (3 x 3, stride 1, dilation 1, one group, two spatial dimensions, padding 0 or 1 per axis and not all zero)
int fold_geom(int nd, const int* x_shape, const int* padding, const int* w_shape, const int* stride, const int* dilation, int groups, ConvGeom* g,
              bool* fold) {
    NK_CHECK(nd >= 1 && nd <= 3 && x_shape && padding && w_shape && stride && dilation, "bad arguments of a padded convolution entry");
    int pshape[5] = {x_shape[0], x_shape[1], 1, 1, 1};
    bool any = false, small = true;
    for (int d = 0; d < nd; ++d) {
        NK_CHECK(padding[d] >= 0, "negative padding on axis %d", d);
        pshape[2 + d] = x_shape[2 + d] + 2 * padding[d];
        any = any || padding[d] != 0;
        small = small && padding[d] <= 1;
    }
    const int rc = make_geom(nd, pshape, w_shape, stride, dilation, groups, g);
    if (rc) return rc;
    *fold = nd == 2 && any && small && wino_shape(*g);
    return NK_OK;
}

int conv_padding_folds(nk_device* dev, int nd, const int* x_shape, const int* padding, const int* w_shape, const int* stride, const int* dilation,
                       int groups, int* folds) {
    NK_USE(dev);
    NK_CHECK(folds != nullptr, "null result pointer");
    *folds = 0;
    ConvGeom g;
    bool fold = false;
    int rc = fold_geom(nd, x_shape, padding, w_shape, stride, dilation, groups, &g, &fold);
    if (rc || !fold) return rc;
    bool fwd = false, dwk = false;  // by the rules in force on this handle: both kernels would be the Winograd ones anyway
    rc = wino_launch(dev, false, nullptr, nullptr, nullptr, nullptr, g.N, g.Cin, g.Cout, x_shape[2], x_shape[3], g.out[1], g.out[2], padding[0],
                     padding[1], 1, 0.0, &fwd, false, true);
    if (rc) return rc;
    rc = wino_dw_launch(dev, nullptr, nullptr, nullptr, nullptr, g.N, g.Cin, g.Cout, g.in[1], g.in[2], 1, 1, 0.0, &dwk, padding[0], padding[1], false,
                        true);
    if (rc) return rc;
    *folds = fwd && dwk ? 1 : 0;
    return NK_OK;
}

// Pad::forward (Zero) of `x` into the device's operand scratch - the region no convolution kernel uses for its own slabs / tables
int padded_copy(nk_device* dev, int nd, const float* x, const int* x_shape, const int* padding, float** xp, int* pshape) {
    size_t elems = (size_t)x_shape[0] * x_shape[1];
    pshape[0] = x_shape[0]; pshape[1] = x_shape[1];
    for (int d = 0; d < nd; ++d) {
        pshape[2 + d] = x_shape[2 + d] + 2 * padding[d];
        elems *= (size_t)pshape[2 + d];
    }
    void* p = nullptr;
    int rc = nk_operand_scratch(dev, elems * sizeof(float), &p);
    if (rc) return rc;
    *xp = (float*)p;
    return nk_pad_const_fwd(dev, nd, x, x_shape, *xp, padding, 0.f);
}

int conv_fwd_padded(nk_device* dev, int nd, const float* x, const int* x_shape, const int* padding, const float* w, const int* w_shape,
                    const float* bias, float* y, const int* stride, const int* dilation, int groups) {
    NK_USE(dev);
    ConvGeom g;
    bool fold = false;
    int rc = fold_geom(nd, x_shape, padding, w_shape, stride, dilation, groups, &g, &fold);
    if (rc) return rc;
    NK_CHECK(x && w && y, "null pointer in nk_conv_bias_fwd_padded");
    bool taken = false;
    if (fold && dev->tune_conv_winograd != 0)  // (the knob's "never" holds here too; `force` only skips the block-count rule)
        rc = wino_launch(dev, false, x, w, y, bias, g.N, g.Cin, g.Cout, x_shape[2], x_shape[3], g.out[1], g.out[2], padding[0], padding[1], 1,
                         2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK, &taken, true);
    if (rc) return rc;
    if (taken) return NK_OK;
    if (nd == 2 && groups == 1 && w_shape[2] == 3 && w_shape[3] == 3 && stride[0] == 2 && stride[1] == 2 && dilation[0] == 1 && dilation[1] == 1 &&
        padding[0] == 1 && padding[1] == 1) {  // the stride-2 forward reads through out-of-range-is-zero buffer loads too (nk_conv_s2fwd.h)
        rc = s2f_launch(dev, x, w, bias, y, g.N, g.Cin, g.Cout, x_shape[2], x_shape[3], g.out[1], g.out[2], 1,
                        2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK, &taken);
        if (rc || taken) return rc;
    }
    // No kernel folds the padding for this geometry - or the rules in force decline (a caller that asked nk_conv_padding_folds at
    // graph-build time may find the knobs changed by the time it runs): the two nodes the entry stands for, Pad::forward into the
    // device's operand scratch (pad/zero/mod.rs:5-31), then the convolution on the copy - the bits of the two-node path.
    float* xp = nullptr;
    int pshape[5];
    rc = padded_copy(dev, nd, x, x_shape, padding, &xp, pshape);
    if (rc) return rc;
    return conv_fwd(dev, nd, xp, pshape, w, w_shape, bias, y, stride, dilation, groups);
}

int conv_bwd_kernel_padded(nk_device* dev, int nd, float* dw, float* db, const int* w_shape, const float* gy, const float* x, const int* x_shape,
                           const int* padding, const int* stride, const int* dilation, int groups, int assign, int assign_b) {
    NK_USE(dev);
    ConvGeom g;
    bool fold = false;
    int rc = fold_geom(nd, x_shape, padding, w_shape, stride, dilation, groups, &g, &fold);
    if (rc) return rc;
    NK_CHECK(dw && gy && x, "null pointer in nk_conv_bwd_kernel_bias_padded");
    bool taken = false;
    if (fold && dev->tune_conv_winograd != 0 && dev->tune_conv_wino_dw != 0)
        rc = wino_dw_launch(dev, gy, x, dw, db, g.N, g.Cin, g.Cout, g.in[1], g.in[2], assign, assign_b, 2.0 * g.N * (double)g.Cout * g.L * g.Cg * g.KK,
                            &taken, padding[0], padding[1], true);
    if (rc) return rc;
    if (taken) return NK_OK;
    float* xp = nullptr;  // as in conv_fwd_padded: pad, then the kernel gradient against the copy (the two-node path's bits)
    int pshape[5];
    rc = padded_copy(dev, nd, x, x_shape, padding, &xp, pshape);
    if (rc) return rc;
    return conv_bwd_kernel(dev, nd, dw, w_shape, gy, xp, pshape, stride, dilation, groups, assign, db, assign_b);
}

}  // namespace

extern "C" {

int nk_conv_padding_folds(nk_device* dev, int nd, const int* x_shape, const int* padding, const int* w_shape, const int* stride,
                          const int* dilation, int groups, int* folds) {
    return conv_padding_folds(dev, nd, x_shape, padding, w_shape, stride, dilation, groups, folds);
}
int nk_conv_bias_fwd_padded(nk_device* dev, int nd, const float* x, const int* x_shape, const int* padding, const float* w, const int* w_shape,
                            const float* bias, float* y, const int* stride, const int* dilation, int groups) {
    return conv_fwd_padded(dev, nd, x, x_shape, padding, w, w_shape, bias, y, stride, dilation, groups);
}
int nk_conv_bwd_kernel_bias_padded(nk_device* dev, int nd, float* dw, float* db, const int* w_shape, const float* gy, const float* x,
                                   const int* x_shape, const int* padding, const int* stride, const int* dilation, int groups, int assign_dw,
                                   int assign_db) {
    return conv_bwd_kernel_padded(dev, nd, dw, db, w_shape, gy, x, x_shape, padding, stride, dilation, groups, assign_dw ? 1 : 0, assign_db ? 1 : 0);
}

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
int nk_conv_bwd_kernel_bias(nk_device* dev, int nd, float* dw, float* db, const int* w_shape, const float* gy, const float* x,
                            const int* x_shape, const int* stride, const int* dilation, int groups, int assign_dw, int assign_db) {
    NK_CHECK(db, "null bias gradient in nk_conv_bwd_kernel_bias");
    return conv_bwd_kernel(dev, nd, dw, w_shape, gy, x, x_shape, stride, dilation, groups, assign_dw ? 1 : 0, db, assign_db ? 1 : 0);
}

}  // extern "C"
