// Direct convolution kernels for few channels per group (depthwise / small grouped convolutions)  -  part of the convolution translation unit (included by nk_conv.hip inside its anonymous
// namespace; not a stand-alone header).
#pragma once

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

// Strided backward-input (some stride > 1, every dilation 1) WITHOUT a division per tap (round 6): of the taps of an axis only those
// with k = (p + pad) mod stride reach position p, so a lane walks its OWN residue class - k = r, r + s, r + 2 s, ... against output
// coordinates o0, o0 - 1, o0 - 2, ... - where the general kernel above tests every tap with `%` and `/` (two integer divisions of
// ~ 20 instructions each per tap and position) and masks three of four at stride 2.  The residue differs from lane to lane, so the
// weight is a per-lane load (the whole kernel tensor is a few KB: L1 hits) instead of a scalar broadcast.  Same (m, k0, k1, k2)
// accumulation order per element as the general kernel - the skipped taps contributed exact zeros: bit-identical results.
// The 7 x 7 / stride-2 stem (3 -> 64 channels at 224 x 224, N = 128): 3.8 ms -> see profiles/r06_conv_shapes.jsonl.
template <int PT>
__global__ __launch_bounds__(256) void conv_direct_bwd_input_strided_kernel(float* __restrict__ dx, const float* __restrict__ gy,
                                                                            const float* __restrict__ w, ConvGeom g) {
    const int nc = blockIdx.x, cabs = nc % g.Cin, n = nc / g.Cin, grp = cabs / g.Cg, ci = cabs - grp * g.Cg;
    const int p0 = blockIdx.y * (256 * PT) + threadIdx.x;
    int r[PT][3], o0[PT][3];  // per position and axis: the residue class (first tap) and the output coordinate that tap meets
    float acc[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        int pos = p0 + 256 * i;
        pos = pos < g.uinplane ? pos : 0;
        int pc[3];
        pc[2] = pos % g.uin[2] + g.pad[2]; pos /= g.uin[2];
        pc[1] = pos % g.uin[1] + g.pad[1];
        pc[0] = pos / g.uin[1] + g.pad[0];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            r[i][d] = pc[d] % g.stride[d];
            o0[i][d] = (pc[d] - r[i][d]) / g.stride[d];
        }
        acc[i] = 0.f;
    }
    const int J0 = (g.k[0] + g.stride[0] - 1) / g.stride[0], J1 = (g.k[1] + g.stride[1] - 1) / g.stride[1],
              J2 = (g.k[2] + g.stride[2] - 1) / g.stride[2];  // taps per residue class and axis, at most
    const float* gs = gy + ((long long)n * g.Cout + (long long)grp * g.Mg) * g.L;
    for (int m = 0; m < g.Mg; ++m) {
        const float* gc = gs + (long long)m * g.L;
        const float* wc = w + ((long long)(grp * g.Mg + m) * g.Cg + ci) * g.KK;
        for (int j0 = 0; j0 < J0; ++j0)
            for (int j1 = 0; j1 < J1; ++j1) {
                int rbase[PT], wbase[PT];  // gradient row offset and weight row offset, or -1
#pragma unroll
                for (int i = 0; i < PT; ++i) {
                    const int k0 = r[i][0] + j0 * g.stride[0], k1 = r[i][1] + j1 * g.stride[1];
                    const int a = o0[i][0] - j0, b = o0[i][1] - j1;
                    const bool ok = k0 < g.k[0] && k1 < g.k[1] && a >= 0 && a < g.out[0] && b >= 0 && b < g.out[1];
                    rbase[i] = ok ? (a * g.out[1] + b) * g.out[2] : -1;
                    wbase[i] = (k0 * g.k[1] + k1) * g.k[2];
                }
                for (int j2 = 0; j2 < J2; ++j2) {
#pragma unroll
                    for (int i = 0; i < PT; ++i) {  // branch-free: clamped (always valid) addresses, the product masked
                        const int k2 = r[i][2] + j2 * g.stride[2], c = o0[i][2] - j2;
                        const bool ok = rbase[i] >= 0 && k2 < g.k[2] && c >= 0 && c < g.out[2];
                        const float gv = gc[ok ? rbase[i] + c : 0];
                        const float wv = wc[ok ? wbase[i] + k2 : 0];
                        acc[i] = fmaf(ok ? wv : 0.f, ok ? gv : 0.f, acc[i]);  // (a tap that does not reach the position adds an exact 0, whatever its weight)
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
// BUF (round 6, tensors below 2 GB): the segment comes in as 8-byte buffer loads (merged to 16 + 8 / 16 + 16 bytes, any 4-byte boundary) from
// gr + c0 instead of 4 + TK2 - 1 masked scalar loads - the depthwise / grouped input gradients are bound by the rate at which a CU issues
// loads, not by HBM; an address before / past the tensor reads 0 through the descriptor, elements of a neighbouring row are masked as before.
template <int TK1, int TK2, bool BUF = false>
__global__ __launch_bounds__(256) void conv_direct_bwd_input_rows_kernel(float* __restrict__ dx, const float* __restrict__ gy,
                                                                         const float* __restrict__ w, ConvGeom g) {
    const int nc = blockIdx.x, cabs = nc % g.Cin, n = nc / g.Cin, grp = cabs / g.Cg, ci = cabs - grp * g.Cg;
    const int qpr = g.uin[2] / 4;
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= g.uin[1] * qpr) return;
    const int a = q / qpr, b = (q - a * qpr) * 4;
    const int pa = a + g.pad[1], c0 = b + g.pad[2] - (TK2 - 1);  // gradient column of segment element 0
    const float* gs = gy + ((long long)n * g.Cout + (long long)grp * g.Mg) * g.L;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)gy, 0, BUF ? (int)((long long)g.N * g.Cout * g.L * 4) : 0, 0x00020000);
    const long long gs_off = ((long long)n * g.Cout + (long long)grp * g.Mg) * g.L;
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
            // element offset of segment element 0 in gy; negative only in the tensor's very first row (n = 0, first channel, row 0) for the
            // first quad of a row: a byte offset that wrapped below zero does not wrap back inside a multi-dword load, so the ONE wave of
            // the launch that holds such a lane takes the scalar loads for that row (wave-uniform branch)
            const int eoff = (int)(gs_off + (long long)m * g.L) + ra * g.out[2] + c0;
            if (BUF && !__any(rowok && eoff < 0)) {
                static_assert(!BUF || TK2 == 3 || TK2 == 5, "segments of 6 or 8 floats");
                const unsigned vo = rowok ? (unsigned)eoff * 4u : 0x80000000u;  // (a row that does not exist: out of range, zeros)
                float raw[8];
#pragma unroll
                for (int k = 0; k < (4 + TK2 - 1 + 1) / 2; ++k) {  // (the compiler merges neighbours into 16-byte loads; any 4-byte boundary)
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(grs, vo + 8u * k, 0, 0);
                    static_assert(sizeof(v) == 8, "two dwords");
                    const float2 f = __builtin_bit_cast(float2, v);
                    raw[2 * k] = f.x; raw[2 * k + 1] = f.y;
                }
#pragma unroll
                for (int j = 0; j < 4 + TK2 - 1; ++j) {
                    const int col = c0 + j;
                    seg[j] = rowok && col >= 0 && col < g.out[2] ? raw[j] : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4 + TK2 - 1; ++j) {
                    const int col = c0 + j;
                    const bool ok = rowok && col >= 0 && col < g.out[2];
                    const float v = gr[ok ? col : 0];
                    seg[j] = ok ? v : 0.f;
                }
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

