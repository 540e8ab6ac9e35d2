// Forward of the 3x3 / STRIDE 2 / dilation 1 / groups 1 convolution (Convolution::forward, node/convolution/mod.rs:85-123) as nine tap
// products on staged tap planes - the forward twin of nk_conv_s2dx.h, part of the convolution translation unit (included by nk_conv.hip
// inside its anonymous namespace, after nk_conv_s2dx.h).
//
// Why (round 6): the implicit GEMM of nk_conv_fast.h runs the down-sampling layers' forward at 0.54 - 0.57 of the f32 MFMA peak (stride-2
// gathers, 18 - 36 k-tiles per tile); the block form that took their input gradient from 0.33 to 0.6 - 0.7 - several small blocks per CU,
// operands staged once per chunk, weights as MFMA fragments straight from L2, no hand pipelining - applies to the forward unchanged:
// a lane owns one OUTPUT position (MFMA column) and 16 output channels (one accumulator tile); per chunk of KC input channels a block of
// 32 positions stages the nine tap planes x[n][ci][2a + ky - p][2b + kx - p] (raw values, [tap][channel / 4][position][channel % 4]: one
// ds_read_b128 = the B values of four MFMA steps) and issues nine products with the taps' weight fragments.  Elements outside the image
// (the module's zero padding p = 1 folded in, or positions past the end) sit at buffer offset 0x80000000: zeros.
//   WIDE  four waves x 32 output channels = 128, chunks of 32;   NARROW  two waves = 64 output channels, chunks of 16
// Summation order: per (channel, position) one fma chain over (chunk, tap in ky-major order, channel) - deterministic; equal to the
// implicit-GEMM kernel to contraction tolerance, exact on integer data.
#pragma once

struct S2fArgs {
    const float* x;     // (N, Ck = Cin, H, W): the input as the caller holds it (padded by the caller: pad = 0; unpadded: pad = 1)
    const float* u;     // weights in fragment order (s2f_weights_kernel)
    float* y;           // (N, Cm = Cout, Ho, Wo)
    const float* bias;  // optional, per output channel
    int N, Ck, Cm, H, W, Ho, Wo, pad;
    long long P;        // N * Ho * Wo output positions
    int nchunk;
    int src_bytes, u_bytes, dst_bytes;
};

// u[((((ch * 9 + tap) * CBT + cbt) * (KC / 8) + j) * 64 + lane) * 4 + (s & 3)],  lane = r + 32 h: output channel (co) 32 cbt + r,
// reduction channel (ci) ch * KC + (KC / 2) h + s, s = 4 j .. 4 j + 3;  value w[co][ci][tap]  (w is (Cm, Ck, 3, 3))
__global__ void s2f_weights_kernel(float* __restrict__ u, const float* __restrict__ w, int Cm, int Ck, int KC) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cm * Ck) return;
    const int co = idx / Ck, ci = idx % Ck;
    const int KH = KC / 2, CBT = Cm / 32;
    const int cbt = co / 32, r = co % 32, ch = ci / KC, kk = ci % KC, h = kk / KH, s = kk % KH, j = s / 4, tq = s % 4, lane = r + 32 * h;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        u[((((long long)(ch * 9 + tap) * CBT + cbt) * (KC / 8) + j) * 64 + lane) * 4 + tq] = w[((long long)co * Ck + ci) * 9 + tap];
}

template <int CB, int KC>
__global__ __launch_bounds__(64 * CB, 2) void s2f_kernel(S2fArgs a) {
    constexpr int PT = 32, NJ = KC / 8, KQ = KC / 4, VBUF = 9 * KC * PT;
    static_assert(KQ * PT == 64 * CB, "one (position, channel quad) per thread and chunk");
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int pl = t % PT, kq0 = t / PT;
    const int CBT = a.Cm / 32, cbg = blockIdx.y * CB + wid;
    const int xplane = a.H * a.W, per = a.Ho * a.Wo;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, a.u_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.dst_bytes, 0x00020000);
    const int ustep16 = CBT * NJ * 64 * 16;  // bytes from tap to tap + 1
    const unsigned uvoff = (unsigned)(cbg * NJ * 64 + lane) * 16u;
    const int xplane4 = xplane * 4;

    unsigned poff[9];  // byte offset of tap (ky, kx) of this thread's position, channel 4 kq0 of a chunk; 0x80000000 outside the image
    {
        const long long p = (long long)blockIdx.x * PT + pl;
        const bool pvalid = p < a.P;
        const long long pv = pvalid ? p : 0;
        const int n = (int)(pv / per), rem = (int)(pv - (long long)n * per), oa = rem / a.Wo, ob = rem - oa * a.Wo;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oa + ky - a.pad, ix = 2 * ob + kx - a.pad;
                const bool ok = pvalid && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                poff[3 * ky + kx] = ok ? (unsigned)((n * a.Ck + 4 * kq0) * xplane + iy * a.W + ix) * 4u : 0x80000000u;
            }
    }
    float4 d[9];
    auto load = [&](int ch) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            d[tap].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, poff[tap], (ch * KC + 0) * xplane4, 0));
            d[tap].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, poff[tap], (ch * KC + 1) * xplane4, 0));
            d[tap].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, poff[tap], (ch * KC + 2) * xplane4, 0));
            d[tap].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, poff[tap], (ch * KC + 3) * xplane4, 0));
        }
    };
    auto stage = [&](float* v) {
        float4* vp = reinterpret_cast<float4*>(v) + kq0 * PT + pl;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) vp[tap * KQ * PT] = d[tap];
    };

    nkmma::f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    load(0);
    stage(V);
    __syncthreads();
    for (int ch = 0; ch < a.nchunk; ++ch) {
        if (ch + 1 < a.nchunk) load(ch + 1);  // in flight under this chunk's MFMAs
        const float4* const vcur = reinterpret_cast<const float4*>(V + (ch & 1) * VBUF) + (NJ * h) * PT + c;
        float4 af[2][NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            af[0][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff + j * 1024, (ch * 9) * ustep16, 0));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    af[(tap + 1) & 1][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff + j * 1024, (ch * 9 + tap + 1) * ustep16, 0));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 bv = vcur[(tap * KQ + j) * PT];
                const float4 av = af[tap & 1][j];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            }
        }
        if (ch + 1 < a.nchunk) stage(V + ((ch + 1) & 1) * VBUF);
        __syncthreads();
    }

    // ---- output: lane (c, h) owns position c of the block and channels 32 cbg + 8 Q + 4 h + el (MFMA C layout, e = 4 Q + el); a store
    // instruction writes one channel's 32 consecutive positions
    const long long p = (long long)blockIdx.x * PT + c;
    const bool pvalid = p < a.P;
    const long long pv = pvalid ? p : 0;
    const int n = (int)(pv / per), rem = (int)(pv - (long long)n * per);
    const unsigned ovoff = pvalid ? (unsigned)((n * a.Cm + 32 * cbg + 4 * h) * per + rem) * 4u : 0x80000000u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int chn = 8 * (e >> 2) + (e & 3);
        const float bv = a.bias ? a.bias[32 * cbg + 4 * h + chn] : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a.bias ? acc[e] + bv : acc[e]), drs, ovoff, chn * per * 4, 0);
    }
}

// Host side.  `taken` = false: not a case for this path.  x (N, Cin, H, W) as the caller holds it; pad = 0 (the caller padded) or 1 (folded)
int s2f_launch(nk_device* dev, const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout, int H, int W, int Ho, int Wo,
               int pad, double flop, bool* taken) {
    *taken = false;
    if (dev->tune_conv_s2dx == 0) return NK_OK;   // (one knob for the stride-2 pair of kernels)
    if (pad < 0 || pad > 1 || Ho < 1 || Wo < 1) return NK_OK;
    if (Ho != (H + 2 * pad - 3) / 2 + 1 || Wo != (W + 2 * pad - 3) / 2 + 1) return NK_OK;
    const bool wide_ok = Cout % 128 == 0 && Cin % 32 == 0, narrow_ok = Cout % 64 == 0 && Cin % 16 == 0;
    if (!wide_ok && !narrow_ok) return NK_OK;
    const long long P = (long long)N * Ho * Wo;
    // Block shape and rule.  Nine tap planes per chunk are staged by EVERY channel block of a position block, so the forward pays for
    // narrow blocks and small grids where its input-gradient twin (four planes) does not.  Measured, N = 128, same box (benchmarks/ab_s2dx.py
    // 128 fwd; implicit GEMM / narrow / wide, us): 64 -> 128 at 56 x 56 176 / 175 / **155**; 128 -> 256 at 28 x 28 **164** / 189 / 171;
    // 256 -> 512 at 14 x 14 **193** / 214 / 197; 64 -> 64 at 112 x 112 **345** / 373 / -.  By rule therefore only with wide blocks and
    // eight or more of them per CU (the first of the four); knob 1 takes it whenever the shape allows, 2 / 3 force narrow / wide.
    const bool wide = dev->tune_conv_s2dx == 2 ? !narrow_ok : wide_ok;
    const int KC = wide ? 32 : 16, CM = wide ? 128 : 64, PT = 32;
    const long long src_bytes = (long long)N * Cin * H * W * 4, dst_bytes = (long long)N * Cout * Ho * Wo * 4, u_bytes = 9LL * Cin * Cout * 4;
    if (P >= (1LL << 30) || src_bytes >= 0x7fffffffLL || dst_bytes >= 0x7fffffffLL || u_bytes >= 0x7fffffffLL) return NK_OK;
    const long long npb = (P + PT - 1) / PT;
    if (dev->tune_conv_s2dx < 0 && (!wide || npb * (Cout / CM) < 8LL * dev->num_cus)) return NK_OK;
    void* ws = nullptr;
    int rc = nk_workspace(dev, (size_t)u_bytes, &ws);
    if (rc) return rc;
    rc = nk_prof_start(dev, NK_KERNEL_CONV, flop);
    if (rc) return rc;
    hipLaunchKernelGGL(s2f_weights_kernel, dim3((unsigned)((Cin * Cout + 255) / 256)), dim3(256), 0, dev->compute, (float*)ws, w, Cout, Cin, KC);
    NK_LAUNCH_CHECK();
    S2fArgs a{};
    a.x = x; a.u = (const float*)ws; a.y = y; a.bias = bias;
    a.N = N; a.Ck = Cin; a.Cm = Cout; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.pad = pad; a.P = P;
    a.nchunk = Cin / KC;
    a.src_bytes = (int)src_bytes; a.u_bytes = (int)u_bytes; a.dst_bytes = (int)dst_bytes;
    const dim3 grid((unsigned)npb, (unsigned)(Cout / CM));
    if (wide) hipLaunchKernelGGL((s2f_kernel<4, 32>), grid, dim3(256), 0, dev->compute, a);
    else hipLaunchKernelGGL((s2f_kernel<2, 16>), grid, dim3(128), 0, dev->compute, a);
    NK_LAUNCH_CHECK();
    *taken = true;
    return nk_prof_stop(dev);
}
