// Input gradient of the 3x3 / STRIDE 2 / dilation 1 / groups 1 convolution (ConvolutionBackwardInput, node/convolution/mod.rs:146-189,
// 256-274; the down-sampling layers of a CNN), all four stride phases of a tile in ONE block walk - part of the convolution translation
// unit (included by nk_conv.hip inside its anonymous namespace, after nk_conv_winograd.h).
//
// Why (round 6): the per-phase implicit GEMMs of nk_conv_fast.h ran this pass at 0.33 of the f32 MFMA peak (3x3 s2 64 -> 128 at 56 x 56:
// 287 us against 176 us for its forward).  A stride-2 input gradient splits by the parity of the input coordinate into four phases with
// 1, 2, 2 and 4 of the nine taps: four GEMMs whose reductions are 4 - 16 k-tiles long (per-tile fixed cost dominates) and whose outputs
// interleave in dX element by element (every store instruction fills half of each line).
//
// The fused form.  Input pixel (2a + ry, 2b + rx) of "super-pixel" (a, b) receives, with padding p per axis,
//   dx[n][ci][2a + ry][2b + rx] = sum over co and the taps (ky, kx) with ky = ry + p, kx = rx + p (mod 2) of
//                                 w[co][ci][ky][kx] * gy[n][co][a + (ry + p - ky) / 2][b + (rx + p - kx) / 2],
// so the four pixels of a super-pixel read the SAME 2 x 2 neighbourhood of gy (origin a + o, b + o with o = 0 for p = 1, o = -1 for
// p = 0) through nine (tap -> phase, neighbour) pairs.  A lane owns one super-pixel (MFMA column) and 16 output channels for all four
// phases: 4 accumulator tiles = 64 registers; per chunk of KC reduction channels a block stages the four neighbour planes of its 32
// super-pixels once (raw values, [plane][channel / 4][tile][channel % 4] as the Winograd kernel's V image: one ds_read_b128 = the B
// values of four MFMA steps) and issues nine products - tap (ky, kx): A = the tap's weights in MFMA operand order straight from L2
// (a reorder kernel writes them that way per launch), B = the tap's neighbour plane, C = the tap's phase.  The output is the 2 x 2 block
// of each super-pixel: float2 stores of whole row pairs, adjacent lanes adjacent super-pixels - every line written once, whole.
//   WIDE    four waves x 32 channels = 128 output channels, chunks of 32;  NARROW  two waves = 64 output channels, chunks of 16
// Several blocks per CU (64 accumulator registers, 16 - 32 KB of LDS each): the hardware interleaves them, the loop needs no hand
// pipelining beyond the register-staged loads of the next chunk.  Out-of-range neighbours (a + 1 = Ho at the bottom / right border for
// p = 1, a - 1 = -1 at the top / left for p = 0) and tiles past the end have buffer offset 0x80000000: the hardware returns 0.
// Summation order: per (phase, channel, pixel) one fma chain over (reduction channel chunk, tap in ky-major order, channel) -
// deterministic; not the per-phase kernels' order: equal to them to contraction tolerance, exact on integer data.
#pragma once

struct S2dxArgs {
    const float* gy;  // (N, Ck = Cout, Ho, Wo)
    const float* u;   // weights in fragment order (s2dx_weights_kernel)
    float* dx;        // (N, Cm = Cin, H, W): the UNPADDED input's gradient, H and W even
    int N, Ck, Cm, Ho, Wo, H, W, TY, TX;
    long long P;      // N * TY * TX super-pixels
    int nchunk, assign;
    int src_bytes, u_bytes, dst_bytes;
};

// U in MFMA A-operand order, as wino_weights_kernel's with the nine taps in place of the sixteen xi:
//   u[((((ch * 9 + tap) * CBT + cbt) * (KC / 8) + j) * 64 + lane) * 4 + (s & 3)],  lane = r + 32 h: output channel (ci) 32 cbt + r,
//   reduction channel (co) ch * KC + (KC / 2) h + s, s = 4 j .. 4 j + 3;  value w[co][ci][tap]  (w is (Ck, Cm, 3, 3))
__global__ void s2dx_weights_kernel(float* __restrict__ u, const float* __restrict__ w, int Cm, int Ck, int KC) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cm * Ck) return;
    const int ci = idx / Ck, co = idx % Ck;
    const int KH = KC / 2, CBT = Cm / 32;
    const int cbt = ci / 32, r = ci % 32, ch = co / KC, kk = co % KC, h = kk / KH, s = kk % KH, j = s / 4, tq = s % 4, lane = r + 32 * h;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        u[((((long long)(ch * 9 + tap) * CBT + cbt) * (KC / 8) + j) * 64 + lane) * 4 + tq] = w[((long long)co * Cm + ci) * 9 + tap];
}

// per axis, padding p: kernel index k -> phase bit r = (k + p) & 1 and neighbour index (r + p - k) / 2 - o in {0, 1}, o = p == 1 ? 0 : -1
__host__ __device__ constexpr int s2dx_phase(int k, int p) { return (k + p) & 1; }
__host__ __device__ constexpr int s2dx_nb(int k, int p) { return (s2dx_phase(k, p) + p - k) / 2 - (p == 1 ? 0 : -1); }

template <int CB, int KC, int PAD>
__global__ __launch_bounds__(64 * CB, 2) void s2dx_kernel(S2dxArgs a) {
    constexpr int PT = 32;       // super-pixels per block
    constexpr int NJ = KC / 8;   // groups of four MFMA steps per tap and chunk
    constexpr int KQ = KC / 4;   // channel quads per chunk
    constexpr int VBUF = 4 * KC * PT;
    static_assert(KQ * PT == 64 * CB, "one (super-pixel, channel quad) per thread and chunk");
    static_assert(s2dx_nb(0, PAD) >= 0 && s2dx_nb(0, PAD) <= 1 && s2dx_nb(1, PAD) >= 0 && s2dx_nb(1, PAD) <= 1 && s2dx_nb(2, PAD) >= 0 && s2dx_nb(2, PAD) <= 1,
                  "every tap reads the 2 x 2 neighbourhood");
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int pl = t % PT, kq0 = t / PT;
    const int CBT = a.Cm / 32, cbg = blockIdx.y * CB + wid;
    const int gplane = a.Ho * a.Wo, per = a.TY * a.TX;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, a.u_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dx, 0, a.dst_bytes, 0x00020000);
    const int ustep16 = CBT * NJ * 64 * 16;  // bytes from tap to tap + 1
    const unsigned uvoff = (unsigned)(cbg * NJ * 64 + lane) * 16u;
    const int gplane4 = gplane * 4;

    // ---- staging: this thread's super-pixel of the block and its channel quad: the 2 x 2 neighbourhood, four channels each
    unsigned poff[4];
    {
        const long long p = (long long)blockIdx.x * PT + pl;
        const bool pvalid = p < a.P;
        const long long pv = pvalid ? p : 0;
        const int n = (int)(pv / per), rem = (int)(pv - (long long)n * per), ty = rem / a.TX, tx = rem - ty * a.TX;
        constexpr int O = PAD == 1 ? 0 : -1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = ty + O + i, ox = tx + O + j;
                const bool ok = pvalid && (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo;
                poff[2 * i + j] = ok ? (unsigned)((n * a.Ck + 4 * kq0) * gplane + oy * a.Wo + ox) * 4u : 0x80000000u;
            }
    }
    float4 d[4];
    auto load = [&](int ch) {
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            d[pos].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, poff[pos], (ch * KC + 0) * gplane4, 0));
            d[pos].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, poff[pos], (ch * KC + 1) * gplane4, 0));
            d[pos].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, poff[pos], (ch * KC + 2) * gplane4, 0));
            d[pos].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, poff[pos], (ch * KC + 3) * gplane4, 0));
        }
    };
    auto stage = [&](float* v) {
        float4* vp = reinterpret_cast<float4*>(v) + kq0 * PT + pl;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) vp[pos * KQ * PT] = d[pos];
    };

    nkmma::f32x16 acc[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ph][e] = 0.f;

    load(0);
    stage(V);
    __syncthreads();
    for (int ch = 0; ch < a.nchunk; ++ch) {
        const float* const vbase = V + (ch & 1) * VBUF;
        if (ch + 1 < a.nchunk) load(ch + 1);  // in flight under this chunk's MFMAs
        const float4* const vcur = reinterpret_cast<const float4*>(vbase) + (NJ * h) * PT + c;
        float4 af[2][NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            af[0][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff + j * 1024, (ch * 9) * ustep16, 0));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int phase = 2 * s2dx_phase(ky, PAD) + s2dx_phase(kx, PAD), plane = 2 * s2dx_nb(ky, PAD) + s2dx_nb(kx, PAD);
            if (tap + 1 < 9) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    af[(tap + 1) & 1][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff + j * 1024, (ch * 9 + tap + 1) * ustep16, 0));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 bv = vcur[(plane * KQ + j) * PT];
                const float4 av = af[tap & 1][j];
                acc[phase] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[phase], 0, 0, 0);
                acc[phase] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[phase], 0, 0, 0);
                acc[phase] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[phase], 0, 0, 0);
                acc[phase] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[phase], 0, 0, 0);
            }
        }
        if (ch + 1 < a.nchunk) stage(V + ((ch + 1) & 1) * VBUF);
        __syncthreads();  // the next chunk's image is complete; everybody has read this one
    }

    // ---- output: lane (c, h) owns super-pixel c of the block and channels 32 cbg + 8 Q + 4 h + el (MFMA C layout, e = 4 Q + el):
    // the 2 x 2 block, one float2 per row.  Loads (the `+=` form) before the stores of the same quarter.
    const long long p = (long long)blockIdx.x * PT + c;
    const bool pvalid = p < a.P;
    const long long pv = pvalid ? p : 0;
    const int n = (int)(pv / per), rem = (int)(pv - (long long)n * per), ty = rem / a.TX, tx = rem - ty * a.TX;
    const int oplane = a.H * a.W;
    const unsigned ovoff = pvalid ? (unsigned)((n * a.Cm + 32 * cbg + 4 * h) * oplane + 2 * ty * a.W + 2 * tx) * 4u : 0x80000000u;
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) {
        float2 y[4][2];
#pragma unroll
        for (int el = 0; el < 4; ++el)
#pragma unroll
            for (int r = 0; r < 2; ++r) y[el][r] = make_float2(acc[2 * r][4 * Q + el], acc[2 * r + 1][4 * Q + el]);
        if (!a.assign) {
            float2 old[4][2];
#pragma unroll
            for (int el = 0; el < 4; ++el)
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    old[el][r] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(drs, ovoff, ((8 * Q + el) * oplane + r * a.W) * 4, 0));
#pragma unroll
            for (int el = 0; el < 4; ++el)
#pragma unroll
                for (int r = 0; r < 2; ++r) { y[el][r].x += old[el][r].x; y[el][r].y += old[el][r].y; }
        }
#pragma unroll
        for (int el = 0; el < 4; ++el)
#pragma unroll
            for (int r = 0; r < 2; ++r)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wino_u2, y[el][r]), drs, ovoff, ((8 * Q + el) * oplane + r * a.W) * 4, 0);
    }
}

// Host side.  `taken` = false: not a case for this path (the caller goes on to the per-phase kernels).
//   gy (N, Cout, Ho, Wo), w (Cout, Cin, 3, 3), dx (N, Cin, H, W) with H, W even; pad = the zero padding folded in (0 or 1, both axes alike)
int s2dx_launch(nk_device* dev, const float* gy, const float* w, float* dx, int N, int Cout, int Cin, int Ho, int Wo, int H, int W, int pad,
                int assign, double flop, bool* taken) {
    *taken = false;
    if (dev->tune_conv_s2dx == 0) return NK_OK;
    if (pad < 0 || pad > 1 || H < 2 || W < 2 || H % 2 != 0 || W % 2 != 0) return NK_OK;
    if (Ho != (H + 2 * pad - 3) / 2 + 1 || Wo != (W + 2 * pad - 3) / 2 + 1) return NK_OK;
    const bool wide_ok = Cin % 128 == 0 && Cout % 32 == 0, narrow_ok = Cin % 64 == 0 && Cout % 16 == 0;
    if (!wide_ok && !narrow_ok) return NK_OK;
    if (!al16(dx)) return NK_OK;
    const long long P = (long long)N * (H / 2) * (W / 2);
    // Block shape: wide (four waves, 128 channels, half the weight-fragment traffic per MFMA) where there are blocks in plenty; narrow
    // (two waves, 64 channels) when the wide grid is only a few blocks per CU - 3x3 s2 128 -> 256 at 28 x 28 is 784 wide blocks = 3.06 per
    // CU (a fourth, nearly empty round: 192 us) or 1568 narrow ones.  Knob values 2 / 3 force narrow / wide.
    const bool wide = dev->tune_conv_s2dx == 3 ? wide_ok : dev->tune_conv_s2dx == 2 ? !narrow_ok
                      : wide_ok && (!narrow_ok || (P + 31) / 32 * (Cin / 128) >= 8LL * dev->num_cus);
    const int KC = wide ? 32 : 16, CM = wide ? 128 : 64, PT = 32;
    const long long src_bytes = (long long)N * Cout * Ho * Wo * 4, dst_bytes = (long long)N * Cin * H * W * 4, u_bytes = 9LL * Cin * Cout * 4;
    if (P >= (1LL << 30) || src_bytes >= 0x7fffffffLL || dst_bytes >= 0x7fffffffLL || u_bytes >= 0x7fffffffLL) return NK_OK;
    const long long npb = (P + PT - 1) / PT;
    // by rule: from one block per CU on (below that the per-phase kernels' k-split of a small grid fills the chip better)
    if (dev->tune_conv_s2dx < 0 && npb * (Cin / CM) < dev->num_cus) return NK_OK;
    void* ws = nullptr;
    int rc = nk_workspace(dev, (size_t)u_bytes, &ws);
    if (rc) return rc;
    rc = nk_prof_start(dev, NK_KERNEL_CONV, flop);
    if (rc) return rc;
    hipLaunchKernelGGL(s2dx_weights_kernel, dim3((unsigned)((Cin * Cout + 255) / 256)), dim3(256), 0, dev->compute, (float*)ws, w, Cin, Cout, KC);
    NK_LAUNCH_CHECK();
    S2dxArgs a{};
    a.gy = gy; a.u = (const float*)ws; a.dx = dx;
    a.N = N; a.Ck = Cout; a.Cm = Cin; a.Ho = Ho; a.Wo = Wo; a.H = H; a.W = W; a.TY = H / 2; a.TX = W / 2; a.P = P;
    a.nchunk = Cout / KC; a.assign = assign;
    a.src_bytes = (int)src_bytes; a.u_bytes = (int)u_bytes; a.dst_bytes = (int)dst_bytes;
    const dim3 grid((unsigned)npb, (unsigned)(Cin / CM));
    if (wide && pad == 1) hipLaunchKernelGGL((s2dx_kernel<4, 32, 1>), grid, dim3(256), 0, dev->compute, a);
    else if (wide) hipLaunchKernelGGL((s2dx_kernel<4, 32, 0>), grid, dim3(256), 0, dev->compute, a);
    else if (pad == 1) hipLaunchKernelGGL((s2dx_kernel<2, 16, 1>), grid, dim3(128), 0, dev->compute, a);
    else hipLaunchKernelGGL((s2dx_kernel<2, 16, 0>), grid, dim3(128), 0, dev->compute, a);
    NK_LAUNCH_CHECK();
    *taken = true;
    return nk_prof_stop(dev);
}
