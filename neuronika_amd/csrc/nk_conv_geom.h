// Convolution geometry, table pre-kernels and column helpers  -  part of the convolution translation unit (included by nk_conv.hip inside its anonymous
// namespace; not a stand-alone header).
#pragma once


struct ConvGeom {
    int N, Cin, Cout, groups, Cg, Mg;  // Cg = Cin/groups, Mg = Cout/groups
    int in[3], out[3], k[3], stride[3], dil[3];  // padded in front with 1s to 3 spatial dims
    int inplane, L, KK;                           // prod(in), prod(out), prod(k)
    const float* bias;                            // forward: optional per-output-channel bias added in the epilogue
    int assign;                                   // backward: write instead of `+=` (destination's zero fill pending)
    // backward-input through a zero Pad node: dX has the UNPADDED extents `uin` and input coordinate q of dX is
    // coordinate q + pad of the (virtual) padded input `in`.  pad = 0, uin = in otherwise.
    int uin[3], pad[3], uinplane;
};

// ---- tables (tiny pre-kernels into the device workspace) ---------------------------------------
// koff(k), k = ci*KK + kidx : input offset of kernel element k relative to the window origin
__device__ __forceinline__ int conv_koff(const ConvGeom& g, int k) {
    const int ci = k / g.KK;
    int rem = k % g.KK;
    const int k2 = rem % g.k[2]; rem /= g.k[2];
    const int k1 = rem % g.k[1];
    const int k0 = rem / g.k[1];
    return ci * g.inplane + (k0 * g.dil[0] * g.in[1] + k1 * g.dil[1]) * g.in[2] + k2 * g.dil[2];
}
__global__ void conv_koff_kernel(int* __restrict__ koff, ConvGeom g) {
    const int K = g.Cg * g.KK;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) koff[k] = conv_koff(g, k);
}
// ktab[k'], k' = co*KK + kidx : {co*L, k0*dil0, k1*dil1, k2*dil2}
__global__ void conv_ktab_kernel(int4* __restrict__ ktab, ConvGeom g) {
    const int K = g.Mg * g.KK;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int co = k / g.KK;
        int rem = k % g.KK;
        const int k2 = rem % g.k[2]; rem /= g.k[2];
        const int k1 = rem % g.k[1];
        const int k0 = rem / g.k[1];
        ktab[k] = make_int4(co * g.L, k0 * g.dil[0], k1 * g.dil[1], k2 * g.dil[2]);
    }
}
// Wt[grp][ci][co][kidx] = W[grp*Mg + co][ci][kidx]
__global__ void conv_wt_kernel(float* __restrict__ wt, const float* __restrict__ w, ConvGeom g) {
    const long long total = (long long)g.Cout * g.Cg * g.KK;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int kidx = (int)(i % g.KK);
        long long rem = i / g.KK;
        const int co = (int)(rem % g.Mg); rem /= g.Mg;
        const int ci = (int)(rem % g.Cg);
        const int grp = (int)(rem / g.Cg);
        wt[i] = w[((long long)(grp * g.Mg + co) * g.Cg + ci) * g.KK + kidx];
    }
}

// ---- column helpers -----------------------------------------------------------------------------
// flat output position l -> offset of its window origin inside one input plane
__device__ __forceinline__ int window_origin(const ConvGeom& g, int l) {
    const int o2 = l % g.out[2];
    int rem = l / g.out[2];
    const int o1 = rem % g.out[1];
    const int o0 = rem / g.out[1];
    return (o0 * g.stride[0] * g.in[1] + o1 * g.stride[1]) * g.in[2] + o2 * g.stride[2];
}

