// Loss criteria reducing to a scalar (HBM-bound: every operand element is read once forward, and
// x, t, dx once each + one dx write backward; bound 8 TB/s).
//   AbsoluteError        node/absolute_error/mod.rs:42-58 (fwd), :93-123 (bwd)
//   BinaryCrossEntropy   node/bce/mod.rs:42-62, :97-127
//   BCEWithLogits        node/bce_with_logits/mod.rs:42-66, :101-131
//   KLDiv                node/kldiv/mod.rs:42-59, :92-113
//   NegativeLogLikelihood node/nll/mod.rs:43-69, :104-137
// The reference folds sequentially on one thread; here: per-block partial sums (fixed grid, fixed
// order -> run-to-run deterministic) and a single-block final sum.
#include "nk_common.h"

namespace {

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int RB = 256;
constexpr int MAX_PART = 1024;
constexpr float F32_EPSILON = 1.1920929e-07f;

// ---- per-element forward terms ---------------------------------------------------------------
template <int LOSS>
__device__ __forceinline__ float loss_term(float x, float t) {
    if (LOSS == NK_LOSS_MAE) return fabsf(x - t);
    if (LOSS == NK_LOSS_BCE) {
        // `- t * ln(x).clamp(-100, MAX) + (t - 1) * ln(1 - x).clamp(-100, MAX)`  bce/mod.rs:52-55
        float lx = logf(x), l1x = logf(1.f - x);
        lx = lx < -100.f ? -100.f : lx;      // keeps NaN like f32::clamp
        l1x = l1x < -100.f ? -100.f : l1x;
        return -t * lx + (t - 1.f) * l1x;
    }
    if (LOSS == NK_LOSS_BCE_WITH_LOGITS) {
        // `(1 - t) * x + max + ln(exp(-max) + exp(-x - max))`, max = max(-x, 0)  bce_with_logits/mod.rs:52-57
        const float m = fmaxf(-x, 0.f);
        return (1.f - t) * x + m + logf(expf(-m) + expf(-x - m));
    }
    // KLDiv: `t * (ln t - x)` where t > 0, else 0.  kldiv/mod.rs:50-52 multiplies by the (t > 0) flag
    // AFTER forming 0 * (-inf - x) = NaN, so a zero target poisons the reference's sum; its own vectors
    // (kldiv/test.rs:10,16 has a 0.0 target, expects 0.1530) need the masked form built here.
    return t > 0.f ? t * (logf(t) - x) : 0.f;
}

// ---- per-element backward increments (g = upstream scalar, den = divisor or 1) -----------------
template <int LOSS, bool MEAN>
__device__ __forceinline__ float loss_dterm(float x, float t, float g, float den) {
    if (LOSS == NK_LOSS_MAE) {
        // `((diff != 0) as f32) * (diff.signum() * g / n)`  absolute_error/mod.rs:107-119
        const float diff = x - t;
        const float sg = diff != diff ? diff : copysignf(1.f, diff);  // f32::signum: NaN stays NaN
        const float v = MEAN ? sg * g / den : sg * g;
        return (diff != 0.f ? 1.f : 0.f) * v;
    }
    if (LOSS == NK_LOSS_BCE) {
        // `(x - t) / ((1 - x) * x).max(EPSILON) * g / n`  bce/mod.rs:111-123
        const float v = (x - t) / fmaxf((1.f - x) * x, F32_EPSILON) * g;
        return MEAN ? v / den : v;
    }
    if (LOSS == NK_LOSS_BCE_WITH_LOGITS) {
        // `(sigmoid(x) - t) * g / n`  bce_with_logits/mod.rs:115-127
        const float s = 1.f / (1.f + expf(-x));
        return MEAN ? (s - t) * g / den : (s - t) * g;
    }
    return MEAN ? -t * g / den : -t * g;  // kldiv/mod.rs:103-109
}

template <int LOSS>
__global__ void loss_partial_kernel(const float* __restrict__ x, const float* __restrict__ t, size_t n,
                                    float* __restrict__ part) {
    __shared__ float red[RB / 64];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i], w = reinterpret_cast<const float4*>(t)[i];
        a0 += loss_term<LOSS>(v.x, w.x); a1 += loss_term<LOSS>(v.y, w.y);
        a2 += loss_term<LOSS>(v.z, w.z); a3 += loss_term<LOSS>(v.w, w.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        a0 += loss_term<LOSS>(x[i], t[i]);
    }
    const float s = nk_block_sum<RB>((a0 + a1) + (a2 + a3), red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out = sign * sum(part) [/ den]
__global__ void loss_final_kernel(const float* __restrict__ part, int nparts, float sign, float den, float* __restrict__ out) {
    __shared__ float red[RB / 64];
    float a = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) a += part[i];
    const float s = nk_block_sum<RB>(a, red);
    if (threadIdx.x == 0) out[0] = den > 0.f ? sign * s / den : sign * s;
}

template <int LOSS, bool MEAN>
__global__ void loss_bwd_kernel(float* __restrict__ dx, const float* __restrict__ gs, const float* __restrict__ x,
                                const float* __restrict__ t, size_t n, float den) {
    const float g = gs[0];
    const size_t n4 = n / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 d = reinterpret_cast<float4*>(dx)[i];
        const float4 w = reinterpret_cast<const float4*>(t)[i];
        float4 v = w;
        if (LOSS != NK_LOSS_KLDIV) v = reinterpret_cast<const float4*>(x)[i];
        d.x += loss_dterm<LOSS, MEAN>(v.x, w.x, g, den); d.y += loss_dterm<LOSS, MEAN>(v.y, w.y, g, den);
        d.z += loss_dterm<LOSS, MEAN>(v.z, w.z, g, den); d.w += loss_dterm<LOSS, MEAN>(v.w, w.w, g, den);
        nk_store_stream(reinterpret_cast<float4*>(dx) + i, d);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        dx[i] += loss_dterm<LOSS, MEAN>(LOSS != NK_LOSS_KLDIV ? x[i] : t[i], t[i], g, den);
    }
}

// ---- negative log likelihood --------------------------------------------------------------------
// x: (N, C, inner) log-probabilities, target: (N, inner) class indices stored as f32.
// `target as usize` (nll/mod.rs:57): Rust's saturating cast - NaN and negatives become 0, the
// fraction is dropped; an index >= C selects nothing.
__device__ __forceinline__ long long rust_f32_as_usize(float t) {
    if (!(t > 0.f)) return 0;                       // NaN, -x, 0
    if (t >= 9.2233720368547758e18f) return 0x7fffffffffffffffLL;
    return (long long)t;                            // trunc toward zero
}

__global__ void nll_partial_kernel(const float* __restrict__ x, const float* __restrict__ target, long long positions,
                                   int C, long long inner, float* __restrict__ part) {
    __shared__ float red[RB / 64];
    float a = 0.f;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < positions; p += (long long)gridDim.x * blockDim.x) {
        const long long cls = rust_f32_as_usize(target[p]);
        if (cls < C) {
            const long long n = p / inner, r = p % inner;
            a += x[(n * C + cls) * inner + r];
        }
    }
    const float s = nk_block_sum<RB>(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

template <bool MEAN>
__global__ void nll_bwd_kernel(float* __restrict__ dx, const float* __restrict__ gs, const float* __restrict__ target,
                               long long positions, int C, long long inner, float den) {
    const float g = gs[0];
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < positions; p += (long long)gridDim.x * blockDim.x) {
        const long long cls = rust_f32_as_usize(target[p]);
        if (cls < C) {
            const long long n = p / inner, r = p % inner;
            dx[(n * C + cls) * inner + r] -= MEAN ? g * 1.f / den : g * 1.f;  // `*grad_el -= gradient * 1. / n` :120,130
        }
    }
}

int final_sum(nk_device* dev, const float* part, int parts, float sign, float den, float* out) {
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(RB), 0, dev->compute, part, parts, sign, den, out);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

template <int LOSS>
int loss_fwd(nk_device* dev, const float* x, const float* t, size_t n, float den, float* out) {
    void* ws = nullptr;
    int rc = nk_workspace(dev, MAX_PART * sizeof(float), &ws);
    if (rc) return rc;
    int parts = nk_stream_grid(n / 4 + 1, RB);
    if (parts > MAX_PART) parts = MAX_PART;
    hipLaunchKernelGGL((loss_partial_kernel<LOSS>), dim3(parts), dim3(RB), 0, dev->compute, x, t, n, (float*)ws);
    NK_LAUNCH_CHECK();
    return final_sum(dev, (const float*)ws, parts, 1.f, den, out);
}

template <int LOSS>
int loss_bwd(nk_device* dev, float* dx, const float* g, const float* x, const float* t, size_t n, float den, bool mean) {
    const dim3 grid(nk_stream_grid(n / 4 + 1, 256)), block(256);
    if (mean) hipLaunchKernelGGL((loss_bwd_kernel<LOSS, true>), grid, block, 0, dev->compute, dx, g, x, t, n, den);
    else hipLaunchKernelGGL((loss_bwd_kernel<LOSS, false>), grid, block, 0, dev->compute, dx, g, x, t, n, 1.f);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

int loss_geometry(int loss, const int* shape, int nd, int reduction, size_t* n, float* den) {
    NK_CHECK(loss >= NK_LOSS_MAE && loss <= NK_LOSS_KLDIV, "unknown loss %d", loss);
    NK_CHECK(reduction == NK_REDUCTION_SUM || reduction == NK_REDUCTION_MEAN, "unknown reduction %d", reduction);
    NK_CHECK(nd >= 0 && nd <= NK_MAX_DIMS && (nd == 0 || shape), "bad shape");
    size_t total = 1;
    for (int i = 0; i < nd; ++i) { NK_CHECK(shape[i] >= 0, "negative extent"); total *= (size_t)shape[i]; }
    *n = total;
    // Mean divides by the element count, KLDiv by `len_of(Axis(0))` (kldiv/mod.rs:55, :103)
    *den = loss == NK_LOSS_KLDIV ? (float)(nd ? shape[0] : 1) : (float)total;
    return NK_OK;
}

}  // namespace

extern "C" {

int nk_loss_fwd(nk_device* dev, int loss, const float* x, const float* target, const int* shape, int nd, int reduction,
                float* out) {
    NK_USE(dev);
    size_t n; float den;
    int rc = loss_geometry(loss, shape, nd, reduction, &n, &den);
    if (rc) return rc;
    NK_CHECK(out != nullptr, "null output scalar");
    NK_CHECK(n == 0 || (x && target), "null input/target");
    NK_CHECK(al16(x) && al16(target), "loss inputs must be 16-byte aligned");
    if (reduction != NK_REDUCTION_MEAN) den = 0.f;
    switch (loss) {
        case NK_LOSS_MAE: return loss_fwd<NK_LOSS_MAE>(dev, x, target, n, den, out);
        case NK_LOSS_BCE: return loss_fwd<NK_LOSS_BCE>(dev, x, target, n, den, out);
        case NK_LOSS_BCE_WITH_LOGITS: return loss_fwd<NK_LOSS_BCE_WITH_LOGITS>(dev, x, target, n, den, out);
        default: return loss_fwd<NK_LOSS_KLDIV>(dev, x, target, n, den, out);
    }
}

int nk_loss_bwd(nk_device* dev, int loss, float* dx, const float* g, const float* x, const float* target, const int* shape,
                int nd, int reduction) {
    NK_USE(dev);
    size_t n; float den;
    int rc = loss_geometry(loss, shape, nd, reduction, &n, &den);
    if (rc) return rc;
    if (n == 0) return NK_OK;
    NK_CHECK(dx && g && target && (x || loss == NK_LOSS_KLDIV), "null pointer");
    NK_CHECK(al16(dx) && al16(target) && al16(x), "loss buffers must be 16-byte aligned");
    const bool mean = reduction == NK_REDUCTION_MEAN;
    switch (loss) {
        case NK_LOSS_MAE: return loss_bwd<NK_LOSS_MAE>(dev, dx, g, x, target, n, den, mean);
        case NK_LOSS_BCE: return loss_bwd<NK_LOSS_BCE>(dev, dx, g, x, target, n, den, mean);
        case NK_LOSS_BCE_WITH_LOGITS: return loss_bwd<NK_LOSS_BCE_WITH_LOGITS>(dev, dx, g, x, target, n, den, mean);
        default: return loss_bwd<NK_LOSS_KLDIV>(dev, dx, g, x, target, n, den, mean);
    }
}

static int nll_geometry(const int* shape, int nd, int reduction, long long* N, int* C, long long* inner) {
    NK_CHECK(reduction == NK_REDUCTION_SUM || reduction == NK_REDUCTION_MEAN, "unknown reduction %d", reduction);
    NK_CHECK(nd >= 2 && nd <= NK_MAX_DIMS && shape, "nll input must be (minibatch, C, d1..dk), got %d dims", nd);
    *N = shape[0]; *C = shape[1]; *inner = 1;
    for (int i = 2; i < nd; ++i) *inner *= shape[i];
    NK_CHECK(*N >= 0 && *C >= 0 && *inner >= 0, "negative extent");
    return NK_OK;
}

int nk_nll_fwd(nk_device* dev, const float* x, const float* target, const int* shape, int nd, int reduction, float* out) {
    NK_USE(dev);
    long long N, inner; int C;
    int rc = nll_geometry(shape, nd, reduction, &N, &C, &inner);
    if (rc) return rc;
    NK_CHECK(out != nullptr, "null output scalar");
    const long long positions = N * inner;
    NK_CHECK(positions == 0 || (x && target), "null input/target");
    void* ws = nullptr;
    rc = nk_workspace(dev, MAX_PART * sizeof(float), &ws);
    if (rc) return rc;
    int parts = nk_stream_grid((size_t)positions + 1, RB);
    if (parts > MAX_PART) parts = MAX_PART;
    hipLaunchKernelGGL(nll_partial_kernel, dim3(parts), dim3(RB), 0, dev->compute, x, target, positions, C, inner, (float*)ws);
    NK_LAUNCH_CHECK();
    // Mean: `-total / len_of(Axis(0))` (nll/mod.rs:64)
    return final_sum(dev, (const float*)ws, parts, -1.f, reduction == NK_REDUCTION_MEAN ? (float)N : 0.f, out);
}

int nk_nll_bwd(nk_device* dev, float* dx, const float* g, const float* target, const int* shape, int nd, int reduction) {
    NK_USE(dev);
    long long N, inner; int C;
    int rc = nll_geometry(shape, nd, reduction, &N, &C, &inner);
    if (rc) return rc;
    const long long positions = N * inner;
    if (positions == 0 || C == 0) return NK_OK;
    NK_CHECK(dx && g && target, "null pointer");
    const dim3 grid(nk_stream_grid((size_t)positions, 256)), block(256);
    // Mean: `n = target.len()` (nll/mod.rs:114)
    if (reduction == NK_REDUCTION_MEAN)
        hipLaunchKernelGGL(nll_bwd_kernel<true>, grid, block, 0, dev->compute, dx, g, target, positions, C, inner, (float)positions);
    else
        hipLaunchKernelGGL(nll_bwd_kernel<false>, grid, block, 0, dev->compute, dx, g, target, positions, C, inner, 1.f);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

}  // extern "C"
