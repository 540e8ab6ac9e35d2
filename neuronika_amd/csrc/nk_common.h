// Internal definitions shared by the HIP translation units of libneuronika_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "neuronika_hip.h"

struct nk_prof_rec {
    hipEvent_t start, stop;
    int klass;
    double flop;
};

// conv kernel gradient, mixed launch: what a 64-wide block's k-tile costs in percent of a 128-wide one's - 65 with 128-row tiles
// (C3 sweep, same box: uniform launch 528.1 us; prices 50 / 60 / 65 / 70 / 75 / 80 / 90 -> 547.1 / 501.9 / 495.9 / 503.3 / 509.2 /
// 511.5 / 521.3 us), 80 with 64-row tiles (3 x 3, 64 -> 64 channels at 56 x 56: uniform 311.0 us; 70 / 75 / 80 / 85 / 90 -> 310.4 /
// 301.9 / 299.9 / 301.2 / 303.5 us).  -1 in the device handle = these rules.
constexpr int NK_CONV_NARROW_128 = 65, NK_CONV_NARROW_64 = 80;
struct nk_device {
    int idx = 0;
    hipStream_t compute = nullptr;  // tape-ordered kernels
    hipStream_t comm = nullptr;     // RCCL all-reduce (side stream)
    hipStream_t copy = nullptr;     // H2D input pipeline (pinned staging, overlaps compute)
    hipEvent_t fork = nullptr;      // compute -> comm ordering helper
    hipEvent_t join = nullptr;      // comm -> compute ordering helper
    void* workspace = nullptr;      // stream-ordered scratch (split-K slabs, reduction partials)
    size_t workspace_bytes = 0;
    void* operand_scratch = nullptr;  // second scratch region: an operand built for a kernel that uses the workspace itself
    size_t operand_scratch_bytes = 0;
    int graphs_alive = 0;           // nk_graph objects of this device: their kernels have workspace pointers baked in
    std::vector<void*> workspace_retired;  // outgrown workspaces kept while any graph may still replay into them
    int num_cus = 256;
    int busy_slots = 0;             // resident-block slots held by work on another stream (nk_device_set_busy_slots)
    // development overrides, set through nk_dev_tune (schedule sweeps, schedule-against-schedule parity tests); the library
    // reads no environment variable
    int tune_gemm[6] = {0, 0, 0, 0, 0, 0};  // ti, tj, splits[, tiles per block[, tile-order group height[, look-ahead threshold]]]
    int tune_gemm_n = 0;                     // how many of them are set (< 3: the rules decide)
    int tune_kpair = -1;                     // k-pair blocks: -1 rule, 0 never, 1 lock-step groups, 2 skewed groups
    int tune_chain = -1;                     // chained launches: -1 rule (chains of at most 2048), 0 one chain whatever K, > 0 that length
    int tune_pair = -1;                      // nk_sgemm_pair: -1 rule, 0 always two launches, 1 one launch whenever eligible
    int tune_conv_narrow = -1;                // conv kernel gradient, mixed launch: -1 the rules above, 0 uniform launch, 1..100 the price in percent
    unsigned long long wino_launches = 0;    // convolution launches that took the Winograd kernels (nk_conv_winograd_launches)
    int tune_conv_s2dx = -1;                 // 3x3 stride-2 input gradient, the fused-phase kernel: -1 rule, 0 never, 1 whenever the shape allows
    int tune_conv_wino_dw = -1;              // Winograd kernel gradient: -1 rule, 0 never, 1 whenever the shape allows
    int tune_conv_wino_shape = -1;           // Winograd block shape: -1 rule, 0 narrow (two waves, 64 channels), 1 wide (four waves, 128 channels)
    int tune_conv_wino_stagger = -1;         // staggered start of the Winograd blocks: -1 rule (a quarter of a tile block's MFMA time), 0 off, > 0 shader clocks
    int tune_conv_winograd = -1;             // 3x3 s1 d1 g1 forward / input gradient: -1 rule, 0 never Winograd, 1 whenever the shape allows
    int tune_attn_occ = 0;                   // attention forward: 2 = size the register budget for two blocks per CU
    // bench instrumentation (nk_profile_begin/end)
    bool prof_on = false;
    bool prof_window = false;           // between nk_profile_begin and the first nk_profile_end
    std::vector<nk_prof_rec> prof;      // records of the current window
    std::vector<nk_prof_rec> prof_free;  // recycled event pairs
};

// Event bracket around one kernel launch when profiling is on (no-ops otherwise).
int nk_prof_start(nk_device* dev, int klass, double flop);
int nk_prof_stop(nk_device* dev);

struct nk_event {  // self-contained: stays valid (for destroy) after its device handle is gone
    int idx;
    hipStream_t compute, comm, copy;
    hipEvent_t ev;
};

void nk_set_error(const char* fmt, ...);
int nk_fail_hip(hipError_t e, const char* what, const char* file, int line);
// Stream-ordered scratch of at least `bytes` (grown by realloc when too small; the old block
// is released after a device sync, so kernels already enqueued keep a valid pointer).
int nk_workspace(nk_device* dev, size_t bytes, void** out);
int nk_operand_scratch(nk_device* dev, size_t bytes, void** out);

#define NK_HIP(call)                                                         \
    do {                                                                     \
        hipError_t _e = (call);                                              \
        if (_e != hipSuccess) return nk_fail_hip(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define NK_CHECK(cond, ...)              \
    do {                                 \
        if (!(cond)) {                   \
            nk_set_error(__VA_ARGS__);   \
            return NK_ERR_INVALID;       \
        }                                \
    } while (0)

#define NK_USE(dev)                                        \
    do {                                                   \
        NK_CHECK((dev) != nullptr, "null device handle");  \
        NK_HIP(hipSetDevice((dev)->idx));                  \
    } while (0)

#define NK_LAUNCH_CHECK() NK_HIP(hipGetLastError())

// Host scalars that change from call to call become kernel ARGUMENTS (an optimizer's step-dependent factors, the Philox offset of a
// dropout forward): a hipGraph replay would freeze them at the captured call - the optimizer would silently leave its schedule, every
// replay would drop the same elements - so such a call refuses to be captured.
static inline int nk_refuse_capture(nk_device* dev, const char* what, const char* advice) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    NK_HIP(hipStreamIsCapturing(dev->compute, &cs));
    NK_CHECK(cs == hipStreamCaptureStatusNone, "%s would be frozen at the captured call by a graph replay: %s", what, advice);
    return NK_OK;
}

static inline size_t nk_numel(const int* shape, int nd) {
    size_t n = 1;
    for (int i = 0; i < nd; ++i) n *= (size_t)shape[i];
    return n;
}

// Grid for HBM-bound grid-stride kernels: enough blocks to fill 256 CUs x 8 waves several times over, capped
// (cdna_hip_programming.md Guideline 11; a cap of 8192 measured 4-8 % faster than 2048 on the ReLU / MSE streams).
static inline int nk_stream_grid(size_t work_items, int block) {
    size_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (int)b;
}

// Span walk of a streaming kernel (round 6; benchmarks/native/rmw_stream.hip, profiles/r06_stream_walk.md): block b owns the
// CONTIGUOUS span [b per, (b + 1) per) of the launch's n items (16-byte groups, usually) and walks it blockDim.x items at a time with
// the loads of U trips issued before the first use - instead of the grid-stride walk, in which the whole chip moves through memory
// as one front of grid x block items and a lane has one load per stream in flight.  At 1 GiB per tensor, same box: three reads + one
// write (ReLU / dropout / MSE backward, SGD) 4.4 -> 5.7 TB/s, two reads + one write 4.8 -> 6.0, a copy 5.1 -> 5.7; at 256 MB per
// tensor (Infinity-Cache assisted) equal or better.  `load(i)` returns what item i needs (a struct of float4s), `store(i, r)` finishes it.
template <int U, class Load, class Store>
__device__ __forceinline__ void nk_span_walk(size_t n, Load load, Store store) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, end = lo + per < n ? lo + per : n;
    const size_t step = blockDim.x;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * step < end; i += U * step) {
        decltype(load(i)) r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = load(i + u * step);
#pragma unroll
        for (int u = 0; u < U; ++u) store(i + u * step, r[u]);
    }
    for (; i < end; i += step) store(i, load(i));
}

// Streaming 16-byte store of the HBM-bound kernels (`global_store_dwordx4 ... nt`): their outputs are far larger than
// L2 and are not read back by the kernel that writes them; measured +23 % (softmax fwd 5.46 -> 6.73 TB/s) and +35 %
// (dropout fwd 5.1 -> 6.9 TB/s) against plain stores.
typedef float nk_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nk_store_stream(float4* p, const float4& v) {
    nk_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<nk_v4f*>(p));
}


// Streaming 16-byte load (`global_load_dwordx4 ... nt`) for operands a multi-stream pass reads exactly once.  It pays only
// when the pass's working set is beyond the 256 MB Infinity Cache: three-reads-one-write row kernels on 0.8 - 1.1 GB gain
// 10 % (softmax backward 5.6 -> 6.3 TB/s, attention-probability backward 5.75 -> 6.35 TB/s, dropout backward 5.1 -> 5.4),
// the same loops on 200 - 270 MB (ReLU / MSE backward at 4096^2, served largely from the cache) LOSE 10 % with it, and a
// single-input forward kernel loses 8 %.  Hence a launch-time switch: `nk_streams_past_cache(bytes)`.
__device__ __forceinline__ float4 nk_load_stream(const float4* p, bool nt) {
    if (nt) {
        const nk_v4f t = __builtin_nontemporal_load(reinterpret_cast<const nk_v4f*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
    return *p;
}
static inline bool nk_streams_past_cache(size_t bytes_touched) { return bytes_touched > (size_t(384) << 20); }

constexpr int NK_WAVE = 64;

__device__ __forceinline__ float nk_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float nk_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// Block-wide sum for blocks of NT threads (NT multiple of 64, <= 1024); result in all threads.
template <int NT>
__device__ __forceinline__ float nk_block_sum(float v, float* smem /* NT/64 floats */) {
    v = nk_wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += smem[i];
    return r;
}
template <int NT>
__device__ __forceinline__ float nk_block_max(float v, float* smem) {
    v = nk_wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = smem[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) r = fmaxf(r, smem[i]);
    return r;
}

// ---- Philox4x32-10 (Salmon et al., SC'11): the device RNG of Dropout ------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
        const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// ---- Bernoulli(keep) draws of Dropout (oracle: dropout_noise / bernoulli_threshold) ----------------------------------
// The reference draws `Bernoulli::new(1. - p)` (node/dropout/mod.rs:46) - rand 0.8's integer construction
// `p_int = (p * 2^64) as u64; rng.gen::<u64>() < p_int` - from the non-reproducible thread_rng; here the same construction
// on 32-bit values out of Philox4x32-10.  ONE Philox call serves EIGHT consecutive elements: element 8 j + 2 k + r takes word
// k of call j, as it is (r = 0) or rotated by 16 bits (r = 1), and is kept iff that value < floor(keep * 2^32).  Every draw
// resolves the keep probability to 2^-32 (its own 16 bits decide, the partner's only break ties); the two elements sharing
// a word are independent except on those 2^-16 ties.  Half the Philox rounds per element of a word per element.
static inline unsigned nk_keep_threshold(double keep) {
    const double t = __builtin_floor(keep * 4294967296.0);
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (t <= 0.0 ? 0u : (unsigned)t);
}
__device__ __forceinline__ unsigned nk_rot16(unsigned w) { return __builtin_amdgcn_alignbit(w, w, 16); }
// Philox counter of the call that holds element `i` (flat row-major index) and the 0/1 draws of the aligned group of
// four elements starting at i (i % 4 == 0): words (0,1) of the call for the lower half of its eight, (2,3) for the upper.
__device__ __forceinline__ uint4 nk_draw_call(unsigned long long elem, unsigned long long offset, uint2 key) {
    const unsigned long long ctr = elem / 8 + offset;
    return philox4x32_10(make_uint4((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u), key);
}
__device__ __forceinline__ float4 nk_keep4(unsigned wa, unsigned wb, unsigned keep_lt) {
    return make_float4(wa < keep_lt ? 1.f : 0.f, nk_rot16(wa) < keep_lt ? 1.f : 0.f, wb < keep_lt ? 1.f : 0.f,
                       nk_rot16(wb) < keep_lt ? 1.f : 0.f);
}
// the group of four at `elem` (elem % 4 == 0) out of its call
__device__ __forceinline__ float4 nk_keep4_at(unsigned long long elem, unsigned long long offset, uint2 key, unsigned keep_lt) {
    const uint4 r = nk_draw_call(elem, offset, key);
    const bool hi = (elem >> 2) & 1;
    return nk_keep4(hi ? r.z : r.x, hi ? r.w : r.y, keep_lt);
}

