// f32 GEMM on the gfx950 matrix cores: replaces `ndarray::linalg::general_mat_mul` at the six
// call sites of the reference's MatMul nodes
//   node/matrix_matrix_mul/mod.rs:33 (NN, beta 0), :65 (NT, beta 1), :97 (TN, beta 1)
//   node/matrix_matrix_mul_t/mod.rs:33 (NT, beta 0), :65 (NN, beta 1), :97 (TN, beta 1)
// Bound: f32 MFMA (157.3 TFLOP/s).  Structure: (64*TI)x(64*TJ)x32 block tile (128x128 by default,
// 64-wide variants for narrow or small problems), double-buffered LDS, register-staged global
// loads issued before the MFMAs of the current tile and written to LDS behind them (one barrier
// per k-tile), XCD-aware tile order, split-K with a deterministic second pass when M*N alone
// cannot fill the chip.
#include "nk_mma.h"
#include <utility>

using namespace nkmma;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
constexpr int GEMM_CHAIN_K = 2048;   // longest f32 chain of an unsplit plain-epilogue launch (chained launches, gemm_impl)
constexpr int PF2_MIN_KTILES = 48;  // default threshold of the two-k-tile look-ahead loop (per-layout rules in gemm_impl)
// Summation order (a contract, pinned bit for bit by tests/test_gpu_parity.py::test_sgemm_is_the_device_order_model_bit_for_bit
// against oracle/device_order_sgemm.c): every output is ONE f32 fma chain over the block's k range in the MFMA feeding order;
// split-K adds the splits' chains in split order, a k-pair block adds its two halves.  The reference's sgemm (crate
// matrixmultiply) packs K in blocks of 256 and adds each block's register sum to C, so at K = 4096 the device's error is
// ~10x the blocked CPU sum's (still 4e-10 absolute on the C4 gradients).  Ending the chain every 1024 / 2048 products was
// built three ways in round 4 - a second register set, a cold in-loop block spilling to a workspace slab, whole pipeline runs
// per chain - and measured at 1.5 - 20 % of GEMM time (the 128x128 kernels have no register to spare and every variant
// perturbed the k-loop's allocation); it is not in the product.  Numbers, and the parity policy that follows from them:
// DESIGN.md section 5, profiles/r04_c4_tolerance_model.json, profiles/r04_kfold_sessions.md.

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    long long lda, ldb, ldc;
    float alpha, beta;
    const float* bias;  // optional column bias (length N) added in the epilogue: C = alpha*A.B + bias + beta*C
    // Fused ReLU forms of `nn::Linear` (node/relu/mod.rs:29-38, 67-79 joined to the GEMM that produces their operand):
    int relu;           // forward:  C = max(alpha*A.B + bias, 0)          (`o.max(0.)`: a NaN gives 0)
    const float* mask;  // backward: C = beta*C + m(alpha*A.B), m(v) = v where mask[row][col] > 0, 0*v elsewhere (0 * inf = NaN,
    long long ldm;      //           as `((x > 0.) as usize as f32) * g`); mask is M x N with leading dimension ldm
    // two-level batch
    int batch_inner;
    long long sAo, sAi, sBo, sBi, sCo, sCi;
    // split-K
    int splits;       // >= 1
    int k_per_split;  // multiple of BK
    float* slabs;     // [splits][batch][M][N] partials when splits > 1
    int tiles_m, tiles_n;
    // Short reductions (attention's K = 64: two k-tiles per output tile): a block walks `chunk` consecutive tiles of the
    // tile sequence and loads the first k-tile of the next one before the last MFMA block of the current one: block
    // dispatch and the first-load latency are paid once per chunk (scores GEMM of C5: 975 -> 865 us).  What is left is the
    // output: 2.1 GB of C tiles leave at ~2.5 TB/s while the MFMA pipe is 55 % busy (rocprofv3: TCC_EA0_WRREQ 2.2 GB,
    // L2 hit rate 0.62 because the C lines push Q / K out, effective clock 1.96 GHz).  Tried on top and dropped, measured:
    // a second accumulator set with the previous tile's stores interleaved behind the MFMA groups of the next one, with
    // transposed MFMA blocks so that they are 16-byte stores (bit-identical results, 880 - 970 us: not faster), and a
    // row-major tile order so that one block writes whole 4 KB rows (no gain).
    int chunk;        // >= 1 tiles per block (1: one tile per block, the classic grid)
    int group_m;      // tile order: column-major inside groups of `group_m` tile rows (1: row-major, tn fastest)
    int pf2_min;      // reductions of at least this many k-tiles take the two-k-tile look-ahead loop
    int kskew;        // k-pair blocks: group 1 runs half a k-tile out of phase with group 0
};

// C tile <- accumulators (or the split's slab).  Every load (bias, old C) is issued first and folded into the accumulators
// in registers; the stores come last.  (A one-walk `*q = f(*q)` serialises 16*TI*TJ load -> store round trips per lane,
// and loads still pending when the per-element conditional store blocks are entered make each of them wait for the
// previous store as well.)
// EPX ("extended"): the instantiation also carries the ReLU forms (`relu`, `mask`).  A template parameter and not just two
// run-time flags because the mere presence of that code changes what the compiler makes of the WHOLE kernel: same box, 4096^3,
// plain launches through a kernel with / without it - NT 135.7 / 133.5, NN 135.9 / 135.9, TN 133.2 / 137.5 TFLOP/s (round 4,
// session l).  gemm_impl therefore takes the extended kernel where it is needed or faster (NT) and the plain one elsewhere.
template <bool ALIGNED, int TI, int TJ, bool EPX>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TI][TJ], int m0, int n0, int bo, int bi, int split,
                                              int batch, int wr, int wc, int lane) {
    if (p.splits > 1) {
        float* S = p.slabs + ((long long)split * gridDim.z + batch) * (long long)p.M * p.N;
        const int M = p.M, N = p.N;
        acc_foreach<TI, TJ>(acc, wr, wc, lane, [&](int r, int c, float v) {
            const int row = m0 + r, col = n0 + c;
            if (ALIGNED || (row < M && col < N)) S[(long long)row * N + col] = v;
        });
        return;
    }
    float* C = p.C + bo * p.sCo + bi * p.sCi;
    const float alpha = p.alpha, beta = p.beta;
    const int M = p.M, N = p.N;
    const long long ldc = p.ldc;
    if (EPX && p.mask != nullptr) {
        // ReLU backward joined to the store (nk_linear_bwd_input_relu): C = beta * C + m(alpha * acc).  Its own block, so
        // that the common epilogue below stays the code it was; two round trips (mask, then old C), 16 values at a time.
        const float* Mk = p.mask;
        const long long ldm = p.ldm;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float mv[16], old[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + (wr * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = n0 + (wc * TJ + j) * 32 + (lane & 31);
                    const bool ok = ALIGNED || (row < M && col < N);
                    mv[e] = ok ? Mk[row * ldm + col] : 0.f;
                    old[e] = (ok && beta != 0.f) ? C[row * ldc + col] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + (wr * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = n0 + (wc * TJ + j) * 32 + (lane & 31);
                    float o = alpha * acc[i][j][e];
                    o = mv[e] > 0.f ? o : 0.f * o;
                    if (ALIGNED || (row < M && col < N)) C[row * ldc + col] = beta == 0.f ? o : fmaf(beta, old[e], o);
                }
            }
        return;
    }
    if (p.bias != nullptr || beta != 0.f || alpha != 1.f || (EPX && p.relu)) {
        float old[TI][TJ][16];
        if (beta != 0.f)
            acc_foreach_idx<TI, TJ>(acc, wr, wc, lane, [&](int i, int j, int e, int r, int c, float) {
                const int row = m0 + r, col = n0 + c;
                old[i][j][e] = (ALIGNED || (row < M && col < N)) ? C[row * ldc + col] : 0.f;
            });
        float bv[TJ];  // a lane owns TJ columns
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = n0 + (wc * TJ + j) * 32 + (lane & 31);
            bv[j] = (p.bias != nullptr && (ALIGNED || col < N)) ? p.bias[col] : 0.f;
        }
        const bool relu = EPX && p.relu != 0;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float o = alpha * acc[i][j][e];
                    if (p.bias != nullptr) o += bv[j];  // Linear: fl(acc + bias[col]) == the separate Addition node
                    if constexpr (EPX) o = relu ? fmaxf(o, 0.f) : o;  // ... followed by the ReLU node (`o.max(0.)`: a NaN gives 0)
                    acc[i][j][e] = beta == 0.f ? o : fmaf(beta, old[i][j][e], o);
                }
    }
    acc_foreach<TI, TJ>(acc, wr, wc, lane, [&](int r, int c, float v) {
        const int row = m0 + r, col = n0 + c;
        if (ALIGNED || (row < M && col < N)) C[row * ldc + col] = v;
    });
}

// ---- the two-k-tile look-ahead loop of sgemm_kernel (one tile) -----------------------------------------------------
// Two k-tiles of look-ahead in registers: tile it+1 (P, loaded during the previous trip) goes to LDS at the START of a
// trip, the loads of tile it+2 (Q) are issued in front of it and have a whole trip plus to land.  The end of a trip is
// then MFMAs -> barrier, instead of MFMAs -> wait for this trip's own loads -> 8 LDS writes -> barrier.  Unrolled by
// two so that P / Q and the LDS buffers are static (even tiles in buf0, odd tiles in buf1).
// `skew` (k-pair blocks, group 1): the first half of a k-tile's MFMAs is issued BEFORE the trip's staging stores.
template <bool AKC, bool BKC, bool ALIGNED, int TI, int TJ, int KG>
__device__ __forceinline__ void gemm_loop_lookahead2(TileLoader<AKC, 64 * TI>& la, TileLoader<BKC, 64 * TJ>& lb, f32x16 (&acc)[TI][TJ],
                                                     float* smem, int nt, bool skew, int t, int wr, int wc, int lane) {
    constexpr int BM = 64 * TI, BN = 64 * TJ;
    constexpr int TA_FLOATS = tile_floats<AKC, BM>(), STAGE = TA_FLOATS + tile_floats<BKC, BN>();
    float* const buf0 = smem;
    float* const buf1 = smem + STAGE;
    Stage<BM / 32> pa, qa;
    Stage<BN / 32> pb, qb;
    if (nt > 0) {
        pa = la.template load<ALIGNED>(t);
        pb = lb.template load<ALIGNED>(t);
        stage_store<AKC, BM>(buf0, pa, t);
        stage_store<BKC, BN>(buf0 + TA_FLOATS, pb, t);
    }
    __syncthreads();
    if (nt > 1) {  // tile 1 -> P
        pa = la.template load<ALIGNED>(t);
        pb = lb.template load<ALIGNED>(t);
    }
    int it = 0;  // invariant: tile `it` (even) is in buf0, tile it+1 in P
    for (; it + 3 < nt; it += 2) {
        if (KG == 2 && skew) {
            mma_tile<AKC, BKC, TI, TJ, 0, BK / 16>(buf0, buf0 + TA_FLOATS, acc, wr, wc, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        qa = la.template load<ALIGNED>(t);  // tile it+2
        qb = lb.template load<ALIGNED>(t);
        stage_store<AKC, BM>(buf1, pa, t);
        stage_store<BKC, BN>(buf1 + TA_FLOATS, pb, t);
        __builtin_amdgcn_sched_barrier(0);
        if (KG == 2 && skew) mma_tile<AKC, BKC, TI, TJ, BK / 16, BK / 8>(buf0, buf0 + TA_FLOATS, acc, wr, wc, lane);
        else mma_tile<AKC, BKC, TI, TJ>(buf0, buf0 + TA_FLOATS, acc, wr, wc, lane);
        __syncthreads();
        if (KG == 2 && skew) {
            mma_tile<AKC, BKC, TI, TJ, 0, BK / 16>(buf1, buf1 + TA_FLOATS, acc, wr, wc, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        pa = la.template load<ALIGNED>(t);  // tile it+3
        pb = lb.template load<ALIGNED>(t);
        stage_store<AKC, BM>(buf0, qa, t);
        stage_store<BKC, BN>(buf0 + TA_FLOATS, qb, t);
        __builtin_amdgcn_sched_barrier(0);
        if (KG == 2 && skew) mma_tile<AKC, BKC, TI, TJ, BK / 16, BK / 8>(buf1, buf1 + TA_FLOATS, acc, wr, wc, lane);
        else mma_tile<AKC, BKC, TI, TJ>(buf1, buf1 + TA_FLOATS, acc, wr, wc, lane);
        __syncthreads();
    }
    const int left = nt - it;  // 0 (nt == 0), 1, 2 or 3 tiles: `it` in buf0, it+1 in P
    if (left == 3) {
        qa = la.template load<ALIGNED>(t);
        qb = lb.template load<ALIGNED>(t);
    }
    if (left >= 2) {
        stage_store<AKC, BM>(buf1, pa, t);
        stage_store<BKC, BN>(buf1 + TA_FLOATS, pb, t);
    }
    if (left >= 1) mma_tile<AKC, BKC, TI, TJ>(buf0, buf0 + TA_FLOATS, acc, wr, wc, lane);
    if (left >= 2) {
        __syncthreads();
        if (left == 3) {
            stage_store<AKC, BM>(buf0, qa, t);
            stage_store<BKC, BN>(buf0 + TA_FLOATS, qb, t);
        }
        mma_tile<AKC, BKC, TI, TJ>(buf1, buf1 + TA_FLOATS, acc, wr, wc, lane);
        if (left == 3) {
            __syncthreads();
            mma_tile<AKC, BKC, TI, TJ>(buf0, buf0 + TA_FLOATS, acc, wr, wc, lane);
        }
    }
}

// KG = 2 ("k-pair"): a 512-thread block whose two groups of four waves each run this loop over one HALF of the block's
// reduction, with their own LDS images, and add the two accumulator sets through LDS in a fixed order before the epilogue.
// For grids of at most one 128x128 block per CU (2048^3: 256 tiles): a CU then holds two waves per SIMD - what a 4096^3
// launch gets from two resident blocks - without split-K's slabs and second pass.  The two groups share the block's
// barriers (same trip count); with `kskew` group 1 issues the first half of a k-tile's MFMAs BEFORE its staging stores, so
// that the two waves of a SIMD are not in their staging phase at the same time.
// floats of LDS one block of a given instantiation needs (both stages of every wave group)
template <bool TA, bool TB, int TI, int TJ, int KG>
constexpr int gemm_smem_floats() { return KG * 2 * (tile_floats<!TA, 64 * TI>() + tile_floats<TB, 64 * TJ>()); }

template <bool TA, bool TB, bool ALIGNED, int TI, int TJ, int KG = 1, bool EPX = false>
__global__ __launch_bounds__(NT * KG, (min_waves<TI, TJ, !TA && TB>())) void sgemm_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem_all[gemm_smem_floats<TA, TB, TI, TJ, KG>()];  // <= 73,728 B at 128x128 (k-pair: twice that)
#define NK_GEMM_BX blockIdx.x
#define NK_GEMM_NBX gridDim.x
#define NK_GEMM_SPLIT blockIdx.y
#include "nk_gemm_body.h"
#undef NK_GEMM_BX
#undef NK_GEMM_NBX
#undef NK_GEMM_SPLIT
}

// one problem of sgemm_pair_kernel: the same block program for block `bx` of `nbx`
template <bool TA, bool TB, bool ALIGNED, int TI, int TJ, int KG, bool EPX>
__device__ __forceinline__ void sgemm_body(const GemmArgs& p, float* smem_all, int bx, int nbx) {
#define NK_GEMM_BX bx
#define NK_GEMM_NBX nbx
#define NK_GEMM_SPLIT blockIdx.y
#include "nk_gemm_body.h"
#undef NK_GEMM_BX
#undef NK_GEMM_NBX
#undef NK_GEMM_SPLIT
}
// ... and with the block's range of the reduction passed in (sgemm_tail_kernel)
template <bool TA, bool TB, bool ALIGNED, int TI, int TJ, int KG, bool EPX>
__device__ __forceinline__ void sgemm_body_split(const GemmArgs& p, float* smem_all, int bx, int nbx, int split_no) {
#define NK_GEMM_BX bx
#define NK_GEMM_NBX nbx
#define NK_GEMM_SPLIT split_no
#include "nk_gemm_body.h"
#undef NK_GEMM_BX
#undef NK_GEMM_NBX
#undef NK_GEMM_SPLIT
}

// Two independent GEMMs in ONE launch: blocks [0, nblk0) run problem 0, the rest problem 1 (each with its own layout, same
// tile shape).  For the two products of a MatMul node's backward pass (node/matrix_matrix_mul/mod.rs:63-105: dA += G.B^T and
// dB += A^T.G read the same G) when neither fills the chip by itself: at 1024^3 / 2048^3 a launch is 256 blocks - one per CU,
// a prologue, one wave of MFMAs and an epilogue with nothing else resident to hide them behind.  Two problems side by side
// give every CU a second block (what a 4096^3 launch has) and pay the launch boundary, dispatch ramp and tail once; two large
// grids (the dK / dV products of the attention backward) share their last, partly filled wave of resident blocks.
// Aligned, unsplit, one tile per 256-thread block, plain epilogue (beta); every output is the SAME chain of fmas as in a
// launch of its own without k-pair blocks - bit-identical.
struct GemmPairArgs {
    GemmArgs p0, p1;
    int nblk0;
};
template <bool TA0, bool TB0, bool TA1, bool TB1, int TI, int TJ>
__global__ __launch_bounds__(NT, (min_waves<TI, TJ, (!TA0 && TB0) || (!TA1 && TB1)>())) void sgemm_pair_kernel(GemmPairArgs pp) {
    constexpr int F0 = gemm_smem_floats<TA0, TB0, TI, TJ, 1>(), F1 = gemm_smem_floats<TA1, TB1, TI, TJ, 1>();
    __shared__ __attribute__((aligned(16))) float smem_all[F0 > F1 ? F0 : F1];
    const int nblk0 = pp.nblk0;
    if ((int)blockIdx.x < nblk0)
        sgemm_body<TA0, TB0, true, TI, TJ, 1, false>(pp.p0, smem_all, blockIdx.x, nblk0);
    else
        sgemm_body<TA1, TB1, true, TI, TJ, 1, false>(pp.p1, smem_all, blockIdx.x - nblk0, gridDim.x - nblk0);
}

// A grid that shares the chip (nk_device_set_busy_slots): the data-parallel exchange keeps `busy` of the GPU's resident-block
// slots - RCCL's channel workgroups, one slot each - while the backward GEMMs run.  A 4096^3 launch is 1024 tiles = exactly
// two rounds of the 512 slots of an idle chip; with 16 slots gone the tiles no longer divide, a CU that shares its registers
// with a foreign workgroup runs one block at a time and picks up a whole tile (~0.4 ms) just before everybody else is done:
// 10 - 15 % per GEMM with 8 - 64 foreign workgroups at ANY byte rate (profiles/r04_gemm_under_load.md).  The defence is
// granularity where it matters: the grid runs as many WHOLE rounds of the free slots as fit (blocks [0, nfull), problem p0 over
// the head of the tile sequence) and the tiles left over - the tail of the sequence, a rectangle of the last group of tile
// rows: p1, the same product on sub-matrices - are cut along K into as many pieces as there are free slots (blocks
// [nfull, ..), piece = (block - nfull) / ntail).  Pieces store into slabs; a second pass adds them in piece order and applies
// the launch's epilogue (alpha, beta, bias, relu, mask) to that rectangle.  One launch, blocks dispatched in index order: the
// pieces fill the chip as the last whole tiles drain.  Bits are a function of (shape, busy slots): deterministic, and equal to
// a plain launch for every tile outside the rectangle.  Idle chip (busy = 0): never taken.
struct GemmTailArgs {
    GemmArgs p0, p1;
    int nfull, ntail;
};
template <bool TA, bool TB, bool EPX>
__global__ __launch_bounds__(NT, (min_waves<2, 2, !TA && TB>())) void sgemm_tail_kernel(GemmTailArgs pp) {
    __shared__ __attribute__((aligned(16))) float smem_all[gemm_smem_floats<TA, TB, 2, 2, 1>()];
    const int nfull = pp.nfull;
    if ((int)blockIdx.x < nfull) {
        sgemm_body_split<TA, TB, true, 2, 2, 1, EPX>(pp.p0, smem_all, blockIdx.x, nfull, 0);
    } else {
        const int r = (int)blockIdx.x - nfull, ntail = pp.ntail;
        sgemm_body_split<TA, TB, true, 2, 2, 1, false>(pp.p1, smem_all, r % ntail, ntail, r / ntail);
    }
}

// Second pass of split-K: C = alpha * sum_s slab[s] + beta * C, fixed summation order.
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, int M, int N,
                                     long long ldc, int splits, int nbatch, int batch_inner,
                                     long long sCo, long long sCi, float alpha, float beta,
                                     const float* __restrict__ bias, int relu, const float* __restrict__ mask, long long ldm) {
    const long long per = (long long)M * N;
    const long long total = per * nbatch;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long e = i % per;
        const int row = (int)(e / N), col = (int)(e % N);
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += slabs[((long long)k * nbatch + b) * per + e];
        float* q = C + (b / batch_inner) * sCo + (b % batch_inner) * sCi + row * ldc + col;
        float o = alpha * s;
        if (mask) o = mask[row * ldm + col] > 0.f ? o : 0.f * o;  // (unbatched launches only: gemm_impl)
        if (bias) o += bias[col];
        if (relu) o = fmaxf(o, 0.f);
        *q = beta == 0.f ? o : fmaf(beta, *q, o);
    }
}

// Same, for a C that is one dense block (ldc == N, batches back to back): no index arithmetic, 16-byte accesses.
__global__ void splitk_reduce_flat_kernel(const float* __restrict__ slabs, float* __restrict__ C, long long total4, int splits,
                                          float alpha, float beta, const float* __restrict__ bias, int N, int relu,
                                          const float* __restrict__ mask) {  // mask: dense like C (ldm == N)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < splits; ++k) {
            const float4 v = reinterpret_cast<const float4*>(slabs)[(long long)k * total4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float4 o = make_float4(alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w);
        if (mask) {
            const float4 m = reinterpret_cast<const float4*>(mask)[i];
            o.x = m.x > 0.f ? o.x : 0.f * o.x; o.y = m.y > 0.f ? o.y : 0.f * o.y; o.z = m.z > 0.f ? o.z : 0.f * o.z; o.w = m.w > 0.f ? o.w : 0.f * o.w;
        }
        if (bias) {
            const int col = (int)((i * 4) % N);  // N % 4 == 0: the four lanes stay in one row
            const float4 b = *reinterpret_cast<const float4*>(bias + col);
            o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        float4* q = reinterpret_cast<float4*>(C) + i;
        if (beta != 0.f) { const float4 c = *q; o.x = fmaf(beta, c.x, o.x); o.y = fmaf(beta, c.y, o.y); o.z = fmaf(beta, c.z, o.z); o.w = fmaf(beta, c.w, o.w); }
        *q = o;
    }
}


template <bool TA, bool TB, int TI, int TJ, bool EPX>
static int launch_tile(nk_device* dev, const GemmArgs& p, int nbatch, bool aligned, int kg = 1) {
    dim3 grid((p.tiles_m * p.tiles_n + p.chunk - 1) / p.chunk, p.splits, nbatch), block(NT * kg);
    if constexpr (TI * TJ == 4 || TI * TJ == 1) {
        if (kg == 2) {  // gemm_impl: aligned, one tile per block
            hipLaunchKernelGGL((sgemm_kernel<TA, TB, true, TI, TJ, 2, EPX>), grid, block, 0, dev->compute, p);
            NK_LAUNCH_CHECK();
            return NK_OK;
        }
    }
    if (aligned)
        hipLaunchKernelGGL((sgemm_kernel<TA, TB, true, TI, TJ, 1, EPX>), grid, block, 0, dev->compute, p);
    else
        hipLaunchKernelGGL((sgemm_kernel<TA, TB, false, TI, TJ, 1, EPX>), grid, block, 0, dev->compute, p);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

template <bool TA, bool TB, bool EPX>
static int launch_epx(nk_device* dev, const GemmArgs& p, int nbatch, bool aligned, int ti, int tj, int kg) {
    if (ti == 2 && tj == 2) return launch_tile<TA, TB, 2, 2, EPX>(dev, p, nbatch, aligned, kg);
    if (ti == 2 && tj == 1) return launch_tile<TA, TB, 2, 1, EPX>(dev, p, nbatch, aligned);
    if (ti == 1 && tj == 2) return launch_tile<TA, TB, 1, 2, EPX>(dev, p, nbatch, aligned);
    return launch_tile<TA, TB, 1, 1, EPX>(dev, p, nbatch, aligned, kg);
}
// the extended-epilogue kernels (EPX, see gemm_epilogue) where the launch needs them - and for every NT launch, which they run
// 1.6 % faster at 4096^3 than the plain kernel does; the plain kernels elsewhere (TN: +3.2 %)
template <bool TA, bool TB>
static int launch(nk_device* dev, const GemmArgs& p, int nbatch, bool aligned, int ti, int tj, int kg) {
    const bool epx = p.relu || p.mask != nullptr || (!TA && TB);
    return epx ? launch_epx<TA, TB, true>(dev, p, nbatch, aligned, ti, tj, kg) : launch_epx<TA, TB, false>(dev, p, nbatch, aligned, ti, tj, kg);
}

// What gemm_impl decided for one problem: the kernel arguments and the instantiation (tile shape, k-pair, aligned loads).
struct GemmPlan {
    GemmArgs p;
    int ti, tj, kg, nbatch;
    bool aligned, empty;
};

static int gemm_plan(nk_device* dev, int transA, int transB, int M, int N, int K, float alpha,
                     const float* A, int lda, long long sAo, long long sAi, const float* B, int ldb,
                     long long sBo, long long sBi, float beta, float* C, int ldc, long long sCo,
                     long long sCi, int batch_outer, int batch_inner, const float* bias, int relu,
                     const float* mask, long long ldm, GemmPlan* plan, bool allow_kpair = true) {
    NK_USE(dev);
    NK_CHECK(M >= 0 && N >= 0 && K >= 0 && batch_outer >= 0 && batch_inner >= 0, "negative GEMM extent");
    const int nbatch = batch_outer * batch_inner;
    plan->empty = M == 0 || N == 0 || nbatch == 0;
    plan->nbatch = nbatch;
    if (plan->empty) return NK_OK;
    NK_CHECK(A && B && C, "null GEMM operand");
    NK_CHECK(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
             "leading dimension too small (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    NK_CHECK(nbatch <= 65535, "batch count %d exceeds grid.z", nbatch);

    GemmArgs p{};
    p.A = A; p.B = B; p.C = C;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta;
    p.bias = bias;
    p.relu = relu; p.mask = mask; p.ldm = ldm;
    NK_CHECK(mask == nullptr || (nbatch == 1 && ldm >= N), "masked GEMM: one matrix, ldm >= N");
    p.batch_inner = batch_inner;
    p.sAo = sAo; p.sAi = sAi; p.sBo = sBo; p.sBi = sBi; p.sCo = sCo; p.sCi = sCi;

    // Tile shape and split-K, from a measured sweep of every (tile, split) candidate over mid-size, long-K and narrow
    // shapes (benchmarks/ab_force.py; table in DESIGN.md):
    //  * an extent <= 64, or one that is 64- but not 128-divisible, caps that tile dimension at 64 (no padded half tile);
    //  * if the largest allowed tile already gives >= 512 blocks (one full set of resident slots), use it unsplit;
    //  * else keep the large tile when splitting K can raise the grid to >= 384 blocks with >= 16 k-tiles per split
    //    (128-wide tiles do the most MFMA work per LDS byte, but a block needs a long enough k-chain to amortise its
    //    prologue/epilogue and the second pass);
    //  * else 64x64 tiles, split (>= 8 k-tiles per split) only while the grid has fewer than 256 blocks.
    auto blocks = [&](int ti, int tj) {
        return (long long)((M + 64 * ti - 1) / (64 * ti)) * ((N + 64 * tj - 1) / (64 * tj)) * nbatch;
    };
    const int ktiles = (K + BK - 1) / BK;
    int ti = (M <= 64 || (M % 128 != 0 && M % 64 == 0)) ? 1 : 2;
    int tj = (N <= 64 || (N % 128 != 0 && N % 64 == 0)) ? 1 : 2;
    int splits = 1;
    if (blocks(ti, tj) >= 512) {
        // plenty of blocks: weigh the last, partly filled wave of resident slots (2 blocks/CU for 128x128, 3 for the
        // 64-wide shapes) against the lower MFMA density of smaller tiles; many small blocks also tail off smoothly
        auto wave_eff = [&](long long nb, long long slots) { return (double)nb / (double)(((nb + slots - 1) / slots) * slots); };
        const int ti0 = ti, tj0 = tj;
        double best = (ti0 * tj0 == 4 ? 1.0 : ti0 * tj0 == 2 ? 0.9 : 0.8) * wave_eff(blocks(ti0, tj0), ti0 * tj0 == 4 ? 512 : 768);
        if (ti0 * tj0 == 4) {
            const double c21 = 0.9 * wave_eff(blocks(2, 1), 768), c12 = 0.9 * wave_eff(blocks(1, 2), 768);
            if (c12 > best + 0.05) { best = c12; ti = 1; tj = 2; }
            if (c21 > best + 0.05) { best = c21; ti = 2; tj = 1; }
        }
        if (ti0 * tj0 >= 2) {
            const long long nb1 = blocks(1, 1);
            double q = wave_eff(nb1, 768);
            if (nb1 >= 1536 && q < 0.9) q = 0.9;
            if (0.8 * q > best + 0.05) { ti = 1; tj = 1; }
        }
    } else if (blocks(ti, tj) < 512) {
        const long long nb = blocks(ti, tj);
        long long s = (512 + nb - 1) / nb;
        if (s > ktiles / 16) s = ktiles / 16;
        if (ti * tj == 4 && nb >= 256 && ktiles >= PF2_MIN_KTILES) {
            // one 128x128 block per CU and a reduction long enough for the two-k-tile look-ahead: unsplit beats
            // split-K 2 + second pass (2048^3: NN 97 -> 111, NT 110 -> 113, TN 107.6 -> 110.3 TFLOP/s)
            splits = 1;
        } else if (ti * tj > 1 && s >= 1 && nb * s >= 384) {
            splits = (int)s;
        } else {
            ti = tj = 1;
            const long long nb1 = blocks(1, 1);
            if (nb1 < 256) {
                s = 512 / nb1;
                if (s > ktiles / 8) s = ktiles / 8;
                splits = s < 1 ? 1 : (int)s;
            }
        }
    }
    p.tiles_m = (M + 64 * ti - 1) / (64 * ti);
    p.tiles_n = (N + 64 * tj - 1) / (64 * tj);
    int force_chunk = 0, force_group = 0, force_pf2 = 0;
    // schedule sweeps (benchmarks/ab_force.py) and the parity tests that pit one schedule against another: the device handle's
    // NK_TUNE_GEMM_FORCE values "ti, tj, splits[, chunk[, group_m[, lookahead_min]]]" override the rules (nk_dev_tune)
    if (dev->tune_gemm_n >= 3 && (dev->tune_gemm[0] == 1 || dev->tune_gemm[0] == 2) && (dev->tune_gemm[1] == 1 || dev->tune_gemm[1] == 2)) {
        ti = dev->tune_gemm[0]; tj = dev->tune_gemm[1]; splits = dev->tune_gemm[2] < 1 ? 1 : dev->tune_gemm[2];
        p.tiles_m = (M + 64 * ti - 1) / (64 * ti);
        p.tiles_n = (N + 64 * tj - 1) / (64 * tj);
        if (dev->tune_gemm_n >= 4) force_chunk = dev->tune_gemm[3];
        if (dev->tune_gemm_n >= 5) force_group = dev->tune_gemm[4];
        if (dev->tune_gemm_n >= 6) force_pf2 = dev->tune_gemm[5];
    }
    // when the reduction is split anyway, no split takes more than 128 k-tiles: a chain of at most 4096 products per slab
    // (the f32 chain's rounding error grows with its length; 3072 x 1024 x 32768 - the packed C5
    // weight gradient - would otherwise run as 3 chains of 10944)
    if (splits > 1 && dev->tune_gemm_n < 3 && splits < (ktiles + 127) / 128) splits = (ktiles + 127) / 128;
    int kts = (ktiles + splits - 1) / splits;
    if (kts < 1) kts = 1;
    splits = (ktiles + kts - 1) / kts;
    if (splits < 1) splits = 1;
    p.splits = splits;
    p.k_per_split = kts * BK;
    const int pf2_rule_for_chunk = ti * tj == 1 ? 8 : ((!transA && !transB) ? 32 : PF2_MIN_KTILES);
    // Tiles per block: a short reduction (kts k-tiles of ~3.9 us) cannot amortise the ~7 us a block spends being
    // dispatched, waiting for its first loads and draining its stores, so a block takes enough consecutive tiles of the
    // sequence for ~16 k-tiles of work - as long as the grid keeps at least four waves of resident blocks.
    {
        const long long ntiles = (long long)p.tiles_m * p.tiles_n, slots = ti * tj == 4 ? 512 : (ti * tj == 2 ? 768 : 1024);
        long long c = kts >= 16 ? 1 : (16 + kts - 1) / kts;
        const long long cap = ntiles * splits * nbatch / (4 * slots);
        if (c > cap) c = cap;
        if (c > ntiles) c = ntiles;
        // The chunk loop and the two-k-tile look-ahead loop are ALTERNATIVES inside the kernel (the look-ahead branch
        // handles exactly one tile): whenever this launch will take the look-ahead (kts >= its threshold, forced or by
        // rule), the chunk is 1 - also under a forced configuration (nk_dev_tune), whose lookahead_min may lie below the chunk rule's 8.
        const int pf2_effective = force_pf2 > 0 ? force_pf2 : pf2_rule_for_chunk;
        if (c < 1 || kts >= 8) c = 1;
        if (force_chunk > 0) c = force_chunk;
        if (kts >= pf2_effective) c = 1;
        p.chunk = (int)c;
    }
    p.group_m = force_group > 0 ? force_group : 8;
    // Two k-tiles of look-ahead (the loop that ends a trip MFMAs -> barrier): from a same-box sweep of every layout
    // (benchmarks/ab_force.py with the sixth value of NK_TUNE_GEMM_FORCE; U[0,1) operands):
    //   64x64 tiles (a k-tile is only 16 MFMAs per wave, less than an L2 round trip): from 8 k-tiles on, every layout
    //     (1024^3: NN 80.8 -> 84, NT 81.7 -> 88, TN 78.9 -> 83 TFLOP/s);
    //   NN (both operands row-major: the B tile is read k-major, its loads land last): from 32 k-tiles on
    //     (4096^3 116 -> 137 without / with; 4096 x 4096 x 1024 112 -> 122; 32768 x 1024 x 1024 109 -> 118; but the
    //     convolution-like 128 x 401408 x 576, 18 k-tiles and a single tile row, LOSES 7 %: 110 -> 103);
    //   NT / TN / TT: from 48 k-tiles on (TN 4096^3 133.6 -> 137.1; NT loses 3 % below that: 119 -> 116 at K = 1024).
    // C4 step, same box: 8.75 ms with the look-ahead off, 8.50 with these rules.  (What a k-tile of 64x64 needs is time
    // for its loads, not more waves: a 512-thread variant with two wave groups on alternating k-tiles of one tile -
    // two waves per SIMD where 1024^3 has one - measured 78.7 vs 80.1 TFLOP/s and was dropped.)
    const int pf2_rule = ti * tj == 1 ? 8 : ((!transA && !transB) ? 32 : PF2_MIN_KTILES);
    p.pf2_min = force_pf2 > 0 ? force_pf2 : pf2_rule;

    const int BM = 64 * ti, BN = 64 * tj;
    const bool aligned = (M % BM == 0) && (N % BN == 0) && (K % BK == 0) && (lda % 4 == 0) &&
                         (ldb % 4 == 0) && aligned16(A) && aligned16(B) && (sAo % 4 == 0) &&
                         (sAi % 4 == 0) && (sBo % 4 == 0) && (sBi % 4 == 0);
    // k-pair blocks (sgemm_kernel, KG = 2): 128x128 tiles, aligned, one tile per block, a whole number of k-tile pairs per
    // block, and a grid of at most ONE BLOCK PER CU - with more blocks than CUs two 256-thread blocks share a CU anyway.
    // (its two groups' LDS images, 131 - 147 KB, assume the 160 KB of a gfx950 CU: static_assert in sgemm_kernel.)
    // NK_TUNE_GEMM_KPAIR = 0 (never) / 1 (lock-step groups) / 2 (group 1 half a k-tile out of phase) overrides for sweeps.
    const int kpair_tune = allow_kpair ? dev->tune_kpair : 0;
    int kg = 1;
    p.kskew = 1;
    {
        const long long nblk = (long long)p.tiles_m * p.tiles_n * p.splits * nbatch;
        const bool can = (ti * tj == 4 || ti * tj == 1) && aligned && p.chunk == 1 && kts % 2 == 0 && kts >= 8 && K % (2 * BK) == 0 &&
                         (p.splits == 1 || p.k_per_split * p.splits == K);
        // 64x64 tiles (1024^3: 256 blocks, one wave per SIMD otherwise): from 16 k-tiles per block
        const bool want = kpair_tune >= 0 ? kpair_tune > 0 : nblk <= dev->num_cus && kts >= (ti * tj == 4 ? 32 : 16);
        if (can && want) kg = 2;
        if (kpair_tune == 1) p.kskew = 0;
    }
    plan->p = p; plan->ti = ti; plan->tj = tj; plan->kg = kg; plan->aligned = aligned;
    return NK_OK;
}

// The shared-chip schedule (sgemm_tail_kernel): whole rounds of the free slots as one tile per block, the left-over tiles cut
// along K.  `p`: the plan of an aligned, unsplit, unbatched 128x128-tile launch.  *taken = false: the grid divides (or the
// tail cannot be cut usefully) - the caller launches as usual.
template <bool TA, bool TB>
static void launch_tail(nk_device* dev, const GemmTailArgs& pp, dim3 grid, bool epx) {
    if (epx) hipLaunchKernelGGL((sgemm_tail_kernel<TA, TB, true>), grid, dim3(NT), 0, dev->compute, pp);
    else hipLaunchKernelGGL((sgemm_tail_kernel<TA, TB, false>), grid, dim3(NT), 0, dev->compute, pp);
}
static int gemm_tail_launch(nk_device* dev, int transA, int transB, const GemmArgs& p, bool* taken) {
    *taken = false;
    const long long T = (long long)p.tiles_m * p.tiles_n;
    const long long slots = 2LL * dev->num_cus - dev->busy_slots;  // 128x128 blocks: two per CU
    if (slots < dev->num_cus || T <= slots || T % slots == 0) return NK_OK;
    const int gsize = p.tiles_m % p.group_m == 0 ? p.group_m : p.tiles_m % p.group_m;  // tile rows of the last group
    long long tail = T % slots;
    tail = (tail + gsize - 1) / gsize * gsize;  // whole columns of that group: the tail of the tile sequence is a rectangle
    if (tail > (long long)gsize * p.tiles_n || tail >= T) return NK_OK;
    const int ktiles = p.K / BK;
    long long pieces = slots / tail;
    // Fewer than four pieces per left-over tile (more than a quarter of a round left over: 64 busy slots at 4096^3) and the cut
    // costs more than the ragged round it replaces: told 9.8 - 10.4 % against 7.0 - 12.2 % plain with 64 foreign workgroups,
    // while 8 / 16 / 32 give 3.3 - 5.0 / 3.2 - 5.2 / 4.9 - 6.9 % against 7.1 - 13.5 % (profiles/r05_gemm_under_load.md)
    if (pieces < 4) return NK_OK;
    int kts = (int)((ktiles + pieces - 1) / pieces);
    if (kts < 4) kts = 4;  // a piece must outlast its own prologue and slab store
    const int s_eff = (ktiles + kts - 1) / kts;
    if (s_eff < 2) return NK_OK;
    const int cols = (int)(tail / gsize);
    const int m_lo = (p.tiles_m - gsize) * 128, n_lo = (p.tiles_n - cols) * 128;
    const int Ms = p.M - m_lo, Ns = p.N - n_lo;
    void* ws = nullptr;
    int rc = nk_workspace(dev, (size_t)s_eff * Ms * Ns * sizeof(float), &ws);
    if (rc) return rc;
    GemmTailArgs pp;
    pp.p0 = p;
    pp.p1 = p;
    GemmArgs& q = pp.p1;
    q.A = p.A + (transA ? (long long)m_lo : (long long)m_lo * p.lda);
    q.B = p.B + (transB ? (long long)n_lo * p.ldb : (long long)n_lo);
    q.C = p.C + (long long)m_lo * p.ldc + n_lo;
    q.M = Ms; q.N = Ns;
    q.tiles_m = gsize; q.tiles_n = cols;
    q.splits = s_eff; q.k_per_split = kts * BK; q.slabs = (float*)ws;
    q.bias = nullptr; q.relu = 0; q.mask = nullptr;
    pp.nfull = (int)(T - tail);
    pp.ntail = (int)tail;
    rc = nk_prof_start(dev, NK_KERNEL_SGEMM, 2.0 * p.M * p.N * (double)p.K);
    if (rc) return rc;
    const dim3 grid((unsigned)(pp.nfull + tail * s_eff), 1, 1);
    const bool epx = p.relu || p.mask != nullptr || (!transA && transB);
    if (!transA && !transB) launch_tail<false, false>(dev, pp, grid, epx);
    else if (!transA && transB) launch_tail<false, true>(dev, pp, grid, epx);
    else launch_tail<true, false>(dev, pp, grid, epx);
    NK_LAUNCH_CHECK();
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nk_stream_grid((size_t)Ms * Ns, 256)), dim3(256), 0, dev->compute, q.slabs, q.C, Ms, Ns,
                       p.ldc, s_eff, 1, 1, 0LL, 0LL, p.alpha, p.beta, p.bias ? p.bias + n_lo : nullptr, p.relu,
                       p.mask ? p.mask + (long long)m_lo * p.ldm + n_lo : nullptr, p.ldm);
    NK_LAUNCH_CHECK();
    *taken = true;
    return nk_prof_stop(dev);
}

static int gemm_impl(nk_device* dev, int transA, int transB, int M, int N, int K, float alpha,
                     const float* A, int lda, long long sAo, long long sAi, const float* B, int ldb,
                     long long sBo, long long sBi, float beta, float* C, int ldc, long long sCo,
                     long long sCi, int batch_outer, int batch_inner, const float* bias = nullptr, int relu = 0,
                     const float* mask = nullptr, long long ldm = 0, bool allow_kpair = true, bool allow_chain = true) {
    GemmPlan plan;
    int rc = gemm_plan(dev, transA, transB, M, N, K, alpha, A, lda, sAo, sAi, B, ldb, sBo, sBi, beta, C, ldc, sCo, sCi, batch_outer,
                       batch_inner, bias, relu, mask, ldm, &plan, allow_kpair);
    if (rc || plan.empty) return rc;
    // Chained launches: an unsplit reduction whose f32 chain would be longer than GEMM_CHAIN_K products runs as consecutive launches
    // over equal pieces of K, piece i > 0 with beta = 1 on top of what the pieces before it left in C - every output is then
    // fl(fl(beta C + chain_0) + chain_1) ..., chains of at most GEMM_CHAIN_K (a k-pair block's two halves count as two chains).
    // Why: one chain of 4096 / 8192 one-signed products drifts as K sqrt(K) and leaves SURVEY.md 8c(ii)'s bound 1e-6 K |a| |b|
    // (C4 weight gradients 1.07 x, 8192^3 1.19 x of it in round 5); the reference's GEMM (matrixmultiply, matrix_matrix_mul/mod.rs:
    // 33-39) sums K in cache blocks as well.  Every in-kernel form of the cut cost the k-loop's schedule 1.5 - 20 % (DESIGN.md
    // section 8); a second launch costs its prologue / epilogue and one more pass over C.  Plain-epilogue launches only (MatMul,
    // MatMulT, every weight gradient): the epilogue functions (bias, ReLU, mask) act on the WHOLE sum.  Bits stay a function of
    // the shape.  NK_TUNE_GEMM_CHAIN overrides the length (0: one chain), a forced configuration (NK_TUNE_GEMM_FORCE) is one chain.
    {
        const long long chain = dev->tune_chain < 0 ? GEMM_CHAIN_K : dev->tune_chain;
        const long long reach = chain * plan.kg;  // the K one launch may cover
        if (allow_chain && chain > 0 && K > reach && plan.p.splits == 1 && !bias && !relu && !mask && dev->tune_gemm_n < 3) {
            const int pieces = (int)((K + reach - 1) / reach);
            const int per = ((K + pieces - 1) / pieces + 2 * BK - 1) / (2 * BK) * (2 * BK);  // whole k-tile pairs (k-pair blocks, 16-byte rows)
            for (int k0 = 0; k0 < K && !rc; k0 += per) {
                const int kk = K - k0 < per ? K - k0 : per;
                rc = gemm_impl(dev, transA, transB, M, N, kk, alpha, A + (transA ? (long long)k0 * lda : (long long)k0), lda, sAo, sAi,
                               B + (transB ? (long long)k0 : (long long)k0 * ldb), ldb, sBo, sBi, k0 == 0 ? beta : 1.f, C, ldc, sCo, sCi,
                               batch_outer, batch_inner, nullptr, 0, nullptr, 0, allow_kpair, false);
            }
            return rc;
        }
    }
    GemmArgs& p = plan.p;
    const int nbatch = plan.nbatch, ti = plan.ti, tj = plan.tj, kg = plan.kg;
    const bool aligned = plan.aligned;
    if (dev->busy_slots > 0 && ti == 2 && tj == 2 && kg == 1 && aligned && p.splits == 1 && p.chunk == 1 && nbatch == 1 &&
        !(transA && transB) && dev->tune_gemm_n < 3) {
        bool taken = false;
        rc = gemm_tail_launch(dev, transA, transB, p, &taken);
        if (rc || taken) return rc;
    }
    if (p.splits > 1) {
        void* ws = nullptr;
        rc = nk_workspace(dev, (size_t)p.splits * nbatch * M * N * sizeof(float), &ws);
        if (rc) return rc;
        p.slabs = (float*)ws;
    }
    rc = nk_prof_start(dev, NK_KERNEL_SGEMM, 2.0 * M * N * (double)K * nbatch);
    if (rc) return rc;
    if (!transA && !transB) rc = launch<false, false>(dev, p, nbatch, aligned, ti, tj, kg);
    else if (!transA && transB) rc = launch<false, true>(dev, p, nbatch, aligned, ti, tj, kg);
    else if (transA && !transB) rc = launch<true, false>(dev, p, nbatch, aligned, ti, tj, kg);
    else rc = launch<true, true>(dev, p, nbatch, aligned, ti, tj, kg);
    if (rc) return rc;
    if (p.splits > 1) {
        const long long total = (long long)M * N * nbatch;
        const bool dense = ldc == N && N % 4 == 0 && aligned16(C) && (!bias || aligned16(bias)) && (!mask || (ldm == N && aligned16(mask))) &&
                           (nbatch == 1 || (sCi == (long long)M * N && (batch_outer == 1 || sCo == (long long)batch_inner * M * N)));
        if (dense)
            hipLaunchKernelGGL(splitk_reduce_flat_kernel, dim3(nk_stream_grid((size_t)(total / 4), 256)), dim3(256), 0, dev->compute,
                               p.slabs, C, total / 4, p.splits, alpha, beta, bias, N, relu, mask);
        else
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nk_stream_grid((size_t)total, 256)), dim3(256), 0,
                           dev->compute, p.slabs, C, M, N, (long long)ldc, p.splits, nbatch, batch_inner,
                           sCo, sCi, alpha, beta, bias, relu, mask, ldm);
        NK_LAUNCH_CHECK();
    }
    return nk_prof_stop(dev);
}

// ---- two problems, one launch (sgemm_pair_kernel) -----------------------------------------------------------------------
struct GemmProblem {
    int transA, transB, M, N, K;
    const float* A; int lda; long long sAo, sAi;
    const float* B; int ldb; long long sBo, sBi;
    float beta; float* C; int ldc; long long sCo, sCi;
};

template <bool TA0, bool TB0>
static int launch_pair(nk_device* dev, const GemmPairArgs& pp, dim3 grid, int ti, int tj) {
    if (ti == 2 && tj == 2) hipLaunchKernelGGL((sgemm_pair_kernel<TA0, TB0, true, false, 2, 2>), grid, dim3(NT), 0, dev->compute, pp);
    else if (ti == 2 && tj == 1) hipLaunchKernelGGL((sgemm_pair_kernel<TA0, TB0, true, false, 2, 1>), grid, dim3(NT), 0, dev->compute, pp);
    else hipLaunchKernelGGL((sgemm_pair_kernel<TA0, TB0, true, false, 1, 1>), grid, dim3(NT), 0, dev->compute, pp);
    NK_LAUNCH_CHECK();
    return NK_OK;
}

// C0 = op(A0).op(B0) + beta0*C0 and C1 = op(A1).op(B1) + beta1*C1, each over the same two-level batch.  One launch when the
// pair is eligible (see sgemm_pair_kernel: aligned, unsplit, equal tile shapes, second product TN) and - by rule - when the two
// grids together fit the resident slots (two launches would each leave CUs without a second block).  Two ordinary launches
// otherwise.
static int gemm_pair_impl(nk_device* dev, const GemmProblem& a, const GemmProblem& b, int batch_outer, int batch_inner) {
    NK_USE(dev);
    // (the plan of a launch of its own WITHOUT k-pair blocks, like the pair's: the bits of a MatMul node's gradients are a function
    //  of the shapes alone - not of which of the two routes the fit-the-chip rule picks, nor of whether the node's other operand
    //  is differentiable: the single-product entry points nk_mm_bwd_left / _right plan the same way)
    auto alone = [&](const GemmProblem& q) {
        return gemm_impl(dev, q.transA, q.transB, q.M, q.N, q.K, 1.f, q.A, q.lda, q.sAo, q.sAi, q.B, q.ldb, q.sBo, q.sBi, q.beta, q.C, q.ldc,
                         q.sCo, q.sCi, batch_outer, batch_inner, nullptr, 0, nullptr, 0, false);
    };
    auto two_launches = [&]() { const int rc = alone(a); return rc ? rc : alone(b); };
    const int mode = dev->tune_pair;  // -1 rule, 0 never, 1 whenever eligible
    if (mode == 0) return two_launches();
    // the plans of 256-thread blocks (a k-pair block is a launch's way to a second wave per SIMD; the pair has the other problem)
    GemmPlan q0, q1;
    int rc = gemm_plan(dev, a.transA, a.transB, a.M, a.N, a.K, 1.f, a.A, a.lda, a.sAo, a.sAi, a.B, a.ldb, a.sBo, a.sBi, a.beta, a.C, a.ldc,
                       a.sCo, a.sCi, batch_outer, batch_inner, nullptr, 0, nullptr, 0, &q0, false);
    if (!rc)
        rc = gemm_plan(dev, b.transA, b.transB, b.M, b.N, b.K, 1.f, b.A, b.lda, b.sAo, b.sAi, b.B, b.ldb, b.sBo, b.sBi, b.beta, b.C, b.ldc,
                       b.sCo, b.sCi, batch_outer, batch_inner, nullptr, 0, nullptr, 0, &q1, false);
    if (rc) return rc;
    if (q0.empty || q1.empty) return two_launches();
    {   // a reduction longer than one chain runs as chained launches (gemm_impl) - on its own
        const long long chain = dev->tune_chain < 0 ? GEMM_CHAIN_K : dev->tune_chain;
        if (chain > 0 && dev->tune_gemm_n < 3 && ((a.K > chain && q0.p.splits == 1) || (b.K > chain && q1.p.splits == 1))) return two_launches();
    }
    const long long batch = (long long)batch_outer * batch_inner;
    {   // One launch runs the problems concurrently: an output that overlaps the other problem's output or operands (x.mm(x):
        // both gradients are one buffer) keeps the order of two launches.  An operand's footprint is `rows` runs of `width`
        // floats, `ld` apart (heads that are column blocks of one matrix widen the run; samples stack rows); two footprints
        // with the same `ld` are also disjoint when their column ranges are (dK and dV are column blocks of the packed
        // projection gradient).
        struct Foot { const float* p; long long rows, width, ld; bool ok; };
        auto foot = [&](const float* p, long long rows, long long cols, long long ld, long long so, long long si) {
            Foot f{p, rows, cols, ld, true};
            if (batch_inner > 1) {
                if (si < ld) f.width = cols + (batch_inner - 1) * si;
                else if (si % ld == 0) f.rows = rows + (batch_inner - 1) * (si / ld);
                else f.ok = false;
            }
            if (batch_outer > 1) {
                if (so % ld == 0) f.rows += (batch_outer - 1) * (so / ld);
                else f.ok = false;
            }
            if (f.width > ld) f.ok = false;
            if (!f.ok) { f.rows = 1; f.ld = 0; f.width = (batch_outer - 1) * so + (batch_inner - 1) * si + (rows - 1) * ld + cols; }
            return f;
        };
        auto opnd = [&](const GemmProblem& q, int which) {
            if (which == 0) return foot(q.A, q.transA ? q.K : q.M, q.transA ? q.M : q.K, q.lda, q.sAo, q.sAi);
            if (which == 1) return foot(q.B, q.transB ? q.N : q.K, q.transB ? q.K : q.N, q.ldb, q.sBo, q.sBi);
            return foot(q.C, q.M, q.N, q.ldc, q.sCo, q.sCi);
        };
        auto hits = [](const Foot& x, const Foot& y) {
            const uintptr_t x0 = (uintptr_t)x.p, x1 = (uintptr_t)(x.p + (x.rows - 1) * x.ld + x.width);
            const uintptr_t y0 = (uintptr_t)y.p, y1 = (uintptr_t)(y.p + (y.rows - 1) * y.ld + y.width);
            if (!(x0 < y1 && y0 < x1)) return false;  // the address ranges do not meet
            if (x.ok && y.ok && x.ld == y.ld && x.ld > 0) {
                const long long ld = x.ld, r = ((y.p - x.p) % ld + ld) % ld;  // y's first column relative to x's
                if (r >= x.width && r + y.width <= ld) return false;          // disjoint column blocks of rows `ld` apart
            }
            return true;
        };
        bool clash = false;
        for (int w = 0; w < 3; ++w) clash = clash || hits(opnd(a, 2), opnd(b, w)) || hits(opnd(b, 2), opnd(a, w));
        if (clash) return two_launches();
    }
    const bool layouts = !(a.transA && a.transB) && b.transA && !b.transB;  // (NN | NT | TN) + TN
    const bool tiles = q0.ti == q1.ti && q0.tj == q1.tj && (q0.ti == q0.tj || (q0.ti == 2 && q0.tj == 1));
    const bool plain = q0.aligned && q1.aligned && q0.p.splits == 1 && q1.p.splits == 1 && q0.p.chunk == 1 && q1.p.chunk == 1;
    if (!(layouts && tiles && plain)) return two_launches();
    const long long nblk0 = (long long)q0.p.tiles_m * q0.p.tiles_n, nblk1 = (long long)q1.p.tiles_m * q1.p.tiles_n;
    if (nblk0 + nblk1 > 0x7fffffffLL) return two_launches();
    if (mode < 0) {
        // resident 256-thread blocks per CU (min_waves): 2 at 128x128, 3 (2 with two padded images) at 128x64, 4 at 64x64.
        // By rule only when BOTH grids fit the resident slots together.  (Folding the partly filled last waves of two LARGE grids
        // into one was measured too - dK / dV of C5, 2 x 4096 blocks over 768 slots, 12 waves -> 11: 1220.8 us in two launches,
        // 1226 - 1237 in one, same box - and mm backward at 4096^3, 2 + 2 full waves either way: +0.5 %.)
        const int per_cu = q0.ti * q0.tj == 4 ? 2 : (q0.ti * q0.tj == 2 ? ((!a.transA && a.transB) ? 2 : 3) : 4);
        if ((nblk0 + nblk1) * batch > (long long)dev->num_cus * per_cu) return two_launches();
    }
    GemmPairArgs pp;
    pp.p0 = q0.p; pp.p1 = q1.p; pp.nblk0 = (int)nblk0;
    rc = nk_prof_start(dev, NK_KERNEL_SGEMM, 2.0 * batch * ((double)a.M * a.N * a.K + (double)b.M * b.N * b.K));
    if (rc) return rc;
    const dim3 grid((unsigned)(nblk0 + nblk1), 1, (unsigned)batch);
    if (a.transA) rc = launch_pair<true, false>(dev, pp, grid, q0.ti, q0.tj);
    else if (a.transB) rc = launch_pair<false, true>(dev, pp, grid, q0.ti, q0.tj);
    else rc = launch_pair<false, false>(dev, pp, grid, q0.ti, q0.tj);
    if (rc) return rc;
    return nk_prof_stop(dev);
}

extern "C" {

int nk_sgemm(nk_device* dev, int transA, int transB, int M, int N, int K, float alpha, const float* A,
             int lda, const float* B, int ldb, float beta, float* C, int ldc) {
    return gemm_impl(dev, transA, transB, M, N, K, alpha, A, lda, 0, 0, B, ldb, 0, 0, beta, C, ldc, 0, 0, 1, 1);
}

int nk_sgemm_batched(nk_device* dev, int transA, int transB, int M, int N, int K, float alpha,
                     const float* A, int lda, long long sAo, long long sAi, const float* B, int ldb,
                     long long sBo, long long sBi, float beta, float* C, int ldc, long long sCo,
                     long long sCi, int batch_outer, int batch_inner) {
    return gemm_impl(dev, transA, transB, M, N, K, alpha, A, lda, sAo, sAi, B, ldb, sBo, sBi, beta, C, ldc,
                     sCo, sCi, batch_outer, batch_inner);
}

// ---- node-level wrappers: one per reference forward()/backward() body ------------------------
int nk_mm_fwd(nk_device* dev, const float* A, const float* B, float* C, int n, int m, int o) {
    return nk_sgemm(dev, 0, 0, n, o, m, 1.f, A, m, B, o, 0.f, C, o);
}
// one product of a MatMul node's backward pass: the plan of nk_mm_bwd's products (no k-pair blocks), so that a gradient's bits do
// not depend on whether the node's other operand is differentiable
static int node_bwd_product(nk_device* dev, int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                            float* C, int ldc) {
    return gemm_impl(dev, transA, transB, M, N, K, 1.f, A, lda, 0, 0, B, ldb, 0, 0, 1.f, C, ldc, 0, 0, 1, 1, nullptr, 0, nullptr, 0, false);
}
int nk_mm_bwd_left(nk_device* dev, float* dA, const float* G, const float* B, int n, int m, int o) {
    return node_bwd_product(dev, 0, 1, n, m, o, G, o, B, o, dA, m);  // dA += G . B^T
}
int nk_mm_bwd_right(nk_device* dev, float* dB, const float* A, const float* G, int n, int m, int o) {
    return node_bwd_product(dev, 1, 0, m, o, n, A, m, G, o, dB, o);  // dB += A^T . G
}
int nk_sgemm_pair(nk_device* dev, int transA0, int transB0, int M0, int N0, int K0, const float* A0, int lda0, const float* B0, int ldb0,
                  float beta0, float* C0, int ldc0, int transA1, int transB1, int M1, int N1, int K1, const float* A1, int lda1,
                  const float* B1, int ldb1, float beta1, float* C1, int ldc1) {
    const GemmProblem a{transA0, transB0, M0, N0, K0, A0, lda0, 0, 0, B0, ldb0, 0, 0, beta0, C0, ldc0, 0, 0};
    const GemmProblem b{transA1, transB1, M1, N1, K1, A1, lda1, 0, 0, B1, ldb1, 0, 0, beta1, C1, ldc1, 0, 0};
    return gemm_pair_impl(dev, a, b, 1, 1);
}
int nk_sgemm_pair_batched(nk_device* dev, int batch_outer, int batch_inner,
                          int transA0, int transB0, int M0, int N0, int K0, const float* A0, int lda0, long long sA0o, long long sA0i,
                          const float* B0, int ldb0, long long sB0o, long long sB0i, float beta0, float* C0, int ldc0, long long sC0o, long long sC0i,
                          int transA1, int transB1, int M1, int N1, int K1, const float* A1, int lda1, long long sA1o, long long sA1i,
                          const float* B1, int ldb1, long long sB1o, long long sB1i, float beta1, float* C1, int ldc1, long long sC1o, long long sC1i) {
    const GemmProblem a{transA0, transB0, M0, N0, K0, A0, lda0, sA0o, sA0i, B0, ldb0, sB0o, sB0i, beta0, C0, ldc0, sC0o, sC0i};
    const GemmProblem b{transA1, transB1, M1, N1, K1, A1, lda1, sA1o, sA1i, B1, ldb1, sB1o, sB1i, beta1, C1, ldc1, sC1o, sC1i};
    return gemm_pair_impl(dev, a, b, batch_outer, batch_inner);
}
// MatrixMatrixMulBackward::backward (both operands differentiable): dA (+)= G . B^T and dB (+)= A^T . G
int nk_mm_bwd(nk_device* dev, float* dA, float* dB, const float* G, const float* A, const float* B, int n, int m, int o, int assign_a,
              int assign_b) {
    return nk_sgemm_pair(dev, 0, 1, n, m, o, G, o, B, o, assign_a ? 0.f : 1.f, dA, m, 1, 0, m, o, n, A, m, G, o, assign_b ? 0.f : 1.f, dB, o);
}
// MatrixMatrixMulTBackward::backward: dA (+)= G . B and dB (+)= G^T . A, B is (o, m)
int nk_mm_t_bwd(nk_device* dev, float* dA, float* dB, const float* G, const float* A, const float* B, int n, int m, int o, int assign_a,
                int assign_b) {
    return nk_sgemm_pair(dev, 0, 0, n, m, o, G, o, B, m, assign_a ? 0.f : 1.f, dA, m, 1, 0, o, m, n, G, o, A, m, assign_b ? 0.f : 1.f, dB, m);
}
int nk_mm_t_fwd(nk_device* dev, const float* A, const float* B, float* C, int n, int m, int o) {
    return nk_sgemm(dev, 0, 1, n, o, m, 1.f, A, m, B, m, 0.f, C, o);  // C = A . B^T, B is (o,m)
}
int nk_linear_fwd(nk_device* dev, const float* X, const float* W, const float* bias, float* Y, int n, int m, int o) {
    NK_CHECK(bias != nullptr, "null bias");
    return gemm_impl(dev, 0, 1, n, o, m, 1.f, X, m, 0, 0, W, m, 0, 0, 0.f, Y, o, 0, 0, 1, 1, bias);  // Y = X . W^T + b
}
int nk_linear_relu_fwd(nk_device* dev, const float* X, const float* W, const float* bias, float* Y, int n, int m, int o) {
    NK_CHECK(bias != nullptr, "null bias");
    return gemm_impl(dev, 0, 1, n, o, m, 1.f, X, m, 0, 0, W, m, 0, 0, 0.f, Y, o, 0, 0, 1, 1, bias, 1);  // Y = max(X . W^T + b, 0)
}
int nk_linear_bwd_input_relu(nk_device* dev, float* dX, const float* G, const float* W, const float* X, int n, int m, int o, int assign) {
    NK_CHECK(X != nullptr, "null mask operand");
    // dX (+)= (G . W) where X > 0, 0 * (G . W) elsewhere: MatrixMatrixMulTBackwardLeft followed by ReLUBackward of the node that made X
    return gemm_impl(dev, 0, 0, n, m, o, 1.f, G, o, 0, 0, W, m, 0, 0, assign ? 0.f : 1.f, dX, m, 0, 0, 1, 1, nullptr, 0, X, m);
}
int nk_mm_t_bwd_left(nk_device* dev, float* dA, const float* G, const float* B, int n, int m, int o) {
    return node_bwd_product(dev, 0, 0, n, m, o, G, o, B, m, dA, m);  // dA += G . B
}
int nk_mm_t_bwd_right(nk_device* dev, float* dB, const float* G, const float* A, int n, int m, int o) {
    return node_bwd_product(dev, 1, 0, o, m, n, G, o, A, m, dB, m);  // dB += G^T . A
}

}  // extern "C"
