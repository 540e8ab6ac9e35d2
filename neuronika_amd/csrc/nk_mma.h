// f32 MFMA tile core shared by the GEMM and implicit-GEMM convolution kernels (gfx950).
//
// A block is 256 threads = 4 waves arranged 2 x 2; wave (wr, wc) owns TI x TJ MFMA tiles of
// 32 x 32 (v_mfma_f32_32x32x2_f32: exact f32 fma chain at the f32 vector rate, 64 cycles per
// instruction per SIMD; MI355X_MICROARCH.md "Matrix cores"), so the block tile is
// (64*TI) x (64*TJ) x 32 with TI, TJ in {1, 2}: 128x128 for large problems, 64-wide variants for
// operands that are only 64 wide (attention head dim, Cin = 64 convolutions) or too small to
// fill 256 CUs with 128x128 tiles.
//
// LDS image of an operand tile (R rows x 32 k) depends on how the operand lies in HBM, so
// that BOTH the global load and the LDS store stay 16-byte vectors and nothing is transposed
// on the way in:
//   KC  "k-contiguous"   (element (row,k) at X[row*ld + k]):  Xs[row][k], row stride 36 floats.
//        A lane reads its 4 k-values with ONE ds_read_b128; stride 36 (=9 x 16 B) makes the
//        16-lane b128 groups hit 16 distinct 16-B slots (conflict-free).
//   RC  "row-contiguous" (element (row,k) at X[k*ld + row]):  Xs[k][row], k stride R floats.
//        A lane reads 4 ds_read_b32; 32 consecutive lanes read 32 consecutive floats.
// The MFMA consumes k in pairs (lanes 0-31 supply k, lanes 32-63 supply k'), and any pairing
// works as long as A and B agree; we use, inside each group of 8 k:  step i -> lanes<32: k=i,
// lanes>=32: k=4+i, which is exactly what one b128 read of a KC row delivers.
//
// (An LDS-DMA variant — `global_load_lds_dwordx4` into an XOR-swizzled unpadded image, waits
// placed by hand — was built and measured in round 1: same speed as register staging for this
// f32 kernel, 134.7 vs 135.9 TFLOP/s at 4096^3, so it was dropped; see DESIGN.md.)
#pragma once
#include "nk_common.h"

namespace nkmma {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// 16-byte vector with 4-byte alignment: `global_load_dwordx4` at any dword address (gathers)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BK = 32;
constexpr int NT = 256;
constexpr int LDK = BK + 4;  // KC row stride (floats)

// floats of one operand tile with R rows: the KC image is padded (stride 36), the RC image is dense.
// Sizing per layout (not the max of both) is what lets the 64x128 tiles with at least one RC operand keep
// THREE blocks per CU: 2*(64*36 + 128*32)*4 B = 51,200 B  vs  55,296 B (> 160 KB / 3) for two padded images.
template <bool KC, int R>
constexpr int tile_floats() { return KC ? R * LDK : R * BK; }
// blocks per CU the register/LDS budget is sized for -> min waves per SIMD for launch bounds.  A 64x128 tile whose two
// operands are BOTH k-contiguous has two padded images (55,296 B > 160 KB / 3): two blocks per CU by LDS, so its register
// budget is the two-wave one as well (asking for three only made the compiler miss the target and warn).
template <int TI, int TJ, bool BOTH_KC = false>
constexpr int min_waves() { return TI * TJ == 4 ? 2 : (TI * TJ == 2 ? (BOTH_KC ? 2 : 3) : 4); }

// ---- register staging of one operand tile: R/32 float4 per thread ------------------------------
// KC: idx = t + 256*j -> kq = idx&7 (8 lanes cover one row's 128 B), row = kc_row(idx): lanes 8-15 of every
//     16-lane group take the row 8 below lanes 0-7, so the group's ds_write_b128 lands in 16 distinct 16-B
//     slots of the stride-36 image (rows r and r+1 would collide in one slot: 9*1 + 7 = 16 = 0 mod 16).
// RC: idx = t + 256*j -> k = idx/(R/4), rq = idx%(R/4)     (R/4 lanes cover one k-row)
// (named members, returned by value: keeps the staging registers out of scratch.)
template <int NV>
struct Stage;
template <>
struct Stage<4> {
    float4 v0, v1, v2, v3;
};
template <>
struct Stage<2> {
    float4 v0, v1;
};

__device__ __forceinline__ int kc_row(int idx) { return ((idx >> 7) << 4) + (((idx >> 3) & 1) << 3) + ((idx >> 4) & 7); }
__device__ __forceinline__ int kc_q(int idx) { return idx & 7; }

template <bool KC, int R>
__device__ __forceinline__ int lds_slot(int idx) {
    return KC ? kc_row(idx) * LDK + kc_q(idx) * 4 : (idx / (R / 4)) * R + (idx % (R / 4)) * 4;
}

template <bool KC, int R>
__device__ __forceinline__ void stage_store(float* Xs, const Stage<R / 32>& r, int t) {
    *reinterpret_cast<float4*>(&Xs[lds_slot<KC, R>(t)]) = r.v0;
    *reinterpret_cast<float4*>(&Xs[lds_slot<KC, R>(t + NT)]) = r.v1;
    if constexpr (R == 128) {
        *reinterpret_cast<float4*>(&Xs[lds_slot<KC, R>(t + 2 * NT)]) = r.v2;
        *reinterpret_cast<float4*>(&Xs[lds_slot<KC, R>(t + 3 * NT)]) = r.v3;
    }
}

// Dense-matrix loader.  Element (row,k) of the operand lies at KC: X[row*ld + k],
// RC: X[k*ld + row].  The fast path keeps ONE wave-uniform tile pointer (advanced per k-tile)
// plus per-thread 32-bit element offsets, so each load is `global_load_dwordx4 v, voff, s[base]`.
template <bool KC, int R>
struct TileLoader {
    const float* base;  // &X[tile origin] for the current k-tile (wave-uniform)
    long long kstep;    // elements to advance per k-tile
    unsigned o0, o1, o2, o3;
    // guarded-path state (problems that are not tile / 16-byte aligned)
    const float* X;
    long long ld;
    int row0, rows, kend, k0;
    bool interior;  // every row of the tile exists (block-uniform): whole k-tiles take unaligned 16-byte loads

    __device__ __forceinline__ void init(const float* X_, long long ld_, int row0_, int k0_, int rows_,
                                         int kend_, int t) {
        X = X_; ld = ld_; row0 = row0_; rows = rows_; kend = kend_; k0 = k0_;
        interior = row0_ + R <= rows_;
        base = KC ? X_ + (long long)row0_ * ld_ + k0_ : X_ + (long long)k0_ * ld_ + row0_;
        kstep = KC ? BK : (long long)BK * ld_;
        o0 = off(t, ld_); o1 = off(t + NT, ld_); o2 = off(t + 2 * NT, ld_); o3 = off(t + 3 * NT, ld_);
    }
    static __device__ __forceinline__ unsigned off(int idx, long long ld_) {
        return KC ? (unsigned)(kc_row(idx) * ld_ + kc_q(idx) * 4)
                  : (unsigned)((idx / (R / 4)) * ld_ + (idx % (R / 4)) * 4);
    }
    __device__ __forceinline__ float4 guarded(int idx) const {
        float v[4];
        if (KC) {
            const int row = row0 + kc_row(idx), k = k0 + kc_q(idx) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (row < rows && k + c < kend) ? X[row * ld + k + c] : 0.f;
        } else {
            const int k = k0 + idx / (R / 4), row = row0 + (idx % (R / 4)) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (k < kend && row + c < rows) ? X[k * ld + row + c] : 0.f;
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
    // loads the current k-tile and advances to the next one
    template <bool ALIGNED>
    __device__ __forceinline__ Stage<R / 32> load(int t) {
        Stage<R / 32> r;
        if (ALIGNED) {
            r.v0 = *reinterpret_cast<const float4*>(base + o0);
            r.v1 = *reinterpret_cast<const float4*>(base + o1);
            if constexpr (R == 128) {
                r.v2 = *reinterpret_cast<const float4*>(base + o2);
                r.v3 = *reinterpret_cast<const float4*>(base + o3);
            }
            base += kstep;
        } else if (interior && k0 + BK <= kend) {
            // ragged problem, but this tile and this k-tile are whole: 16-byte loads that only need 4-byte alignment
            // (an odd leading dimension or base pointer costs a few split cache lines, not the 32 guarded scalar loads)
#define NK_LDU(V, O) { const f32x4u q = *reinterpret_cast<const f32x4u*>(base + O); V = make_float4(q.x, q.y, q.z, q.w); }
            NK_LDU(r.v0, o0) NK_LDU(r.v1, o1)
            if constexpr (R == 128) { NK_LDU(r.v2, o2) NK_LDU(r.v3, o3) }
#undef NK_LDU
            base += kstep;
            k0 += BK;
        } else {
            r.v0 = guarded(t);
            r.v1 = guarded(t + NT);
            if constexpr (R == 128) {
                r.v2 = guarded(t + 2 * NT);
                r.v3 = guarded(t + 3 * NT);
            }
            base += kstep;
            k0 += BK;
        }
        return r;
    }
};

// ---- fragment reads + MFMA over one staged tile -------------------------------------------------
template <bool KC, int R>
__device__ __forceinline__ void frag_read(float (&f)[4], const float* Xs, int rowbase, int k8, int lane) {
    const int r = lane & 31, h = lane >> 5;
    if (KC) {
        const float4 v = *reinterpret_cast<const float4*>(&Xs[(rowbase + r) * LDK + k8 + 4 * h]);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = Xs[(k8 + 4 * h + i) * R + rowbase + r];
    }
}

// Fragments of k-group g+1 are read from LDS BEFORE the MFMAs of group g are issued (the
// sched_barrier pins that order), so the ds_read latency hides under the MFMA cycles.
// [G0, G1): the k-groups (8 k each) of the staged tile this call covers - the whole tile by default; the k-pair GEMM's
// second wave group issues a tile in two halves around its staging stores.
template <bool AKC, bool BKC, int TI, int TJ, int G0 = 0, int G1 = BK / 8>
__device__ __forceinline__ void mma_tile(const float* As, const float* Bs, f32x16 (&acc)[TI][TJ], int wr, int wc,
                                         int lane) {
    float a[2][TI][4], b[2][TJ][4];  // [buffer][tile][step]
#pragma unroll
    for (int i = 0; i < TI; ++i) frag_read<AKC, 64 * TI>(a[G0 & 1][i], As, (wr * TI + i) * 32, G0 * 8, lane);
#pragma unroll
    for (int j = 0; j < TJ; ++j) frag_read<BKC, 64 * TJ>(b[G0 & 1][j], Bs, (wc * TJ + j) * 32, G0 * 8, lane);
#pragma unroll
    for (int g = G0; g < G1; ++g) {
        const int cb = g & 1, nb = cb ^ 1;
        if (g + 1 < G1) {
#pragma unroll
            for (int i = 0; i < TI; ++i) frag_read<AKC, 64 * TI>(a[nb][i], As, (wr * TI + i) * 32, (g + 1) * 8, lane);
#pragma unroll
            for (int j = 0; j < TJ; ++j) frag_read<BKC, 64 * TJ>(b[nb][j], Bs, (wc * TJ + j) * 32, (g + 1) * 8, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][i][s], b[cb][j][s], acc[i][j], 0, 0, 0);
    }
}

template <int TI, int TJ>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[TI][TJ]) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Calls f(row_in_block_tile, col_in_block_tile, value) for the values this lane owns.
template <int TI, int TJ, class F>
__device__ __forceinline__ void acc_foreach(const f32x16 (&acc)[TI][TJ], int wr, int wc, int lane, F&& f) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wr * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int col = (wc * TJ + j) * 32 + c;
                f(row, col, acc[i][j][e]);
            }
}

// Same walk, column first: `fc(col)` is evaluated once per owned column (a lane owns TJ columns and 16*TI rows of each),
// `fe(row, ctx, value)` per element - for epilogues whose column -> address decode is expensive (integer divisions).
template <int TI, int TJ, class FC, class FE>
__device__ __forceinline__ void acc_foreach_cols(const f32x16 (&acc)[TI][TJ], int wr, int wc, int lane, FC&& fc, FE&& fe) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const auto ctx = fc((wc * TJ + j) * 32 + c);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) fe((wr * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, ctx, acc[i][j][e]);
    }
}

// Pins the point where loaded registers may first be touched: an empty asm that "modifies" them.  Masks / selects written
// after it cannot be hoisted above it by the optimiser (they are pure, and __builtin_amdgcn_sched_barrier only constrains
// the machine scheduler), so the wait for the loads lands behind whatever precedes the pin (the MFMA block).  Only where
// the ISA shows the hoisting: a pin is a full wait at one point and costs 2-4 % where the compiler already interleaves the
// progressive waits and the selects with the MFMA tail by itself (measured on the fast conv kernels).
__device__ __forceinline__ void pin_regs(float4& q) { asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w)); }

// Index-passing forms: f(i, j, e, row, col, value) / fe(i, j, e, row, ctx, value) with i, j, e compile-time constants after
// unrolling, so an epilogue can keep per-element state in a register array filled by a FIRST walk (all loads issued) and
// consumed by a SECOND walk (stores).  A one-walk read-modify-write (`*q = f(*q)`) or a bias load between stores
// serialises 16*TI*TJ load -> store round trips per lane: a store may alias the next load, so the compiler keeps the order.
template <int TI, int TJ, class F>
__device__ __forceinline__ void acc_foreach_idx(const f32x16 (&acc)[TI][TJ], int wr, int wc, int lane, F&& f) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                f(i, j, e, (wr * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, (wc * TJ + j) * 32 + c, acc[i][j][e]);
}
template <int TI, int TJ, class FC, class FE>
__device__ __forceinline__ void acc_foreach_cols_idx(const f32x16 (&acc)[TI][TJ], int wr, int wc, int lane, FC&& fc, FE&& fe) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const auto ctx = fc((wc * TJ + j) * 32 + c);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) fe(i, j, e, (wr * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, ctx, acc[i][j][e]);
    }
}

// XCD-aware, L2-friendly block -> tile mapping.  Blocks are dispatched round-robin over the 8
// XCDs (block b -> XCD b%8, observed; used for speed only): give each XCD a contiguous chunk
// of the tile sequence, and order the sequence in column-major groups of GROUP_M tile rows so
// that the tiles resident on one XCD at a time share A row-panels and B column-panels in its
// private 4 MiB L2 (cdna_hip_programming.md T1; bijective form).
// first half of tile_coords: block id -> position in the tile sequence (each XCD gets a contiguous chunk)
__device__ __forceinline__ int xcd_chunk(int bid, int nblk) {
    constexpr int NXCD = 8;
    const int q = nblk / NXCD, rem = nblk % NXCD;
    const int xcd = bid % NXCD, loc = bid / NXCD;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
}
// second half: position in the sequence -> tile (column-major groups of GROUP_M tile rows)
__device__ __forceinline__ void tile_of_seq(int swz, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP_M = 8) {
    const int per_group = GROUP_M * tiles_n;
    const int group = swz / per_group;
    const int first_m = group * GROUP_M;
    const int gsize = min(tiles_m - first_m, GROUP_M);
    const int in_group = swz % per_group;
    tm = first_m + in_group % gsize;
    tn = in_group / gsize;
}
__device__ __forceinline__ void tile_coords(int bid, int nblk, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int NXCD = 8;
    const int q = nblk / NXCD, rem = nblk % NXCD;
    const int xcd = bid % NXCD, loc = bid / NXCD;
    const int swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int group = swz / per_group;
    const int first_m = group * GROUP_M;
    const int gsize = min(tiles_m - first_m, GROUP_M);
    const int in_group = swz % per_group;
    tm = first_m + in_group % gsize;
    tn = in_group / gsize;
}

}  // namespace nkmma
