// f32 MFMA tile core shared by the GEMM and implicit-GEMM convolution kernels (gfx950).
//
// Block tile 128 x 128 x 32, 256 threads = 4 waves arranged 2 x 2, each wave owns a 64 x 64
// sub-tile = 2 x 2 MFMA tiles of 32 x 32 (v_mfma_f32_32x32x2_f32: exact f32 fma chain at the
// f32 vector rate, 64 cycles/instruction/SIMD; MI355X_MICROARCH.md "Matrix cores").
//
// LDS image of an operand tile (R rows x 32 k) depends on how the operand lies in HBM, so
// that BOTH the global load and the LDS store stay 16-byte vectors and nothing is transposed
// on the way in:
//   KC  "k-contiguous"   (element (row,k) at X[row*ld + k]):  Xs[row][k], row stride 36 floats.
//        A lane reads its 4 k-values with ONE ds_read_b128; stride 36 (=9 x 16 B) makes the
//        16-lane b128 groups hit 16 distinct 16-B slots (conflict-free).
//   RC  "row-contiguous" (element (row,k) at X[k*ld + row]):  Xs[k][row], k stride 128 floats.
//        A lane reads 4 ds_read_b32; 32 consecutive lanes read 32 consecutive floats.
// The MFMA consumes k in pairs (lanes 0-31 supply k, lanes 32-63 supply k'), and any pairing
// works as long as A and B agree; we use, inside each group of 8 k:  step i -> lanes<32: k=i,
// lanes>=32: k=4+i, which is exactly what one b128 read of a KC row delivers.
#pragma once
#include "nk_common.h"

namespace nkmma {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 32;
constexpr int NT = 256;
constexpr int LDK = BK + 4;           // KC row stride (floats)
constexpr int LDR = 128;              // RC k stride (floats)
constexpr int TILE_FLOATS = 128 * LDK;  // >= BK*LDR, one operand tile
constexpr int STAGE_FLOATS = 2 * TILE_FLOATS;
constexpr int SMEM_BYTES = 2 * STAGE_FLOATS * 4;  // 73,728 B -> 2 blocks per CU

// ---- register staging of one operand tile: 4 float4 per thread -----------------------------
// KC: idx = t + 256*j -> row = idx>>3, kq = idx&7   (8 lanes cover one row's 128 contiguous B)
// RC: idx = t + 256*j -> k   = idx>>5, rq = idx&31  (32 lanes cover one k-row's 512 B)
// (named members, not arrays: keeps the staging registers out of scratch.)
struct Stage {
    float4 v0, v1, v2, v3;
};

template <bool KC>
__device__ __forceinline__ int lds_slot(int idx) {
    return KC ? (idx >> 3) * LDK + (idx & 7) * 4 : (idx >> 5) * LDR + (idx & 31) * 4;
}

template <bool KC>
__device__ __forceinline__ void stage_store(float* Xs, const Stage& r, int t) {
    *reinterpret_cast<float4*>(&Xs[lds_slot<KC>(t)]) = r.v0;
    *reinterpret_cast<float4*>(&Xs[lds_slot<KC>(t + NT)]) = r.v1;
    *reinterpret_cast<float4*>(&Xs[lds_slot<KC>(t + 2 * NT)]) = r.v2;
    *reinterpret_cast<float4*>(&Xs[lds_slot<KC>(t + 3 * NT)]) = r.v3;
}

// Dense-matrix loader.  Element (row,k) of the operand lies at KC: X[row*ld + k],
// RC: X[k*ld + row].  The fast path keeps ONE wave-uniform tile pointer (advanced per k-tile)
// plus four per-thread 32-bit element offsets, so each load is `global_load_dwordx4 v, voff, s[base]`.
template <bool KC>
struct TileLoader {
    const float* base;      // &X[tile origin] for the current k-tile (wave-uniform)
    long long kstep;        // elements to advance per k-tile
    unsigned o0, o1, o2, o3;
    // slow-path state
    const float* X;
    long long ld;
    int row0, rows, kend, k0;

    __device__ __forceinline__ void init(const float* X_, long long ld_, int row0_, int k0_, int rows_,
                                         int kend_, int t) {
        X = X_; ld = ld_; row0 = row0_; rows = rows_; kend = kend_; k0 = k0_;
        if (KC) {
            base = X_ + (long long)row0_ * ld_ + k0_;
            kstep = BK;
            o0 = off(t, ld_); o1 = off(t + NT, ld_); o2 = off(t + 2 * NT, ld_); o3 = off(t + 3 * NT, ld_);
        } else {
            base = X_ + (long long)k0_ * ld_ + row0_;
            kstep = (long long)BK * ld_;
            o0 = off(t, ld_); o1 = off(t + NT, ld_); o2 = off(t + 2 * NT, ld_); o3 = off(t + 3 * NT, ld_);
        }
    }
    static __device__ __forceinline__ unsigned off(int idx, long long ld_) {
        return KC ? (unsigned)((idx >> 3) * ld_ + (idx & 7) * 4) : (unsigned)((idx >> 5) * ld_ + (idx & 31) * 4);
    }
    __device__ __forceinline__ float4 guarded(int idx) const {
        float v[4];
        if (KC) {
            const int row = row0 + (idx >> 3), k = k0 + (idx & 7) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (row < rows && k + c < kend) ? X[row * ld + k + c] : 0.f;
        } else {
            const int k = k0 + (idx >> 5), row = row0 + (idx & 31) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (k < kend && row + c < rows) ? X[k * ld + row + c] : 0.f;
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
    // loads the current k-tile into `r` and advances to the next one
    template <bool ALIGNED>
    __device__ __forceinline__ Stage load(int t) {
        Stage r;
        if (ALIGNED) {
            r.v0 = *reinterpret_cast<const float4*>(base + o0);
            r.v1 = *reinterpret_cast<const float4*>(base + o1);
            r.v2 = *reinterpret_cast<const float4*>(base + o2);
            r.v3 = *reinterpret_cast<const float4*>(base + o3);
            base += kstep;
        } else {
            r.v0 = guarded(t);
            r.v1 = guarded(t + NT);
            r.v2 = guarded(t + 2 * NT);
            r.v3 = guarded(t + 3 * NT);
            k0 += BK;
        }
        return r;
    }
};

// ---- fragment reads + MFMA over one staged 128 x 128 x 32 tile ---------------------------------
template <bool KC>
__device__ __forceinline__ void frag_read(float (&f)[4], const float* Xs, int rowbase, int k8, int lane) {
    const int r = lane & 31, h = lane >> 5;
    if (KC) {
        const float4 v = *reinterpret_cast<const float4*>(&Xs[(rowbase + r) * LDK + k8 + 4 * h]);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = Xs[(k8 + 4 * h + i) * LDR + rowbase + r];
    }
}

// ---- direct global->LDS staging (LDS-DMA) --------------------------------------------------------
// `global_load_lds_dwordx4` writes wave-uniform-base + lane*16 B, so the LDS image must be
// lane-linear: KC tiles become UNPADDED [row][32] with the 16-B k-quad XOR-swizzled by (row & 7)
// on the SOURCE address (and the same XOR on the read: cdna_hip_programming.md rule 21); RC tiles
// [k][128] are lane-linear as they are.  No staging VGPRs, no ds_write instructions.
constexpr int LDKS = BK;                     // swizzled KC row stride (floats)
constexpr int GTILE_FLOATS = 128 * BK;       // one operand tile, 16 KiB
constexpr int GSTAGE_FLOATS = 2 * GTILE_FLOATS;

template <bool KC>
struct GldsLoader {
    const float* base;   // wave-uniform tile origin, advanced per k-tile
    long long kstep;
    unsigned o0, o1, o2, o3;  // per-thread element offsets of the four 16-B pieces
    __device__ __forceinline__ void init(const float* X, long long ld, int row0, int k0, int t) {
        base = KC ? X + (long long)row0 * ld + k0 : X + (long long)k0 * ld + row0;
        kstep = KC ? BK : (long long)BK * ld;
        o0 = off(t, ld); o1 = off(t + NT, ld); o2 = off(t + 2 * NT, ld); o3 = off(t + 3 * NT, ld);
    }
    static __device__ __forceinline__ unsigned off(int idx, long long ld) {
        if (KC) {
            const int row = idx >> 3, kq = (idx & 7) ^ (row & 7);
            return (unsigned)(row * ld + kq * 4);
        }
        return (unsigned)((idx >> 5) * ld + (idx & 31) * 4);
    }
    // One asynchronous 16-B-per-lane copy HBM/L2 -> LDS.  Inline asm on purpose: hipcc would
    // otherwise wait vmcnt(0) before the next LDS read (it cannot tell which LDS bytes a DMA
    // writes) and serialise the prefetch; here the wait is placed by hand, once per k-tile, in
    // front of the barrier (cdna_hip_programming.md 5.7: M0 is set and restored inside the asm).
    static __device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_byte_addr) {
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_byte_addr)
            : "memory");
    }
    // issue the four copies of the current k-tile into the tile at LDS byte address `tile_addr`
    // (wave-uniform) and advance to the next k-tile
    __device__ __forceinline__ void issue(unsigned tile_addr, int t) {
        const unsigned wbase = __builtin_amdgcn_readfirstlane(tile_addr + (unsigned)(t & ~63) * 16u);
        glds16(base + o0, wbase);
        glds16(base + o1, wbase + NT * 16u);
        glds16(base + o2, wbase + 2u * NT * 16u);
        glds16(base + o3, wbase + 3u * NT * 16u);
        base += kstep;
    }
};

// LDS byte address of a __shared__ object (for M0)
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

template <bool KC>
__device__ __forceinline__ void frag_read_g(float (&f)[4], const float* Xs, int rowbase, int k8, int lane) {
    const int r = lane & 31, h = lane >> 5;
    if (KC) {
        const int row = rowbase + r, q = (k8 >> 2) + h;
        const float4 v = *reinterpret_cast<const float4*>(&Xs[row * LDKS + ((q ^ (row & 7)) << 2)]);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = Xs[(k8 + 4 * h + i) * LDR + rowbase + r];
    }
}

template <bool AKC, bool BKC>
__device__ __forceinline__ void mma_tile_g(const float* As, const float* Bs, f32x16 (&acc)[2][2],
                                           int wr, int wc, int lane) {
    float a[2][2][4], b[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) frag_read_g<AKC>(a[0][i], As, wr * 64 + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) frag_read_g<BKC>(b[0][j], Bs, wc * 64 + j * 32, 0, lane);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
        const int cb = g & 1, nb = cb ^ 1;
        if (g + 1 < BK / 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) frag_read_g<AKC>(a[nb][i], As, wr * 64 + i * 32, (g + 1) * 8, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) frag_read_g<BKC>(b[nb][j], Bs, wc * 64 + j * 32, (g + 1) * 8, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][i][s], b[cb][j][s], acc[i][j], 0, 0, 0);
    }
}

// Fragments of k-group g+1 are read from LDS BEFORE the 16 MFMAs of group g are issued (the
// sched_barrier pins that order), so the ds_read latency hides under 16 x 64 MFMA cycles.
template <bool AKC, bool BKC, bool PRIO = false>
__device__ __forceinline__ void mma_tile(const float* As, const float* Bs, f32x16 (&acc)[2][2],
                                         int wr, int wc, int lane) {
    float a[2][2][4], b[2][2][4];  // [buffer][tile][step]
#pragma unroll
    for (int i = 0; i < 2; ++i) frag_read<AKC>(a[0][i], As, wr * 64 + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) frag_read<BKC>(b[0][j], Bs, wc * 64 + j * 32, 0, lane);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
        const int cb = g & 1, nb = cb ^ 1;
        if (g + 1 < BK / 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) frag_read<AKC>(a[nb][i], As, wr * 64 + i * 32, (g + 1) * 8, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) frag_read<BKC>(b[nb][j], Bs, wc * 64 + j * 32, (g + 1) * 8, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][i][s], b[cb][j][s], acc[i][j], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
}

__device__ __forceinline__ void acc_zero(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Calls f(row_in_block_tile, col_in_block_tile, value) for the 64 values this lane owns.
template <class F>
__device__ __forceinline__ void acc_foreach(const f32x16 (&acc)[2][2], int wr, int wc, int lane, F&& f) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int col = wc * 64 + j * 32 + c;
                f(row, col, acc[i][j][e]);
            }
}

// XCD-aware, L2-friendly block -> tile mapping.  Blocks are dispatched round-robin over the 8
// XCDs (block b -> XCD b%8, observed; used for speed only): give each XCD a contiguous chunk
// of the tile sequence, and order the sequence in column-major groups of GROUP_M tile rows so
// that the tiles resident on one XCD at a time share A row-panels and B column-panels in its
// private 4 MiB L2 (cdna_hip_programming.md T1; bijective form).
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
