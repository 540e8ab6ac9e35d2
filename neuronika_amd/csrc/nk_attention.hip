// Fused attention core of the composed multi-head attention (note below the table: the module is a
// composition of MatrixMatrixMulT a-2, Multiplication a-4, Softmax a-7, Dropout a-8, MatrixMatrixMul a-1 per (sample, head)):
//
//   forward   S = Q_bh . K_bh^T ;  P = softmax(S * scale, axis 1) ;  Pd = dropout(P) ;  O_bh = Pd . V_bh
//   backward  dPd = dO_bh . V_bh^T ;  dP = dPd * noise            (DropoutBackward, node/dropout/mod.rs:113-128: mask only)
//             dS  = P * (dP - sum_k dP * P) * scale               (SoftmaxBackward :84-104, MultiplicationBackwardLeft)
//             dQ_bh += dS . K_bh          [dK_bh += dS^T . Q_bh and dV_bh += Pd^T . dO_bh stay batched GEMMs on dS / Pd]
//
// Why one kernel per direction: composed from GEMMs and row kernels the (B*H, S, S) tensors cross HBM 11 times per step
// (2.1 GB each at C5) and the K = 64 GEMMs that write them run at half the MFMA rate because 2.1 GB of C stores queue
// behind 64-deep reductions.  Here a block owns 128 queries of one (sample, head), walks the keys in tiles of 32 and keeps
// the 32 x 32 score tile ON CHIP between the two MFMA products; the big tensors cross HBM 6 times (forward writes S;
// backward reads S, writes dS and Pd; the two remaining GEMMs read dS and Pd).
//
// Layout trick (what keeps the score tile in registers): the tile is computed TRANSPOSED, C[key][query] = X1 . Bq^T, so a
// lane owns one query (column = lane & 31) and 16 keys of it.  Row statistics (max, sum, the backward dot) are then
// per-lane scalars plus one cross-half shuffle, and the C registers are directly the B operand of the second product
// out^T[dh][query] = X2^T . C (the MFMA wants B[k][n] with n = lane & 31, k chosen by the lane half: exactly what C holds
// when MFMA step e pairs key 16*0 + e with key 16*1 + e).  Rows of the MFMA tile are permuted so that a lane's 16
// registers are 16 CONSECUTIVE keys: two Philox calls of the shared draw layout (8 consecutive elements per call)
// and 64-byte runs towards HBM fall out of that.
//
// The forward uses the online softmax in the base-2 exponent domain (running shift m2 and sum of exp2(s*c1 - m2), c1 =
// scale * log2(e); the shift follows the running maximum lazily - it only has to keep the exponents <= 6 - and the out
// accumulator is rescaled when it moves) and stores (m2, 1 / sum) per row; the backward recomputes
// P = exp2(S * c1 - m2) / sum from the stored scores with those.  Against the node-by-node composition this changes the order
// of the row sum and replaces two divisions by a reciprocal and a product - inside the f32 tolerance of SURVEY.md 8c (ii),
// checked against the oracle, not bit-equal to the three-node path.
#include <cmath>
#include <cstdlib>

#include "nk_mma.h"

namespace {

using nkmma::f32x16;

// Head dimension DH in {32, 64, 128} (template parameter): DH / 32 MFMA column tiles of the output, DH / 8 b128 k-groups of the
// score product.  64 is the C5 geometry the register / LDS budgets were tuned for (three forward blocks per CU); 128 holds 64 + 64
// VGPRs of per-query operand and output accumulators and 66 KB of staged K / V per block - one block per CU, still one kernel per
// direction instead of the node-by-node path; 32 halves everything.
constexpr int A_NT = 256;   // 4 waves, 32 queries each
constexpr int A_QB = 128;   // queries per block
constexpr int SCR_LD = 36;  // per-wave 32 x 32 transposition scratch
// pass-1 operand image [mfma row][dh], padded by 4: b128 fragment reads of 16 rows hit 16 distinct slots
constexpr int x1_ld(int dh) { return dh + 4; }
// pass-2 operand image [key][dh rotated by 32 for keys >= 16]: the two lane halves read disjoint banks (DH = 32: no rotation
// possible, the halves share banks - a two-way conflict on 16 of the tile's LDS reads)
constexpr int x2_ld(int dh) { return dh; }

struct AttnArgs {
    const float* x1;   // pass-1 operand, flat (B*S) x (H*dh) layout: forward K, backward V
    const float* x2;   // pass-2 operand:                               forward V, backward K
    const float* bq;   // per-query operand:                            forward Q, backward dO
    const float* ctx;  // backward only: the forward output O (for sum_k dP * P = keep * dO . O)
    float* out;        // forward O, backward dQ
    float* scores;     // (B*H, S, S) raw scores: written forward, read backward
    float* ds;         // backward: dS
    float* dropped;    // backward: Pd
    unsigned* maskbits;  // (B*H, S/32 query tiles, S/32 key tiles, 32 queries) words: the dropout draws, 1 bit per score (bit 16h + e of
                         // word [bh][qt][kt][q] = key 32 kt + 16 h + e of query 32 qt + q kept): written forward, read backward.  A wave's
                         // 32 words of one tile are ONE 128-byte line (row-major (B*H, S, S/32) made every tile 32 scattered 4-byte stores:
                         // 16.8 M partial-line write requests per C5 forward next to the 33.5 M of the scores)
    float* stats;      // (B*H, S, 2): the shift m2 (log2 units; row max of s*c1 minus at most 6) and 1 / sum_k exp2(s*c1 - m2)
    int S, H, nqb, ntile;
    // row strides (floats) of the projection-layout operands: a (B*S, H*dh) matrix of its own has ld = H*dh; Q, K, V (and dQ,
    // dK, dV) that are column blocks of ONE packed (B*S, 3*H*dh) projection output have ld = 3*H*dh
    int ldx;           // x1, x2
    int ldq;           // bq
    int ldc;           // ctx
    int ldo;           // out
    float scale, keep, dscale;
    float c1;          // scale * log2(e): exponents are taken in base 2
    unsigned keep_lt;  // a draw v is kept iff v < keep_lt = floor((1 - p) * 2^32)  (nk_common.h: the Bernoulli construction)
    unsigned long long seed, offset;
    int assign;        // backward: dQ = (1) or += (0)
    int SP;            // S rounded up to a multiple of 32: row count and row stride of the (B*H, SP, SP) scratch tensors (scores, dS, Pd,
                       // statistics, mask words).  SP == S except for ragged sequence lengths (read by the RAGGED instantiations only)
};

// Wave-private LDS round trip: every lane's ds_write is issued before any lane's ds_read (LDS is in-order per wave); the
// fences keep the compiler from reordering across the hand-over.
__device__ __forceinline__ void wave_lds_handover() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// registers (lane = query q, half h, 16 consecutive keys 16h..16h+15) -> 32 x 32 tile of a row-major (.., S) tensor, as
// 128-byte row segments (8 lanes x 16 B) instead of 64 scattered 16-byte pieces per store instruction.  Two halves, issued
// far apart: the tile goes to the wave's LDS scratch as soon as it is computed, and is read back and stored to HBM after
// the second MFMA product of the iteration - the ds_write -> ds_read round trip (a few hundred cycles, and there are only
// two or three waves per SIMD to hide it) then runs under 32 MFMAs instead of stalling the wave.
__device__ __forceinline__ void tile_write(float* scrw, const float (&v)[16], int lane) {
    const int q = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        *reinterpret_cast<float4*>(&scrw[q * SCR_LD + 16 * h + 4 * c]) = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}
__device__ __forceinline__ void tile_flush(const float* scrw, float* g /* &T[row0][kt*32] */, int S, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), c = 4 * (lane & 7);
        const float4 t = *reinterpret_cast<const float4*>(&scrw[r * SCR_LD + c]);
        nk_store_stream(reinterpret_cast<float4*>(g + (long long)r * S + c), t);
    }
}

// FULL: S is a multiple of 128, every wave of every block has queries.  Not a micro-optimisation: with a run-time `on`
// the loop body sits under a branch, the compiler's wait-count bookkeeping merges the two paths at the join and waits for
// ALL outstanding vector memory operations (the tile stores just issued included) before the staged tile may go to LDS;
// with the branch gone it waits for exactly the staging loads and the stores drain under the next tile's MFMAs.
// KEEP (forward only): write what the backward pass needs (scores, row statistics, dropout bits).  false = inference: O only,
// 2.1 GB of stores and the buffers themselves disappear.
// RAGGED: S is not a multiple of 32.  The scratch tensors are padded to SP = ceil32(S) rows and columns, so every tile access to
// them stays whole; what is left to guard is the caller's projection layout: key rows and query rows beyond S are CLAMPED to row
// S - 1 when loaded (a sample's rows are followed by the next sample's, or by the end of the buffer), padded keys get a score of
// -inf in the forward (probability exactly 0; the stored -inf makes the backward's recomputed probability, dS and Pd exactly 0 too),
// padded queries compute a copy of row S - 1 that is never stored outside the scratch.  The dropout draws of score (bh, r, k) are
// indexed in the PADDED tensor ((bh * SP + r) * SP + k), which keeps a lane's 16 keys on two whole Philox calls.
template <bool BWD, bool MASKED, bool FULL, int OCC, bool KEEP = true, int DH = 64, bool RAGGED = false>
__global__ __launch_bounds__(A_NT, OCC) void attention_kernel(const AttnArgs p) {
    static_assert(!(RAGGED && FULL), "ragged sequence lengths take the guarded instantiation");
    constexpr int X1_LD = x1_ld(DH), X2_LD = x2_ld(DH);
    constexpr int ND = DH / 32;   // output column tiles
    constexpr int NR = DH / 32;   // float4 per thread and operand of a staged key tile (32 keys x DH)
    constexpr int C4 = DH / 4;    // float4 per key row
    __shared__ __attribute__((aligned(16))) float x1s[2][32 * X1_LD];
    __shared__ __attribute__((aligned(16))) float x2s[2][32 * X2_LD];
    __shared__ __attribute__((aligned(16))) float scr[A_NT / 64][(BWD ? 2 : 1) * 32 * SCR_LD];  // per wave: tile scratch (backward: dS | Pd)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane & 31, h = lane >> 5;
    // blocks of one (sample, head) are consecutive in the sequence and one XCD takes a contiguous chunk of it: the 512 KB
    // of K and V a head's blocks share stay in that XCD's L2
    const int seq = nkmma::xcd_chunk(blockIdx.x, gridDim.x);
    const int bh = seq / p.nqb, qb = seq % p.nqb;
    const long long samp = (long long)(bh / p.H) * p.S, hoff = (long long)(bh % p.H) * DH;
    const long long flat0 = samp * p.ldx + hoff;   // (sample, head) origin in x1 / x2; bq, ctx and out have their own row strides
    const int q0 = qb * A_QB + w * 32;
    const bool on = FULL ? true : q0 < p.S;  // wave-uniform: the wave has at least one query
    const int SP = RAGGED ? p.SP : p.S;
    const float* x1 = p.x1 + flat0;
    const float* x2 = p.x2 + flat0;

    // ---- cooperative staging of one key tile (32 keys x DH of each operand): NR + NR float4 per thread ------------------
    unsigned goff[NR], goff_last[NR], s1[NR], s2[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int idx = tid + A_NT * r, key = idx / C4, c4 = idx % C4;
        goff[r] = (unsigned)(key * p.ldx + 4 * c4);
        goff_last[r] = RAGGED ? (unsigned)(min(key, p.S - 1 - 32 * (p.ntile - 1)) * p.ldx + 4 * c4) : goff[r];  // last key tile: rows clamped to S - 1
        // mfma row i supplies key 16*((i>>2)&1) + 4*(i>>3) + (i&3); its inverse places key 16a + 4b + c in row 8b + 4a + c
        s1[r] = (unsigned)((8 * ((key >> 2) & 3) + 4 * (key >> 4) + (key & 3)) * X1_LD + 4 * c4);
        s2[r] = (unsigned)(key * X2_LD + ((4 * c4 + 32 * (key >> 4)) & (DH - 1)));
    }
    // NAMED registers (up to four float4 per operand), not arrays: `float4 st[NR]` written under `if (more)` and read a loop
    // body later is kept in scratch memory by the compiler (measured: 64 bytes of private segment, four scratch stores and
    // loads per tile), and so is a small array of LDS pointers.
    float4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
#define A_STAGE_LOAD(KT)                                                                                         \
    do {                                                                                                         \
        const long long t0_ = (long long)(KT) * 32 * p.ldx;                                                      \
        const bool lt_ = RAGGED && (KT) == p.ntile - 1;   /* wave-uniform */                                     \
        const unsigned g0_ = lt_ ? goff_last[0] : goff[0], g1_ = lt_ ? goff_last[NR > 1 ? 1 : 0] : goff[NR > 1 ? 1 : 0],         \
                       g2_ = lt_ ? goff_last[NR > 2 ? 2 : 0] : goff[NR > 2 ? 2 : 0], g3_ = lt_ ? goff_last[NR > 2 ? 3 : 0] : goff[NR > 2 ? 3 : 0]; \
        sa0 = *reinterpret_cast<const float4*>(x1 + t0_ + g0_);                                                 \
        if constexpr (NR > 1) sa1 = *reinterpret_cast<const float4*>(x1 + t0_ + g1_);                           \
        if constexpr (NR > 2) { sa2 = *reinterpret_cast<const float4*>(x1 + t0_ + g2_);                         \
                                sa3 = *reinterpret_cast<const float4*>(x1 + t0_ + g3_); }                       \
        sb0 = *reinterpret_cast<const float4*>(x2 + t0_ + g0_);                                                 \
        if constexpr (NR > 1) sb1 = *reinterpret_cast<const float4*>(x2 + t0_ + g1_);                           \
        if constexpr (NR > 2) { sb2 = *reinterpret_cast<const float4*>(x2 + t0_ + g2_);                         \
                                sb3 = *reinterpret_cast<const float4*>(x2 + t0_ + g3_); }                       \
        (void)g1_; (void)g2_; (void)g3_;                                                                         \
    } while (0)
#define A_STAGE_STORE(BUF)                                                                                       \
    do {                                                                                                         \
        *reinterpret_cast<float4*>(&x1s[BUF][s1[0]]) = sa0;                                                      \
        if constexpr (NR > 1) *reinterpret_cast<float4*>(&x1s[BUF][s1[NR > 1 ? 1 : 0]]) = sa1;                   \
        if constexpr (NR > 2) { *reinterpret_cast<float4*>(&x1s[BUF][s1[NR > 2 ? 2 : 0]]) = sa2;                 \
                                *reinterpret_cast<float4*>(&x1s[BUF][s1[NR > 2 ? 3 : 0]]) = sa3; }               \
        *reinterpret_cast<float4*>(&x2s[BUF][s2[0]]) = sb0;                                                      \
        if constexpr (NR > 1) *reinterpret_cast<float4*>(&x2s[BUF][s2[NR > 1 ? 1 : 0]]) = sb1;                   \
        if constexpr (NR > 2) { *reinterpret_cast<float4*>(&x2s[BUF][s2[NR > 2 ? 2 : 0]]) = sb2;                 \
                                *reinterpret_cast<float4*>(&x2s[BUF][s2[NR > 2 ? 3 : 0]]) = sb3; }               \
    } while (0)

    // ---- per-query state ----------------------------------------------------------------------------------------
    const int qrow = on ? q0 + q : 0;                        // row of the scratch tensors (padded rows exist there)
    const int row = RAGGED ? min(qrow, p.S - 1) : qrow;      // row of the projection layout
    float4 bq[DH / 8];  // B operand of pass 1: element (query, dh = 8j + 4h + c)
    {
        const float* b = p.bq + samp * p.ldq + hoff + (long long)row * p.ldq + 4 * h;
#pragma unroll
        for (int j = 0; j < DH / 8; ++j) bq[j] = *reinterpret_cast<const float4*>(b + 8 * j);
    }
    float m_run = -1e30f, l_run = 0.f;  // forward: online softmax (shift, sum); backward: the stored shift and 1 / sum
    float dot = 0.f;
    if (BWD) {
        const float2 ms = *reinterpret_cast<const float2*>(p.stats + ((long long)bh * SP + qrow) * 2);
        m_run = ms.x; l_run = ms.y;
        const float* o = p.ctx + samp * p.ldc + hoff + (long long)row * p.ldc + 4 * h;
#pragma unroll
        for (int j = 0; j < DH / 8; ++j) {
            const float4 ov = *reinterpret_cast<const float4*>(o + 8 * j);
            dot += (bq[j].x * ov.x + bq[j].y * ov.y) + (bq[j].z * ov.z + bq[j].w * ov.w);
        }
        dot += __shfl_xor(dot, 32, 64);
        // sum_k dP_k P_k with dP = dPd * noise, while dO . O = sum_k dPd_k P_k noise_k / keep
        if (MASKED) dot *= p.keep;
    }
    f32x16 oacc[ND];  // out^T tiles: oacc[d] holds dh 32 d .. 32 d + 31 of this lane's query
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;

    const long long rowbase = ((long long)bh * SP + (on ? q0 : 0)) * SP;  // element index of (row q0, key 0) in the (B*H, SP, SP) tensors
    const uint2 key = make_uint2((unsigned)p.seed, (unsigned)(p.seed >> 32));
    const unsigned long long ctr0 = (unsigned long long)(rowbase + (long long)q * SP) / 8 + 2 * h + p.offset;  // 8 draws per call
    float* scrw = scr[w];
    float* scrb = scrw + (BWD ? 32 * SCR_LD : 0);
    // backward: the score tile of the NEXT iteration, in the coalesced load layout (lane -> rows 8i + lane/8, 16 B each)
    float4 sn0, sn1, sn2, sn3;
    unsigned mkn = 0;  // and this row's 32 dropout bits of that tile
    // this wave's mask words: [bh][query tile][kt][q]
    unsigned* const mwave = MASKED ? p.maskbits + (((long long)bh * (SP / 32) + (on ? q0 / 32 : 0)) * p.ntile) * 32 + q : nullptr;
    const unsigned* mload = mwave;
    const float* sload = p.scores + rowbase + (long long)(lane >> 3) * SP + 4 * (lane & 7);
#define A_SCORES_LOAD(KT)                                                                        \
    do {                                                                                         \
        sn0 = *reinterpret_cast<const float4*>(sload + (KT) * 32);                               \
        sn1 = *reinterpret_cast<const float4*>(sload + (long long)8 * SP + (KT) * 32);           \
        sn2 = *reinterpret_cast<const float4*>(sload + (long long)16 * SP + (KT) * 32);          \
        sn3 = *reinterpret_cast<const float4*>(sload + (long long)24 * SP + (KT) * 32);          \
        if (MASKED) mkn = mload[(KT) * 32];                                                      \
    } while (0)

#define A_SCORES_TO_LDS()                                                                        \
    do {                                                                                         \
        float* sw_ = &scrw[(lane >> 3) * SCR_LD + 4 * (lane & 7)];                               \
        *reinterpret_cast<float4*>(sw_) = sn0;                                                   \
        *reinterpret_cast<float4*>(sw_ + 8 * SCR_LD) = sn1;                                      \
        *reinterpret_cast<float4*>(sw_ + 16 * SCR_LD) = sn2;                                     \
        *reinterpret_cast<float4*>(sw_ + 24 * SCR_LD) = sn3;                                     \
    } while (0)
    A_STAGE_LOAD(0);
    if (BWD && on) A_SCORES_LOAD(0);
    A_STAGE_STORE(0);
    if (BWD && on) A_SCORES_TO_LDS();
    // every prologue load has landed: without this the compiler's wait-count bookkeeping carries the per-query operand's
    // loads into the loop and waits for the NEXT tile's staging loads in the middle of pass 1, every iteration
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    for (int kt = 0; kt < p.ntile; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < p.ntile;
        if (more) A_STAGE_LOAD(kt + 1);
        if (on) {
            float sv[16];
            unsigned mybits = 0;
            if (BWD) {  // score tile (put into the scratch at the end of the previous iteration) -> lane layout; fetch the next one
                if (MASKED) mybits = mkn >> (16 * h);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 t = *reinterpret_cast<const float4*>(&scrw[q * SCR_LD + 16 * h + 4 * c]);
                    sv[4 * c] = t.x; sv[4 * c + 1] = t.y; sv[4 * c + 2] = t.z; sv[4 * c + 3] = t.w;
                }
                if (more) A_SCORES_LOAD(kt + 1);
            }
            // ---- pass 1: C[key][query] = X1 . Bq^T  (forward: scores; backward: dPd) -------------------------------
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            {
                const float* a1 = &x1s[cur][q * X1_LD + 4 * h];
#pragma unroll
                for (int j = 0; j < DH / 8; ++j) {
                    const float4 a = *reinterpret_cast<const float4*>(a1 + 8 * j);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[j].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[j].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[j].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[j].w, acc, 0, 0, 0);
                }
            }
            // Bernoulli(1 - p) draws of this lane's 16 keys.  Forward: 2 Philox calls (8 consecutive keys each, the draw
            // layout of nk_common.h / nk_dropout_fwd / nk_scale_softmax_dropout_fwd), packed to one bit per score for the
            // backward pass - the Philox rounds were ~2000 of the ~6500 issue cycles of a masked tile with one word per
            // score (v_mad_u64_u32 is quarter rate) and are paid once, not twice; the backward mask is the forward's by
            // construction (the reference shares the noise buffer the same way, node/dropout/mod.rs:113-128).
            bool kp[16];          // forward: the compare results stay lane masks in SGPR pairs (the masked forward is at its VGPR limit)
            int km[16];           // backward: 0 / -1 per key as AND operands (one v_bfe_i32 + one v_and per use instead of bit test + compare + select)
            if (MASKED && !BWD) {
                unsigned bits = 0;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const unsigned long long ctr = ctr0 + (unsigned long long)(kt * 4 + c);
                    const uint4 r = philox4x32_10(make_uint4((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u), key);
                    const unsigned wv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool k0 = wv[k] < p.keep_lt, k1 = nk_rot16(wv[k]) < p.keep_lt;
                        kp[8 * c + 2 * k] = k0;
                        kp[8 * c + 2 * k + 1] = k1;
                        bits |= (k0 ? 1u : 0u) << (8 * c + 2 * k);
                        bits |= (k1 ? 1u : 0u) << (8 * c + 2 * k + 1);
                    }
                }
                const unsigned other = (unsigned)__shfl_xor((int)bits, 32, 64);
                if (KEEP && h == 0) mwave[kt * 32] = bits | (other << 16);
            }
            if (MASKED && BWD) {
#pragma unroll
                for (int e = 0; e < 16; ++e) km[e] = __builtin_amdgcn_sbfe((int)mybits, e, 1);   // v_bfe_i32: 0 / -1
            }
            float bv[16];  // B operand of pass 2
            if (!BWD) {
                float raw[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) raw[e] = acc[e];
                if (RAGGED && kt == p.ntile - 1) {  // keys beyond S: probability exactly 0, here and (through the stored score) in the backward
                    const int nvalid = p.S - 32 * kt - 16 * h;
#pragma unroll
                    for (int e = 0; e < 16; ++e) raw[e] = e < nvalid ? raw[e] : -INFINITY;
                }
                if (KEEP) tile_write(scrw, raw, lane);
                // Online softmax in the base-2 exponent domain: exp(s*scale - m) = exp2(s*c1 - m2), c1 = scale*log2(e), one fma
                // and one v_exp_f32 per element.  f32 MFMA and VALU instructions do NOT overlap on a SIMD (measured,
                // benchmarks/native/mfma_valu_overlap.hip: both run on the f32 lanes), so every VALU instruction here is
                // paid in full on top of the 64 MFMAs of the tile.
                float rmax = raw[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) rmax = fmaxf(rmax, raw[e]);   // scale > 0: max of the scaled = scaled max
                rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
                const float tm2 = rmax * p.c1;
                // The running max only has to bound the exponents, not equal the true max: it moves when a tile exceeds it by
                // more than 2^6 (terms stay <= 64, sums <= 2^16), i.e. after the first tile practically never, and the rescale
                // of the 32 accumulator registers is skipped (wave-uniform test).  Softmax is invariant to the shift, and the
                // backward pass recomputes the probabilities with the stored (shift, 1 / sum) pair.
                if (__any(tm2 > m_run + 6.f)) {
                    const float m_new = fmaxf(m_run, tm2);
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    l_run *= alpha;
                    m_run = m_new;
#pragma unroll
                    for (int d = 0; d < ND; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
                }
                float ps = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { sv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(raw[e], p.c1, -m_run)); ps += sv[e]; }
                ps += __shfl_xor(ps, 32, 64);
                l_run += ps;
                // Dropout: the 1 / (1 - p) factor is applied once, with the normalisation, in the epilogue
#pragma unroll
                for (int e = 0; e < 16; ++e) bv[e] = MASKED ? (kp[e] ? sv[e] : 0.f) : sv[e];
            } else {
                float pd[16];
                const float inv_s = l_run * p.scale;                      // P * scale = e * (1/sum * scale)
                const float inv_d = MASKED ? l_run * p.dscale : l_run;    // Pd = e * (1/sum * 1/(1-p)) where kept
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float ev = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[e], p.c1, -m_run));  // exp(s*scale - shift), as the forward
                    const float gv = MASKED ? __int_as_float(__float_as_int(acc[e]) & km[e]) : acc[e];  // DropoutBackward: g * noise
                    bv[e] = (ev * inv_s) * (gv - dot);                           // SoftmaxBackward, MultiplicationBackwardLeft
                    pd[e] = MASKED ? __int_as_float(__float_as_int(ev * inv_d) & km[e]) : ev * inv_d;    // Dropout forward (for dV = Pd^T . dO)
                }
                tile_write(scrw, bv, lane);   // (the score tile was read out of this region at the top of the iteration)
                tile_write(scrb, pd, lane);
            }
            // ---- pass 2: out^T[dh][query] += X2^T . C ----------------------------------------------------------------
            {
                // column tile d of the output reads dh column 32 d + q of the key rows 16 h + e, stored rotated by 32 h
                const float* const a2b = &x2s[cur][(16 * h) * X2_LD];
                const int c0 = (q + 32 * h) & (DH - 1), c1 = (q + 32 + 32 * h) & (DH - 1), c2 = (q + 64 + 32 * h) & (DH - 1),
                          c3 = (q + 96 + 32 * h) & (DH - 1);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2b[e * X2_LD + c0], bv[e], oacc[0], 0, 0, 0);
                    if constexpr (ND > 1) oacc[ND > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2b[e * X2_LD + c1], bv[e], oacc[ND > 1 ? 1 : 0], 0, 0, 0);
                    if constexpr (ND > 2) {
                        oacc[ND > 2 ? 2 : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2b[e * X2_LD + c2], bv[e], oacc[ND > 2 ? 2 : 0], 0, 0, 0);
                        oacc[ND > 2 ? 3 : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2b[e * X2_LD + c3], bv[e], oacc[ND > 2 ? 3 : 0], 0, 0, 0);
                    }
                }
                (void)c1; (void)c2; (void)c3;
            }
            // ---- the tiles written to the scratch before pass 2 go to HBM now (machine scheduler pinned: hoisting the
            //      ds_reads above the MFMAs would put the LDS round trip back on the critical path) ---------------------
            __builtin_amdgcn_sched_barrier(0);
            wave_lds_handover();
            if (!BWD) {
                if (KEEP) tile_flush(scrw, p.scores + rowbase + kt * 32, SP, lane);
            } else {
                tile_flush(scrw, p.ds + rowbase + kt * 32, SP, lane);
                tile_flush(scrb, p.dropped + rowbase + kt * 32, SP, lane);
                if (more) { wave_lds_handover(); A_SCORES_TO_LDS(); }   // next tile's scores (loaded during this iteration)
            }
        }
        if (more) A_STAGE_STORE(cur ^ 1);
        __syncthreads();
    }
#undef A_STAGE_LOAD
#undef A_STAGE_STORE
#undef A_SCORES_LOAD
#undef A_SCORES_TO_LDS
    if (!on) return;
    // ---- epilogue: lane (query q, half h) owns dh = 32 d + 8 c + 4 h + {0..3} ----------------------------------------
    float* orow = p.out + samp * p.ldo + hoff + (long long)row * p.ldo + 4 * h;
    const bool qvalid = !RAGGED || qrow < p.S;  // a padded query's copy of row S - 1 stays in the scratch
    if (!BWD) {
        const float inv = 1.f / l_run;
        const float io = MASKED ? inv * p.dscale : inv;
        if (qvalid) {
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *reinterpret_cast<float4*>(orow + 32 * d + 8 * c) =
                        make_float4(oacc[d][4 * c] * io, oacc[d][4 * c + 1] * io, oacc[d][4 * c + 2] * io, oacc[d][4 * c + 3] * io);
        }
        if (KEEP && h == 0) *reinterpret_cast<float2*>(p.stats + ((long long)bh * SP + qrow) * 2) = make_float2(m_run, inv);
    } else if (qvalid) {
        // every old value is loaded before the first store (a store may alias the next load: a one-walk `+=` serialises)
        float4 old[ND][4];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                old[d][c] = p.assign ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(orow + 32 * d + 8 * c);
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<float4*>(orow + 32 * d + 8 * c) =
                    make_float4(old[d][c].x + oacc[d][4 * c], old[d][c].y + oacc[d][4 * c + 1], old[d][c].z + oacc[d][4 * c + 2],
                                old[d][c].w + oacc[d][4 * c + 3]);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int attention_check(int B, int S, int H, int dh, double p, int train, float scale) {
    NK_CHECK(p >= 0.0 && p <= 1.0, "Wrong probability received: %g.", p);
    NK_CHECK(scale > 0.f && scale < 1e30f, "fused attention needs a positive finite scale (the row max is taken before scaling), got %g", (double)scale);
    NK_CHECK(B > 0 && S > 0 && H > 0, "attention: non-positive geometry");
    NK_CHECK(nk_attention_supported(S, dh, p, train), "fused attention needs dh in {32, 64, 128} and p < 1 in training (S=%d dh=%d p=%g)", S, dh, p);
    NK_CHECK((long long)B * H * ((S + 31) / 32) * ((S + 31) / 32) < (1ll << 31) / 32, "attention: the mask words exceed 2^31");
    NK_CHECK((long long)B * S * H * dh < (1ll << 31), "attention: the projection layout exceeds 2^31 elements");
    return NK_OK;
}

template <bool BWD, int DH>
int attention_launch_dh(nk_device* dev, AttnArgs& a, int B, int S, int H, double p, int train, uint64_t seed, uint64_t offset, float scale) {
    a.S = S; a.H = H; a.nqb = (S + A_QB - 1) / A_QB; a.SP = (S + 31) / 32 * 32; a.ntile = a.SP / 32;
    a.scale = scale; a.c1 = scale * 1.44269504088896341f; a.keep = (float)(1.0 - p); a.dscale = 1.f / (1.f - (float)p);  // as nk_scale_softmax_dropout_fwd
    a.seed = seed; a.offset = offset;
    a.keep_lt = nk_keep_threshold(1.0 - p);   // Bernoulli::new(1. - p), node/dropout/mod.rs:46
    const bool masked = train && p != 0.0;
    if (!BWD && masked)   // (the backward reads the forward's stored bits: nothing of its own to freeze)
        if (int rc = nk_refuse_capture(dev, "nk_attention_fwd: the Philox offset (every replay would draw the same mask)",
                                       "run the training-mode dropout eagerly, or capture the evaluation graph")) return rc;
    const dim3 grid((unsigned)(B * H * a.nqb)), block(A_NT);
    const bool full = S % A_QB == 0, ragged = S % 32 != 0;
    // OCC = blocks per CU the register budget is sized for.  DH = 64: three forward blocks fit by LDS (52 KB each) at <= 168
    // VGPRs - the masked forward then spills a few registers and is still faster than two blocks without spills (1.36 vs
    // 1.37 - 1.41 ms at C5, round 3) - and the backward holds 70 KB of LDS per block: two blocks per CU.  DH = 128: 85 / 103 KB
    // of LDS and ~240 / ~290 registers per lane: one block per CU (a wave may then use the whole unified register file).
    // DH = 32: the DH = 64 budgets.
    constexpr int OCC_F = DH == 128 ? 1 : 3, OCC_B = DH == 128 ? 1 : 2;   // forward / backward blocks per CU
    constexpr int OCC_DEFAULT = BWD ? OCC_B : OCC_F;
    const bool occ2 = !BWD && DH != 128 && dev->tune_attn_occ == 2;   // (nk_dev_tune, NK_TUNE_ATTENTION_OCC: sweeps only)
#define NK_ATT(M, F, R)                                                                                                    \
    do {                                                                                                                   \
        /* inference forward: KEEP = false (spelled `BWD`, false on the only path that reaches this line) */             \
        if (!BWD && !a.scores) hipLaunchKernelGGL((attention_kernel<BWD, M, F, OCC_DEFAULT, BWD, DH, R>), grid, block, 0, dev->compute, a); \
        else if (occ2) hipLaunchKernelGGL((attention_kernel<BWD, M, F, (DH == 128 ? 1 : 2), true, DH, R>), grid, block, 0, dev->compute, a); \
        else hipLaunchKernelGGL((attention_kernel<BWD, M, F, OCC_DEFAULT, true, DH, R>), grid, block, 0, dev->compute, a);    \
    } while (0)
    if (ragged) { if (masked) NK_ATT(true, false, true); else NK_ATT(false, false, true); }
    else if (masked && full) NK_ATT(true, true, false);
    else if (masked) NK_ATT(true, false, false);
    else if (full) NK_ATT(false, true, false);
    else NK_ATT(false, false, false);
#undef NK_ATT
    NK_LAUNCH_CHECK();
    return NK_OK;
}

template <bool BWD>
int attention_launch(nk_device* dev, AttnArgs& a, int B, int S, int H, int dh, double p, int train, uint64_t seed, uint64_t offset, float scale) {
    if (dh == 32) return attention_launch_dh<BWD, 32>(dev, a, B, S, H, p, train, seed, offset, scale);
    if (dh == 128) return attention_launch_dh<BWD, 128>(dev, a, B, S, H, p, train, seed, offset, scale);
    return attention_launch_dh<BWD, 64>(dev, a, B, S, H, p, train, seed, offset, scale);
}

}  // namespace

extern "C" {

int nk_attention_supported(int S, int dh, double p, int train) {
    return (dh == 32 || dh == 64 || dh == 128) && S > 0 && !(train && 1.0 - p == 0.0);
}

static int attention_fwd_impl(nk_device* dev, const float* Q, const float* K, const float* V, int ld_qkv, float* scores, float* stats,
                              uint32_t* mask_bits, float* O, int B, int S, int H, int dh, float scale, double p, int train, uint64_t seed,
                              uint64_t offset) {
    NK_USE(dev);
    if (int rc = attention_check(B, S, H, dh, p, train, scale)) return rc;
    NK_CHECK(Q && K && V && O, "null pointer in nk_attention_fwd");
    NK_CHECK((scores != nullptr) == (stats != nullptr), "nk_attention_fwd: scores and stats are kept together or not at all");
    NK_CHECK(!scores || mask_bits || !(train && p != 0.0), "nk_attention_fwd: dropout is active, the mask_bits buffer is needed");
    NK_CHECK(al16(Q) && al16(K) && al16(V) && al16(scores) && al16(O) && al16(stats), "nk_attention_fwd needs 16-byte aligned buffers");
    NK_CHECK((long long)B * S * ld_qkv < (1ll << 31), "attention: the packed projection layout exceeds 2^31 elements");
    nk_prof_start(dev, NK_KERNEL_ATTENTION, 4.0 * B * H * (double)S * S * dh);
    AttnArgs a{};
    a.x1 = K; a.x2 = V; a.bq = Q; a.out = O; a.scores = scores; a.stats = stats; a.maskbits = mask_bits;
    a.ldx = ld_qkv; a.ldq = ld_qkv; a.ldc = H * dh; a.ldo = H * dh;
    const int rc = attention_launch<false>(dev, a, B, S, H, dh, p, train, seed, offset, scale);
    nk_prof_stop(dev);
    return rc;
}
int nk_attention_fwd(nk_device* dev, const float* Q, const float* K, const float* V, float* scores, float* stats,
                     uint32_t* mask_bits, float* O, int B, int S, int H, int dh, float scale, double p, int train, uint64_t seed,
                     uint64_t offset) {
    return attention_fwd_impl(dev, Q, K, V, H * dh, scores, stats, mask_bits, O, B, S, H, dh, scale, p, train, seed, offset);
}
int nk_attention_qkv_fwd(nk_device* dev, const float* QKV, float* scores, float* stats, uint32_t* mask_bits, float* O, int B, int S,
                         int H, int dh, float scale, double p, int train, uint64_t seed, uint64_t offset) {
    NK_CHECK(QKV != nullptr, "null pointer in nk_attention_qkv_fwd");
    const int d = H * dh;
    return attention_fwd_impl(dev, QKV, QKV + d, QKV + 2 * d, 3 * d, scores, stats, mask_bits, O, B, S, H, dh, scale, p, train, seed, offset);
}

static int attention_bwd_impl(nk_device* dev, float* dQ, float* dK, float* dV, float* dS, float* dropped, const float* dO, const float* O,
                              const float* scores, const float* stats, const uint32_t* mask_bits, const float* Q, const float* K,
                              const float* V, int ld_qkv, int B, int S, int H, int dh, float scale, double p, int train, int assign_dq,
                              int assign_dk, int assign_dv) {
    NK_USE(dev);
    if (int rc = attention_check(B, S, H, dh, p, train, scale)) return rc;
    NK_CHECK(dQ && dK && dV && dS && dropped && dO && O && scores && stats && Q && K && V, "null pointer in nk_attention_bwd");
    NK_CHECK(mask_bits || !(train && p != 0.0), "nk_attention_bwd: dropout is active, the forward's mask_bits are needed");
    NK_CHECK(al16(dQ) && al16(dK) && al16(dV) && al16(dS) && al16(dropped) && al16(dO) && al16(O) && al16(scores) && al16(stats) &&
                 al16(Q) && al16(K) && al16(V),
             "nk_attention_bwd needs 16-byte aligned buffers");
    NK_CHECK((long long)B * S * ld_qkv < (1ll << 31), "attention: the packed projection layout exceeds 2^31 elements");
    nk_prof_start(dev, NK_KERNEL_ATTENTION, 4.0 * B * H * (double)S * S * dh);
    AttnArgs a{};
    a.x1 = V; a.x2 = K; a.bq = dO; a.ctx = O; a.out = dQ; a.scores = const_cast<float*>(scores); a.ds = dS; a.dropped = dropped;
    a.stats = const_cast<float*>(stats); a.maskbits = const_cast<uint32_t*>(mask_bits); a.assign = assign_dq ? 1 : 0;
    a.ldx = ld_qkv; a.ldq = H * dh; a.ldc = H * dh; a.ldo = ld_qkv;
    int rc = attention_launch<true>(dev, a, B, S, H, dh, p, train, 0, 0, scale);
    nk_prof_stop(dev);
    if (rc) return rc;
    // dK_bh (+)= dS_bh^T . Q_bh and dV_bh (+)= Pd_bh^T . dO_bh: reductions over the queries, i.e. across the blocks above.
    // (Keeping dS / Pd in the kernels' own tile order - no LDS transposition in the kernel, a k-contiguous A operand whose
    // 128 x 32 tiles are single 16 KB runs for these products - was built and measured: kernel and products unchanged.)
    const int d = H * dh, SP = (S + 31) / 32 * 32;  // row stride of the scratch tensors (== S unless S is ragged)
    const long long so = (long long)S * d, sq = (long long)S * ld_qkv, po = (long long)H * SP * SP, pi = (long long)SP * SP;
    // (one launch for both when nk_sgemm_pair's rule says so: the two grids share their last wave of resident blocks)
    return nk_sgemm_pair_batched(dev, B, H, 1, 0, S, dh, S, dS, SP, po, pi, Q, ld_qkv, sq, dh, assign_dk ? 0.f : 1.f, dK, ld_qkv, sq, dh,
                                 1, 0, S, dh, S, dropped, SP, po, pi, dO, d, so, dh, assign_dv ? 0.f : 1.f, dV, ld_qkv, sq, dh);
}
int nk_attention_bwd(nk_device* dev, float* dQ, float* dK, float* dV, float* dS, float* dropped, const float* dO, const float* O,
                     const float* scores, const float* stats, const uint32_t* mask_bits, const float* Q, const float* K,
                     const float* V, int B, int S, int H, int dh, float scale, double p, int train, int assign_dq, int assign_dk,
                     int assign_dv) {
    return attention_bwd_impl(dev, dQ, dK, dV, dS, dropped, dO, O, scores, stats, mask_bits, Q, K, V, H * dh, B, S, H, dh, scale, p, train,
                              assign_dq, assign_dk, assign_dv);
}
int nk_attention_qkv_bwd(nk_device* dev, float* dQKV, float* dS, float* dropped, const float* dO, const float* O, const float* scores,
                         const float* stats, const uint32_t* mask_bits, const float* QKV, int B, int S, int H, int dh, float scale,
                         double p, int train, int assign) {
    NK_CHECK(dQKV && QKV, "null pointer in nk_attention_qkv_bwd");
    const int d = H * dh;
    return attention_bwd_impl(dev, dQKV, dQKV + d, dQKV + 2 * d, dS, dropped, dO, O, scores, stats, mask_bits, QKV, QKV + d, QKV + 2 * d, 3 * d,
                              B, S, H, dh, scale, p, train, assign, assign, assign);
}

}  // extern "C"
