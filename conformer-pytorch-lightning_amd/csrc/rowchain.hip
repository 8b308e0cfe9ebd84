// rowchain.hip -- row-local chains of the conformer block in ONE launch per chain (gfx950).
//
// Everything in a conformer block except attention (mixes frames of an utterance) and the depthwise conv (15-frame
// halo) is ROW-LOCAL: it maps one frame's D-vector to another.  At config 2 there are ~31 rows per CU, so a separate
// launch per linear layer is dominated by its prologue/epilogue.  This kernel runs a whole row-local chain on a
// 32-row tile that never leaves the CU:
//
//     [HEAD]  x  = residual + mask( A16 . Wh^T + bh )          A16 = 16-bit tile from the previous kernel
//      else   x  = rows of the f32 residual stream
//             xn = LN(x; ln)  (optionally zeroing padded rows)                      -> LDS, 16 bit
//     [MID]   y  = x + alpha * ( W2 . act( W1 . xn + b1 ) + b2 )                    (feed-forward module)
//      else   y  = x
//             y1 = LN1(y) -> out_f32 ;   y2 = LN2(y1) -> LDS / out16
//     [TAIL]  t  = y2 . Wt^T + bt   (optionally GLU: a * sigmoid(g))               -> tail_out, 16 bit
//
// used three times per block (encoder_layer.py:56-70):
//     macaron  : MID + TAIL           LN_ffm -> FFN_m -> +res -> LN_mha -> fused QKV projection
//     conv-in  : HEAD + TAIL(GLU)     out-proj + res -> LN_conv (pad mask) -> pointwise-conv-1 + GLU
//     final    : HEAD + MID           pointwise-conv-2 + mask + res -> LN_ff -> FFN -> +res -> LN_final
// so a block is 6 launches: macaron, pos-projection, attention, conv-in, depthwise, final.
//
// Design (measured with scripts/probe_chain.hip and scripts/ubench.hip on MI355X):
//  * weights are 16-bit FRAGMENT-MAJOR: one wavefront-load (lane * 16 B, 1 KB) is one MFMA A fragment; they stream
//    L2 -> VGPR through a register ring, every ring slot a compile-time constant (all loops unrolled, no branches on the
//    load path), so the compiler's waits are exact vmcnt values;
//  * swapped MFMA roles: weights are the A operand, activations (read from an LDS tile by ds_read_b128) the B operand, so
//    a lane owns 4 consecutive output columns of one row -- vector bias / residual / stores;
//  * 16 wavefronts per workgroup (4 per SIMD, <= 128 VGPRs each).  A wavefront pays ~64 clk to ISSUE a 1 KB load and issues
//    in order, so its loads serialise with its own MFMAs; per-CU load throughput scales with the number of wavefronts issuing
//    (57 B/clk/CU at 4, 109 at 8, 155 at 16).  Earlier versions of this kernel split the FFN's K over 4 or 8 wavefronts with
//    the hidden activation fed from accumulators straight into the second product: that needs the whole 32 x D output
//    (128 VGPRs) per wavefront, which caps the workgroup at 8 wavefronts and a 12-deep ring -- 46 k cycles per FFN against
//    a 16 k MFMA floor.  Here the FFN runs in two phases around a 32 x FF 16-bit HIDDEN TILE IN LDS (128 KB of the 160):
//        phase 1  hidden = act(xn . W1^T + b1)   each wavefront owns distinct hidden columns (pairs of 16-column fragments)
//        phase 2  y      = hidden . W2^T         each wavefront owns two 16-column output fragments and half of K
//    so accumulators are 16 VGPRs, every weight fragment is still loaded exactly once per workgroup, and there is no
//    cross-wavefront reduction tree (the two K halves meet in the f32 output tile, fixed order);
//  * LDS tiles are padded so that the row step is 8 words mod 64: the ds_read_b128 lane groups are then conflict-free
//    (a 4-word step, the "+8 halfs" habit, makes every fragment read 2-way conflicted);
//  * everything a phase needs from global memory is requested a phase early (row data, norm parameters, epilogue
//    operands, the next phase's first weights).
#include <stdlib.h>
#include <string>
#include <type_traits>

#include "cfm_common.h"

struct ChainArgs {
    const float* x;           // f32 rows (no HEAD)
    const u16* head_a;        // 16-bit [M,D] (HEAD)
    const u16* head_w;        // fragment-major [D/16][KS][64][8]
    const float* head_b;
    const float* head_res;    // f32 [M,D]
    const uint8_t* head_mask; // zero the head OUTPUT row where 0 (before the residual add)
    const float *dw_w, *dw_b, *dw_scale, *dw_shift;   // HDW: depthwise conv [D,15] + bias + folded BatchNorm applied to head_a first
    int dw_T;                 //      frames per utterance (rows are [B, T] flattened; the conv zero-pads at utterance edges)
    const float *ln_g, *ln_b;
    const uint8_t* ln_mask;   // zero the normalised row where 0
    const u16 *w1f, *w2n;     // fragment-major W1 [FF/16][KS1][64][8] and W2 [D/16][FF/32][64][8]
    const float *b1, *b2;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float* out_f32;
    void* out16;
    float* out2_f32;          // optional f32 copy of y2 = LN2(y1) (the encoder's after_norm on top of the last block's norm_final)
    const u16* tail_w;        // fragment-major [tail_N/16][KS][64][8]
    const float* tail_b;
    void* tail_out;           // 16-bit [M, tail_N] (GLU: [M, tail_N/2])
    int64_t M;
    int tail_N;
    int out16_dtype;
    float alpha, eps;
    // TVT: the value columns [2D, 3D) of the (fused QKV) tail go out transposed per head, vt[((b*H + h)*64 + d) * vt_ld + t], instead of as rows
    u16* tail_vt;
    int vt_T, vt_ld;
    // HATT: the head input is the attention context of this tile's 32 frames, computed here (see the attention stage in the kernel)
    const u16* att_qkv;       // [B*T, 3D]: q | k | (values are read from att_vt)
    const u16* att_vt;        // [B, H, 64, att_vt_ld] transposed values, key columns >= T finite (zero-filled once)
    const u16* att_p;         // projected positional row per batch item (stride att_p_sb; 0: one shared row), or null (plain MHSA)
    const float *att_bias_u, *att_bias_v;
    const uint8_t* att_mask;  // key validity [B, >= T] bytes (stride att_m_sb), or null
    int64_t att_p_sb, att_m_sb;
    int att_T, att_vt_ld;
    float att_scale;
    // SEG2: a SECOND feed-forward segment on the same rows, in registers: after LN1 of the first (out_f32 optional then), LayerNorm
    // (s2_ln) -> feed-forward (s2_*) -> + residual -> s2_out_f32; the second LayerNorm (ln2) and the tail then follow THIS segment.
    const float *s2_ln_g, *s2_ln_b;
    const u16 *s2_w1f, *s2_w2n;
    const float *s2_b1, *s2_b2;
    float* s2_out_f32;
    float s2_alpha;
    // FSPLIT (D = 512): this launch computes ONE half of the feed-forward per workgroup and leaves partial sums psum_out[half][M][D]; the reduce rides in
    // the next launch's row load: rows = x + psum_alpha * (psum_in[0] + psum_in[1] + psum_b2)
    float* psum_out;
    const float *psum_in, *psum_b2;
    float psum_alpha;
    // CIN: the conv-in chain of the SAME block as the input stage of the depthwise stage (out-proj + residual -> LN_conv -> pointwise-conv-1 + GLU), computed
    // for the tile's 32 + 14 halo rows; its GLU rows go to the halo tile in LDS (head_a is not read), its residual rows to cin_out (which the head then reads
    // as head_res: the launcher sets head_res = cin_out)
    const u16* cin_a;         // attention context [M, D], 16 bit
    const u16* cin_w;         // out-proj, fragment-major [D/16][KS][64][8]
    const float* cin_b;
    const float* cin_res;     // f32 [M, D]: the residual stream in front of the out-projection (halo rows are read from OTHER tiles: never written in this launch)
    float* cin_out;           // f32 [M, D]: residual + out-proj, this tile's own rows
    const float *cin_ln_g, *cin_ln_b;
    const uint8_t* cin_mask;  // pad validity per row (zeroes the normalised row), or null
    const u16* cin_tw;        // pointwise-conv-1, GLU-interleaved, fragment-major [2D/16][KS][64][8]
    const float* cin_tb;
};

// Phase stamps for scripts/probe_chain.hip (built with -DCFM_CHAIN_STAMPS; never defined in the product build): thread 0 of each
// workgroup records the shader clock at every phase boundary.
#ifdef CFM_CHAIN_STAMPS
__device__ long long cfm_chain_stamps[1024 * 16];
#define CFM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) { cfm_chain_stamps[blockIdx.x * 16 + (i)] = clock64(); \
        if ((i) == 0 || (i) == 7) cfm_chain_stamps[blockIdx.x * 16 + 8 + ((i) == 7)] = wall_clock64(); } } while (0)   /* [8], [9]: 100 MHz wall clock */
#else
#define CFM_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int RBM = 32;                                    // rows per workgroup
constexpr int NW = 16;                                     // wavefronts per workgroup
constexpr int NT = 64 * NW;
constexpr int RPW = RBM / NW;                              // rows per wavefront in the row-wise (LayerNorm) phases
constexpr int MF = RBM / 16;                               // 16-row MFMA fragments per tile

// One "linear step": NFR 16-column output fragments of this wavefront over the whole K (KS 32-wide slices) of an LDS tile.
// With `refill` the weights of the NEXT step (fragments nx[], clamped by the caller) replace the current ones in the ring as
// they are consumed; the last step of a phase passes false (a compile-time constant after unrolling) and issues no loads.
template <typename HT, int KS, int NFR, int STRIDE, int MFX = MF>
__device__ __forceinline__ void linear_step(const u16* tile, const u32x4* wp, bool refill, const int (&nx)[NFR], u32x4 (&wr)[NFR * KS],
                                            f32x4 (&acc)[MFX][NFR], int g, int l15) {
    auto frag = [&](int mf, int kk) { return *(const u32x4*)(tile + (mf * 16 + l15) * STRIDE + kk * 32 + 8 * g); };
    u32x4 xf[2][MFX];                                       // activation fragments: this kk and the next (one-ahead LDS reads)
#pragma unroll
    for (int mf = 0; mf < MFX; ++mf) {
        xf[0][mf] = frag(mf, 0);
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        if (kk + 1 < KS) {
#pragma unroll
            for (int mf = 0; mf < MFX; ++mf) xf[(kk + 1) & 1][mf] = frag(mf, kk + 1);
        }
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
            for (int mf = 0; mf < MFX; ++mf) acc[mf][nf] = HT::mfma(wr[nf * KS + kk], xf[kk & 1][mf], acc[mf][nf]);
        if (refill) {
#pragma unroll
            for (int nf = 0; nf < NFR; ++nf) wr[nf * KS + kk] = wp[((int64_t)nx[nf] * KS + kk) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same step when the whole K of a step does not fit in registers (D = 512: KS = 16): the steps' fragments are one stream of positions
// pos = (s * KS + kk) * NFR + nf through a ring of RG registers (slot = pos % RG, every slot a compile-time constant after unrolling: `s` is
// a constant at every call site); the slot just consumed is refilled with the fragment RG positions ahead (ptr_of(pos), clamped by the caller).
template <typename HT, int KS, int NFR, int RG, int STRIDE, typename PtrOf, int MFX = MF>
__device__ __forceinline__ void ring_step(const u16* tile, const int s, const int npos, PtrOf ptr_of, u32x4 (&rg)[RG], f32x4 (&acc)[MFX][NFR], int g, int l15) {
    auto frag = [&](int mf, int kk) { return *(const u32x4*)(tile + (mf * 16 + l15) * STRIDE + kk * 32 + 8 * g); };
    u32x4 xf[2][MFX];
#pragma unroll
    for (int mf = 0; mf < MFX; ++mf) {
        xf[0][mf] = frag(mf, 0);
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        if (kk + 1 < KS) {
#pragma unroll
            for (int mf = 0; mf < MFX; ++mf) xf[(kk + 1) & 1][mf] = frag(mf, kk + 1);
        }
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) {
            const int pos = (s * KS + kk) * NFR + nf;
#pragma unroll
            for (int mf = 0; mf < MFX; ++mf) acc[mf][nf] = HT::mfma(rg[pos % RG], xf[kk & 1][mf], acc[mf][nf]);
        }
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) {
            const int pos = (s * KS + kk) * NFR + nf;
            if (pos + RG < npos) rg[pos % RG] = *ptr_of(pos + RG);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// LayerNorm of ROWS rows held one row per wavefront pass (lane owns columns (lane + 64 it) * 4 ..+3), in place.
template <int ROWS, int VPL, int D>
__device__ __forceinline__ void rows_layernorm(f32x4 (&v)[ROWS][VPL], const f32x4 (&gam)[VPL], const f32x4 (&bet)[VPL], float eps, int lane) {
    // No FMA contraction here: under the default (-ffp-contract=fast) the optimiser decides per instance which multiplies and adds to fuse, and instances that
    // must agree bit for bit (a chain with and without the depthwise / conv-in input stage, chained and unchained blocks) then round differently.
#pragma clang fp contract(off)
    float mean[ROWS], rstd[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < VPL; ++it)
            if ((lane + 64 * it) * 4 < D) s += (v[rr][it].x + v[rr][it].y) + (v[rr][it].z + v[rr][it].w);
        mean[rr] = wave_sum(s) * (1.0f / (float)D);
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < VPL; ++it)
            if ((lane + 64 * it) * 4 < D) {
                const f32x4 d = v[rr][it] - mean[rr];
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        rstd[rr] = __builtin_amdgcn_rsqf(wave_sum(q) * (1.0f / (float)D) + eps);   // v_rsq_f32 (1 ulp) instead of the IEEE sqrt + divide sequences
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
        for (int it = 0; it < VPL; ++it)
            if ((lane + 64 * it) * 4 < D) v[rr][it] = (v[rr][it] - mean[rr]) * rstd[rr] * gam[it] + bet[it];
}

template <typename HT, int D, int FF, int HSTEPS, bool HDW, bool MID, int TSTEPS, bool TGLU, bool HATT = false, bool TVT = false, bool SEG2 = false, bool FSPLIT = false, bool TSPLIT = false, bool CIN = false>
__global__ __launch_bounds__(NT) void cfm_rowchain_kernel(const ChainArgs a) {
    constexpr bool HEAD = HSTEPS > 0, TAIL = TSTEPS > 0;
    constexpr int NSEG = SEG2 ? 2 : 1;
    static_assert(!SEG2 || MID, "a second segment is a second feed-forward");
    static_assert(!HATT || (HEAD && !HDW && !MID && D == 256), "the attention input stage: conv-in chain, 4 heads x 64");
    constexpr bool WIDE = D > 256;                         // D = 512 (config 4): feed-forward in two halves of FF, K-chunked head / tail weight rings
    static_assert(!WIDE || (!HATT && !TVT && !SEG2 && D == 512 && (!HDW || (!MID && TSTEPS == 0))), "the wide instances: plain macaron / conv-in / final chains at D = 512 (+ the depthwise head)");
    static_assert(!FSPLIT || (WIDE && MID && TSTEPS == 0), "the pair split: a wide feed-forward chain without a tail");
    static_assert(!TSPLIT || (WIDE && !MID && (TSTEPS > 0 || HSTEPS > 0)), "the tail split: a wide chain without a feed-forward, half of the last product's columns per workgroup");
    constexpr bool PAIRED = FSPLIT || TSPLIT;
    static_assert(!CIN || (HDW && D == 256 && !WIDE), "the conv-in input stage: in front of the depthwise stage, D = 256 (one out-proj fragment and one GLU pair per wavefront)");
    constexpr bool HSPLIT = TSPLIT && TSTEPS == 0;         // no tail: the HEAD's columns are split, its rows go straight to out_f32 (no LayerNorm behind it)
    static_assert(!TVT || (TAIL && !TGLU && D == 256), "transposed values: a fused-QKV tail with 64-wide heads");
    constexpr int DWK = 15, DWH = (DWK - 1) / 2, DWROWS = RBM + DWK - 1;     // depthwise taps, halo, rows of the halo tile
    constexpr int KS1 = (D + 31) / 32;                     // 32-wide K slices of a D-long row
    constexpr int KP = KS1 * 32;
    constexpr int NF2 = D / 16;                            // 16-column fragments of a D-wide output
    constexpr int XS_STRIDE = D + 4;                       // f32 tile rows (floats)
    constexpr int XN_STRIDE = KP + 16;                     // 16-bit tile rows (halfs): row step = 8 words mod 64
    constexpr int FH = MID && WIDE ? 2 : 1;                // WIDE: the 32 x FF hidden tile + the 32 x D operand tile exceed the 160 KB, so FF runs in two halves
    constexpr int FFH = FF / FH;                           //       (phase 1, phase 2, phase 1, phase 2; the output accumulators stay in registers across them)
    constexpr int HS = FFH + 16;                           // hidden tile rows (halfs): same rule (FFH/2 words = 0 or 32 mod 64)
    constexpr int VPL = (D + 255) / 256;
    constexpr int TFR = TGLU ? 2 : 1;                      // fragments per wavefront per tail step ((value, gate) pairs for GLU)
    static_assert(D % 16 == 0 && (D <= 256 || D == 512) && MF == 2 && RPW == 2, "row chain supports D % 16 == 0, D <= 256, and D = 512");
    static_assert(!MID || (FFH % 64 == 0 && (XN_STRIDE / 2) % 64 % 16 == 8 && (HS / 2) % 64 % 16 == 8), "FF % 64 == 0 and conflict-free LDS strides");

    // region A: the f32 x tile between HEAD and LN_in, then the 16-bit hidden tile, then the f32 y tile of the FFN
    constexpr int A_MAIN = MID && RBM * HS * 2 > RBM * XS_STRIDE * 4 ? RBM * HS * 2 : RBM * XS_STRIDE * 4;
    constexpr int CIN_ROWS = 48;                                            // CIN: the 32 + 14 halo rows as three 16-row MFMA fragments
    constexpr int TAPS_OFF = CIN ? ((CIN_ROWS * XS_STRIDE * 4 + 1023) / 1024) * 1024 : DWROWS * D * 2;   // CIN: behind the 48-row f32 tile of the out-projection
    constexpr int A_DW = HDW ? TAPS_OFF + DWK * D * 4 : 0;                // 16-bit halo tile + the f32 taps of the depthwise input stage
    constexpr int A_ATT = HATT ? 4 * RBM * XS_STRIDE * 4 : 0;             // four partial context tiles (one per key quarter), f32
    constexpr int A_BYTES = A_MAIN > A_DW ? (A_MAIN > A_ATT ? A_MAIN : A_ATT) : (A_DW > A_ATT ? A_DW : A_ATT);
    __shared__ __attribute__((aligned(16))) unsigned char lds_a[A_BYTES];
    __shared__ __attribute__((aligned(16))) u16 xn[(CIN ? CIN_ROWS : RBM) * XN_STRIDE];
    float* const xs = (float*)lds_a;
    u16* const hid = (u16*)lds_a;

    const int tid = threadIdx.x;
    int lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    // HATT tiles never cross an utterance (the attention stage reads one utterance's keys): tile = (b, 32 frames), rows past the
    // utterance's end are treated like rows past M everywhere below (clamped loads, no stores)
    // XCD-aware order: workgroups go round-robin over the 8 XCDs, so the tiles of ONE utterance get ids 8 apart -- they share an L2 and its
    // keys / values are fetched from memory once, not once per XCD (groups of 8 utterances; padding workgroups of the last group exit)
    const int att_tiles = HATT ? (a.att_T + RBM - 1) / RBM : 1;
    const int att_j = blockIdx.x >> 3;
    const int att_b = HATT ? (att_j / att_tiles) * 8 + (blockIdx.x & 7) : 0, att_t0 = HATT ? (att_j % att_tiles) * RBM : 0;
    if constexpr (HATT) {
        if ((int64_t)att_b * a.att_T >= a.M) return;     // uniform, before any barrier
    }
    // FSPLIT: workgroup id -> (row tile, half of FF).  Workgroups go round-robin over the 8 XCDs: XCDs 0-3 take the first half, 4-7 the second, so an
    // XCD's L2 streams ONE half's weights (2 MB of a feed-forward, not 4); ids past the last tile exit
    // TSPLIT: the same pairs, each workgroup HALF of the tail's columns (rows and LayerNorm computed by both, the rows written by the first)
    const int fs_xcd = blockIdx.x & 7;
    const int fsel = FSPLIT ? fs_xcd >> 2 : 0, tsel = TSPLIT ? fs_xcd >> 2 : 0;
    const int fs_tile = PAIRED ? (int)(blockIdx.x >> 3) * 4 + (fs_xcd & 3) : (int)blockIdx.x;
    if constexpr (PAIRED) {
        if ((int64_t)fs_tile * RBM >= a.M) return;         // uniform, before any barrier
    }
    const int64_t row0 = HATT ? (int64_t)att_b * a.att_T + att_t0 : (int64_t)fs_tile * RBM;
    const int64_t Mlim = HATT ? (int64_t)(att_b + 1) * a.att_T : a.M;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    CFM_STAMP(0);

    // The tail's weight ring is declared here so that its first fill can be issued a phase or two before the tail runs
    // (with no FFN in between: at kernel start; otherwise at the end of the FFN's second phase).
    const int t_nfrags = TAIL ? a.tail_N / 16 : 1;
    auto t_frag0 = [&](int s) { return (s * NW + wave) * TFR + (TSPLIT ? tsel * (t_nfrags / 2) : 0); };
    auto t_clamp = [&](int f) { return f < t_nfrags ? f : t_nfrags - 1; };
    const u32x4* twp = (const u32x4*)a.tail_w + lane;
    constexpr int TRG = WIDE ? (MID ? (TFR == 2 ? 8 : 6) : 10) : TFR * KS1;      // WIDE: a ring over the steps' positions (ring_step) instead of one step's whole K
    u32x4 twr[TRG];
    auto tail_ptr = [&](int pos) {                         // WIDE: position -> fragment (step, kk, nf)
        const int st = pos / (KS1 * TFR), kk = (pos / TFR) % KS1, nf = pos % TFR;
        return twp + ((int64_t)t_clamp(t_frag0(st) + nf) * KS1 + kk) * 64;
    };
    auto tail_prefetch = [&]() {
        if constexpr (WIDE) {
#pragma unroll
            for (int t = 0; t < TRG; ++t) twr[t] = *tail_ptr(t);
        } else {
#pragma unroll
            for (int nf = 0; nf < TFR; ++nf)
#pragma unroll
                for (int kk = 0; kk < KS1; ++kk) twr[nf * KS1 + kk] = twp[((int64_t)t_clamp(t_frag0(0) + nf) * KS1 + kk) * 64];
        }
    };

    // ---- requests of the LayerNorm phase, issued at kernel entry so that their latency hides behind the head GEMM / the FFN ring
    //      fill: the wavefront's own rows (when they come from memory), the norm parameters, the pad-mask bytes
    f32x4 xres[RPW][VPL];                                  // this wavefront's rows of x: the residual, kept in registers
    int64_t ln_rows[RPW];
    f32x4 ln_gam[VPL], ln_bet[VPL];
    bool ln_keep[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int64_t gr = row0 + wave * RPW + rr;
        ln_rows[rr] = gr < Mlim ? gr : Mlim - 1;
        ln_keep[rr] = true;
        if constexpr (!HEAD) {
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                xres[rr][it] = c < D ? *(const f32x4*)(a.x + ln_rows[rr] * D + c) : zero4;
            }
            if constexpr (WIDE && !MID) {
                if (a.psum_in) {                           // the pair-split feed-forward of the previous launch: the two halves meet here, fixed order
#pragma unroll
                    for (int it = 0; it < VPL; ++it) {
                        const int c = (lane + 64 * it) * 4;
                        if (c < D) {
                            const f32x4 p0 = *(const f32x4*)(a.psum_in + ln_rows[rr] * D + c);
                            const f32x4 p1 = *(const f32x4*)(a.psum_in + (a.M + ln_rows[rr]) * D + c);
                            const f32x4 pb = *(const f32x4*)(a.psum_b2 + c);
                            xres[rr][it] = a.psum_alpha * ((p0 + p1) + pb) + xres[rr][it];
                        }
                    }
                }
            }
        }
    }
    auto ln_requests = [&]() {
        if (a.ln_g) {
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                ln_gam[it] = c < D ? *(const f32x4*)(a.ln_g + c) : zero4;
                ln_bet[it] = c < D ? *(const f32x4*)(a.ln_b + c) : zero4;
            }
        }
        if (a.ln_mask) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) ln_keep[rr] = a.ln_mask[ln_rows[rr]] != 0;
        }
    };
    if constexpr (!CIN) ln_requests();                     // (CIN: behind the conv-in stage -- at kernel entry they would be held, spilled, across it)

    // ================= HEAD: x = res + mask(A . Wh^T + bh) =====================================================
    if constexpr (HEAD) {
        const u32x4* wp = (const u32x4*)a.head_w + lane;
        auto frag0 = [&](int s) { return s * NW + wave + (HSPLIT ? tsel * (NF2 / 2) : 0); };
        auto clampf = [&](int f) { return f < NF2 ? f : NF2 - 1; };   // out-of-range fragments re-read the last one (unused)
        constexpr int HRG = WIDE ? 10 : KS1;
        u32x4 wr[HRG];
        auto head_ptr = [&](int pos) { return wp + ((int64_t)clampf(frag0(pos / KS1)) * KS1 + pos % KS1) * 64; };   // WIDE: position -> (step, kk)
        if constexpr (WIDE && !HDW) {
#pragma unroll
            for (int t = 0; t < HRG; ++t) wr[t] = *head_ptr(t);                                           // weights first
        } else if constexpr (!HDW && !HATT) {
#pragma unroll
            for (int kk = 0; kk < KS1; ++kk) wr[kk] = wp[((int64_t)clampf(frag0(0)) * KS1 + kk) * 64];   // weights first
        }
        constexpr int CPRW = KP / 8;                       // 16-byte chunks per row
        if constexpr (HDW) {
            // The head input is DepthwiseConv15 + bias + BatchNorm(eval, folded) + SiLU of the GLU rows (convolution.py:43-45),
            // computed here instead of in a launch of its own.  A (32 + 14)-row halo tile goes to region A; a thread owns TWO
            // channels of FOUR consecutive frames: 30 taps in registers, an 18-frame register window read once from LDS, packed
            // fp32 FMAs in the tap order of cfm_dwconv_bn_silu (bit-identical results), 16-bit results into the xn tile.
            u16* const halo = (u16*)lds_a;
            constexpr int C8 = D / 8, CP = D / 2, RG = RBM / 4;
            constexpr int NPASS = (CP * RG + NT - 1) / NT;      // (channel pair, frame group) items per thread: 1 up to D = 256, 2 at D = 512
            static_assert(NPASS == 1 || (NT % CP == 0 && (CP * RG) % NT == 0), "whole passes, the same channel pair in every pass of a thread");
            constexpr int NHALO = CIN ? 1 : (DWROWS * C8 + NT - 1) / NT;
            u32x4 hv[NHALO];
            if constexpr (!CIN) {
#pragma unroll
            for (int i = 0; i < NHALO; ++i) {               // all requests first, then the LDS stores.  UNCONDITIONAL loads (clamped address, value
                const int id = tid + i * NT;                // selected afterwards): behind a branch each load waits out its own latency
                const int idc = id < DWROWS * C8 ? id : DWROWS * C8 - 1;
                const int64_t grow = row0 - DWH + idc / C8;
                const int64_t gc = grow < 0 ? 0 : (grow < Mlim ? grow : Mlim - 1);
                const u32x4 v = *(const u32x4*)(a.head_a + gc * D + (idc % C8) * 8);
                hv[i] = (id < DWROWS * C8 && grow >= 0 && grow < Mlim) ? v : (u32x4){0u, 0u, 0u, 0u};
            }
            }
            const int cp = tid < CP * RG ? tid % CP : 0;                         // channel pair (wave-uniform: CP % 64 == 0 or idle tail; the same in every pass)
            float* const taps = (float*)(lds_a + TAPS_OFF);                      // [D][15] as in memory, staged with 16-byte loads
            constexpr int NTAP4 = (DWK * D / 4 + NT - 1) / NT;                   // 16-byte pieces of the taps per thread
            static_assert((DWK * D) % 4 == 0, "16-byte pieces of the taps");
            f32x4 tv[NTAP4];
#pragma unroll
            for (int i = 0; i < NTAP4; ++i) tv[i] = *(const f32x4*)(a.dw_w + 4 * (tid + i * NT < DWK * D / 4 ? tid + i * NT : 0));
            const f32x2 pb = *(const f32x2*)(a.dw_b + 2 * cp), ps = *(const f32x2*)(a.dw_scale + 2 * cp), ph = *(const f32x2*)(a.dw_shift + 2 * cp);
            if constexpr (CIN) {
                // ================= conv-in input stage =====================================================================
                // The halo tile is COMPUTED here: the conv-in chain of this block (attention.py:99 out-projection + residual, encoder_layer.py:62-64
                // LN_conv with the pad mask, convolution.py:41-42 pointwise-conv-1 + GLU) on the tile's 32 + 14 halo rows -- three 16-row MFMA
                // fragments; the 14 rows recomputed per tile cost no extra weight bytes (0.375 MB per workgroup either way) and save a launch, the
                // [M, D] GLU round trip and the halo load's latency.  Same operations on the same values in the same order as the stand-alone conv-in
                // chain, row by row: bit-identical (asserted against the unchained path).
                constexpr int MF3 = CIN_ROWS / 16, RPW3 = CIN_ROWS / NW;
                static_assert(NF2 == NW && RPW3 * NW == CIN_ROWS, "one out-proj fragment and one GLU fragment pair per wavefront, three rows per wavefront");
                const u32x4* cwp = (const u32x4*)a.cin_w + lane;
                u32x4 cwr[KS1];
#pragma unroll
                for (int kk = 0; kk < KS1; ++kk) cwr[kk] = cwp[((int64_t)wave * KS1 + kk) * 64];            // weights first
                constexpr int NCTX = (CIN_ROWS * CPRW + NT - 1) / NT;                                       // the context rows (clamped), 16 bit: all requests first
                u32x4 cxv[NCTX];
#pragma unroll
                for (int i = 0; i < NCTX; ++i) {
                    const int id = tid + i * NT < CIN_ROWS * CPRW ? tid + i * NT : CIN_ROWS * CPRW - 1;
                    int64_t grow = row0 - DWH + id / CPRW;
                    grow = grow < 0 ? 0 : (grow < Mlim ? grow : Mlim - 1);
                    cxv[i] = *(const u32x4*)(a.cin_a + grow * D + (id % CPRW) * 8);
                }
                const int ccol = wave * 16 + 4 * g;
                const f32x4 cbb = *(const f32x4*)(a.cin_b + ccol);
                f32x4 crs[MF3];
#pragma unroll
                for (int mf = 0; mf < MF3; ++mf) {
                    int64_t gr = row0 - DWH + mf * 16 + l15;
                    gr = gr < 0 ? 0 : (gr < Mlim ? gr : Mlim - 1);
                    crs[mf] = *(const f32x4*)(a.cin_res + gr * D + ccol);
                }
#pragma unroll
                for (int i = 0; i < NCTX; ++i) {
                    const int id = tid + i * NT;
                    if (id < CIN_ROWS * CPRW) *(u32x4*)(xn + (id / CPRW) * XN_STRIDE + (id % CPRW) * 8) = cxv[i];
                }
#pragma unroll
                for (int i = 0; i < NTAP4; ++i)
                    if (tid + i * NT < DWK * D / 4) *(f32x4*)(taps + 4 * (tid + i * NT)) = tv[i];
                __syncthreads();
                // the row phase's operands (LN_conv parameters, pad-mask bytes): requested behind the first barrier (it waits for every outstanding request: only the
                // context tile and the residual rows are in front of it), they land during the product
                f32x4 cg[VPL], cb[VPL];
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    cg[it] = c < D ? *(const f32x4*)(a.cin_ln_g + c) : zero4;
                    cb[it] = c < D ? *(const f32x4*)(a.cin_ln_b + c) : zero4;
                }
                unsigned ckb[RPW3];                            // the raw bytes: a comparison here would wait for them in front of the product
#pragma unroll
                for (int rr = 0; rr < RPW3; ++rr) {
                    int64_t gr = row0 - DWH + wave * RPW3 + rr;
                    gr = gr < 0 ? 0 : (gr < Mlim ? gr : Mlim - 1);
                    ckb[rr] = 1u;
                    if (a.cin_mask) ckb[rr] = a.cin_mask[gr];
                }
                {
                    f32x4 cacc[MF3][1];
                    const int nx0[1] = {0};
                    linear_step<HT, KS1, 1, XN_STRIDE, MF3>(xn, cwp, false, nx0, cwr, cacc, g, l15);
#pragma unroll
                    for (int mf = 0; mf < MF3; ++mf) {
                        f32x4 v = cacc[mf][0] + cbb;
                        v += crs[mf];
                        *(f32x4*)(xs + (mf * 16 + l15) * XS_STRIDE + ccol) = v;
                    }
                }
                CFM_STAMP(14);
                // the GLU product's weights (value / gate fragments 2 wave, 2 wave + 1) and the row phase's operands: requested now, used after the barrier
                const u32x4* ctp = (const u32x4*)a.cin_tw + lane;
                constexpr int CRG = 8;                         // a ring over the 2 x KS1 fragments (all 16 at once + three row fragments: over the 128 VGPRs)
                u32x4 ctw[CRG];
                auto ct_ptr = [&](int pos) { return ctp + ((int64_t)(2 * wave + pos % 2) * KS1 + pos / 2) * 64; };
#pragma unroll
                for (int t = 0; t < CRG; ++t) ctw[t] = *ct_ptr(t);
                __syncthreads();
                {   // rows: the residual rows of THIS tile -> cin_out; LN_conv (+ pad mask) of all 48 -> the operand tile
                    f32x4 cv[RPW3][VPL];
#pragma unroll
                    for (int rr = 0; rr < RPW3; ++rr) {
                        const int r = wave * RPW3 + rr;
                        const int64_t gr = row0 - DWH + r;
#pragma unroll
                        for (int it = 0; it < VPL; ++it) {
                            const int c = (lane + 64 * it) * 4;
                            cv[rr][it] = c < D ? *(const f32x4*)(xs + r * XS_STRIDE + c) : zero4;
                            if (c < D && r >= DWH && r < DWH + RBM && gr < Mlim) *(f32x4*)(a.cin_out + gr * D + c) = cv[rr][it];
                        }
                    }
                    rows_layernorm<RPW3, VPL, D>(cv, cg, cb, a.eps, lane);
#pragma unroll
                    for (int rr = 0; rr < RPW3; ++rr)
#pragma unroll
                        for (int it = 0; it < VPL; ++it) {
                            const int c = (lane + 64 * it) * 4;
                            if (c < KP) {
                                const f32x4 o = (c < D && ckb[rr] != 0u) ? cv[rr][it] : zero4;
                                *(u32x2*)(xn + (wave * RPW3 + rr) * XN_STRIDE + c) = (u32x2){pack2<HT>(o.x, o.y), pack2<HT>(o.z, o.w)};
                            }
                        }
                }
                const f32x4 tb0 = *(const f32x4*)(a.cin_tb + (2 * wave) * 16 + 4 * g), tb1 = *(const f32x4*)(a.cin_tb + (2 * wave + 1) * 16 + 4 * g);
                __syncthreads();                               // the operand tile is complete; every read of the f32 tile is done (the halo tile overlays it)
                CFM_STAMP(15);
                {
                    f32x4 tacc[MF3][2];
                    ring_step<HT, KS1, 2, CRG, XN_STRIDE, decltype(ct_ptr), MF3>(xn, 0, 2 * KS1, ct_ptr, ctw, tacc, g, l15);
#pragma unroll
                    for (int mf = 0; mf < MF3; ++mf) {
                        const int r = mf * 16 + l15;
                        const int64_t gr = row0 - DWH + r;
                        f32x4 v0 = tacc[mf][0] + tb0;
                        const f32x4 v1 = tacc[mf][1] + tb1;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v0[q] *= sigmoidf_(v1[q]);
                        const bool in = gr >= 0 && gr < Mlim;
                        const u32x2 pk = in ? (u32x2){pack2<HT>(v0.x, v0.y), pack2<HT>(v0.z, v0.w)} : (u32x2){0u, 0u};
                        if (r < DWROWS) *(u32x2*)(halo + r * D + wave * 16 + 4 * g) = pk;
                    }
                }
                ln_requests();
            } else {
#pragma unroll
            for (int i = 0; i < NHALO; ++i) {
                const int id = tid + i * NT;
                if (id < DWROWS * C8) *(u32x4*)(halo + (id / C8) * D + (id % C8) * 8) = hv[i];
            }
#pragma unroll
            for (int i = 0; i < NTAP4; ++i)
                if (tid + i * NT < DWK * D / 4) *(f32x4*)(taps + 4 * (tid + i * NT)) = tv[i];
            }
            // pad columns of the xn tile (K padded to a multiple of 32)
            if constexpr (KP > D) {
                for (int id = tid; id < RBM * (KP - D) / 2; id += NT) {
                    const int r = id / ((KP - D) / 2), c = D + 2 * (id % ((KP - D) / 2));
                    *(unsigned*)(xn + r * XN_STRIDE + c) = 0u;
                }
            }
            __syncthreads();
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
            const bool worker = tid + pass * NT < CP * RG;
            const int rg = worker ? (tid + pass * NT) / CP : 0;                  // frame group (wave-uniform)
            if (worker) {
                f32x2 tw[DWK];                                 // taps of channels 2cp, 2cp+1: 30 consecutive floats
                {
                    f32x2 raw[DWK];
#pragma unroll
                    for (int k = 0; k < DWK; ++k) raw[k] = *(const f32x2*)(taps + 2 * cp * DWK + 2 * k);
#pragma unroll
                    for (int k = 0; k < DWK; ++k) {
                        const int i0 = k, i1 = DWK + k;            // positions in the 30-float run
                        tw[k] = (f32x2){(i0 & 1) ? raw[i0 >> 1].y : raw[i0 >> 1].x, (i1 & 1) ? raw[i1 >> 1].y : raw[i1 >> 1].x};
                    }
                }
                f32x2 win[4 + DWK - 1];
#pragma unroll
                for (int j = 0; j < 4 + DWK - 1; ++j) {
                    const unsigned v = *(const unsigned*)(halo + (rg * 4 + j) * D + 2 * cp);
                    win[j] = (f32x2){HT::to_f32((u16)(v & 0xffffu)), HT::to_f32((u16)(v >> 16))};
                }
                // frames outside [0, T) of their own utterance are zero padding; only groups near an edge need the selects
                const int64_t g0 = row0 + rg * 4;
                const int t0 = (int)((unsigned)g0 % (unsigned)a.dw_T);
                // wave-uniform, so that it stays a scalar branch: if-converted, both variants would run with selects everywhere
                const bool edge = __any(t0 < DWH || t0 + 3 + DWH >= a.dw_T) != 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ti = (int)((unsigned)(t0 + i) % (unsigned)a.dw_T);   // the group may cross into the next utterance
                    f32x2 acc = (f32x2){0.f, 0.f};
                    if (edge) {
#pragma unroll
                        for (int k = 0; k < DWK; ++k) {
                            const int tt = ti - DWH + k;
                            const f32x2 xv = (tt >= 0 && tt < a.dw_T) ? win[i + k] : (f32x2){0.f, 0.f};
                            acc = (f32x2){fmaf(tw[k].x, xv.x, acc.x), fmaf(tw[k].y, xv.y, acc.y)};
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < DWK; ++k) acc = (f32x2){fmaf(tw[k].x, win[i + k].x, acc.x), fmaf(tw[k].y, win[i + k].y, acc.y)};
                    }
                    const f32x2 y = (f32x2){fmaf(acc.x + pb.x, ps.x, ph.x), fmaf(acc.y + pb.y, ps.y, ph.y)};   // as the scalar kernel contracts it
                    const unsigned o = g0 + i < Mlim ? pack2<HT>(siluf_(y.x), siluf_(y.y)) : 0u;
                    *(unsigned*)(xn + (rg * 4 + i) * XN_STRIDE + 2 * cp) = o;
                }
            }
            }   // passes
        } else if constexpr (HATT) {
            // ================= attention input stage ==================================================================
            // The head input IS the attention context of this tile's 32 frames (attention.py:81-96; the stand-alone kernel is
            // attention.hip cfm_attn2_kernel): no launch of its own, no [M, D] context round trip.  16 wavefronts = 4 heads x 4 KEY
            // QUARTERS (64 keys each, T <= 256).  Swapped MFMA roles as there: S^T[key, q] = K~ . Q~^T with the key fragment as the
            // A operand -- loaded straight from the qkv rows in memory, one 16-byte piece per lane, and shared by BOTH 16-query fragments
            // of the tile from registers -- and O^T[d, q] = V^T . P^T with V^T fragments read from the transposed copy the macaron
            // chain's tail wrote (one 16-byte piece per lane).  Nothing is staged in LDS;
            // what crosses wavefronts is the softmax: per-query maxima of the four quarters (one barrier), then the partial row
            // sums and the four partial context tiles, added in a fixed order by all 1 024 threads while they normalise, round and
            // write the 16-bit context tile the head GEMM reads (two more barriers).  Same arithmetic per element as cfm_attn2_kernel
            // up to the order of the f32 sums over keys.
            // MEASURED (config 2, 12 launches per step, profiles/r02_merged_attention.txt): 27.7 us per launch against 9.7 (cfm_attn2_kernel) +
            // 13.1 (conv-in chain) = 22.8 us for the two launches it replaces -- SLOWER, so the encoder does not use it by default
            // (encoder_layer.MERGE_ATTENTION).  Knock-outs: without any of the stage's global loads the launch takes 18.2 us (13.1 + ~5 of
            // MFMA / softmax / three barriers / the partial-tile exchange); the loads add 10 us -- values 6.1, keys 3.4, mask bytes 2.3,
            // queries 1.7 when removed one at a time -- because a 32-frame tile needs ALL FOUR heads' keys and values of its utterance (255 KB
            // per workgroup, twice the bytes per CU of the 64-query stand-alone kernel), nothing else in the chain can run before them, and at
            // kernel entry they come from memory, not L2 (the L2s are written back and invalidated between dependent launches).  The launch
            // it saves is worth ~5 us; the exposed fill costs more.
            __shared__ float att_mx[16 * RBM], att_ls[16 * RBM];
            __shared__ __attribute__((aligned(16))) uint8_t att_mk[256];
            const int T = a.att_T;
            const int h = wave >> 2, kq = wave & 3;
            if (tid < 256) {                                  // key validity of the utterance, one byte per key (read back as 4-byte words)
                bool ok = tid < T;
                if (ok && a.att_mask) ok = a.att_mask[(int64_t)att_b * a.att_m_sb + tid] != 0;
                att_mk[tid] = ok ? 1 : 0;
            }
            const bool pos = a.att_p != nullptr;
            u32x4 qu[2][2], qv[2][2];
            float bd[2] = {0.f, 0.f};
            {
                u32x4 qraw[2][2], praw[2];
                f32x4 bu[2][2], bv[2][2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int d0 = h * 64 + kk * 32 + g * 8;
#pragma unroll
                    for (int qf = 0; qf < 2; ++qf) {
                        int qr = att_t0 + qf * 16 + l15;
                        qr = qr < T ? qr : T - 1;
                        qraw[qf][kk] = *(const u32x4*)(a.att_qkv + ((int64_t)att_b * T + qr) * (3 * D) + d0);
                    }
                    if (pos) {
                        bu[kk][0] = *(const f32x4*)(a.att_bias_u + d0); bu[kk][1] = *(const f32x4*)(a.att_bias_u + d0 + 4);
                        bv[kk][0] = *(const f32x4*)(a.att_bias_v + d0); bv[kk][1] = *(const f32x4*)(a.att_bias_v + d0 + 4);
                        praw[kk] = *(const u32x4*)(a.att_p + (int64_t)att_b * a.att_p_sb + d0);
                    }
                }
#pragma unroll
                for (int qf = 0; qf < 2; ++qf)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const u32x4 raw = qraw[qf][kk];
                        if (!pos) {
                            qu[qf][kk] = raw;
                            qv[qf][kk] = raw;
                        } else {
                            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
                            float f[8];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                f[2 * i] = HT::to_f32((u16)(w[i] & 0xffffu));
                                f[2 * i + 1] = HT::to_f32((u16)(w[i] >> 16));
                            }
                            const f32x4 fa = {f[0], f[1], f[2], f[3]}, fb = {f[4], f[5], f[6], f[7]};
                            qu[qf][kk] = pack8<HT>(fa + bu[kk][0], fb + bu[kk][1]);
                            qv[qf][kk] = pack8<HT>(fa + bv[kk][0], fb + bv[kk][1]);
                            // bd_i = (q_i + v) . p_b : this lane's 8 d-values of the slice (the positional term of the batch path is one value per query)
                            const unsigned pw[4] = {praw[kk].x, praw[kk].y, praw[kk].z, praw[kk].w};
                            const unsigned qw[4] = {qv[qf][kk].x, qv[qf][kk].y, qv[qf][kk].z, qv[qf][kk].w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                bd[qf] = fmaf(HT::to_f32((u16)(qw[i] & 0xffffu)), HT::to_f32((u16)(pw[i] & 0xffffu)), bd[qf]);
                                bd[qf] = fmaf(HT::to_f32((u16)(qw[i] >> 16)), HT::to_f32((u16)(pw[i] >> 16)), bd[qf]);
                            }
                        }
                    }
                if (pos) {
#pragma unroll
                    for (int qf = 0; qf < 2; ++qf) {
                        bd[qf] += __shfl_xor(bd[qf], 16, 64);
                        bd[qf] += __shfl_xor(bd[qf], 32, 64);
                    }
                }
            }
            // ---- S^T for this wavefront's 64 keys x 32 queries.  Row i of key fragment f stands for key 32 (f >> 1) + 8 (i >> 2) + 4 (f & 1) + (i & 3)
            //      of the quarter (a row permutation of the K~ loads, free): the 8 probabilities a lane packs for one 32-key MFMA step are then
            //      8 CONSECUTIVE keys, and the matching V^T fragment is one 16-byte piece per lane.
            f32x4 s[2][4];
            u32x4 vf[2][4];
            {
                u32x4 kf[4][2];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    int key = kq * 64 + 32 * (f >> 1) + 8 * (l15 >> 2) + 4 * (f & 1) + (l15 & 3);
                    key = key < T ? key : T - 1;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        kf[f][kk] = *(const u32x4*)(a.att_qkv + ((int64_t)att_b * T + key) * (3 * D) + D + h * 64 + kk * 32 + g * 8);
                }
                // V^T fragments of these keys: requested behind the keys, used after the softmax
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int fd = 0; fd < 4; ++fd)
                        vf[k2][fd] = *(const u32x4*)(a.att_vt + ((int64_t)(att_b * 4 + h) * 64 + fd * 16 + l15) * a.att_vt_ld + kq * 64 + k2 * 32 + 8 * g);
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int qf = 0; qf < 2; ++qf) {
                        s[qf][f] = zero4;
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) s[qf][f] = HT::mfma(kf[f][kk], qu[qf][kk], s[qf][f]);
                    }
            }
            __syncthreads();                                  // the validity bytes
            // ---- key validity, scale, the positional constant; per-query maximum of this quarter
            unsigned okbits = 0u;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const unsigned mb = *(const unsigned*)(att_mk + kq * 64 + 32 * (f >> 1) + 8 * g + 4 * (f & 1));
#pragma unroll
                for (int r = 0; r < 4; ++r) okbits |= ((mb >> (8 * r)) & 0xffu) ? 1u << (f * 4 + r) : 0u;
            }
            float tmax[2] = {-INFINITY, -INFINITY};
#pragma unroll
            for (int qf = 0; qf < 2; ++qf) {
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = (okbits >> (f * 4 + r)) & 1u ? (s[qf][f][r] + bd[qf]) * a.att_scale : -INFINITY;
                        s[qf][f][r] = x;
                        tmax[qf] = fmaxf(tmax[qf], x);
                    }
                tmax[qf] = fmaxf(tmax[qf], __shfl_xor(tmax[qf], 16, 64));
                tmax[qf] = fmaxf(tmax[qf], __shfl_xor(tmax[qf], 32, 64));
                if (g == 0) att_mx[wave * RBM + qf * 16 + l15] = tmax[qf];
            }
            __syncthreads();
            float* const opart = (float*)lds_a;
#pragma unroll
            for (int qf = 0; qf < 2; ++qf) {
                const int q = qf * 16 + l15;
                const float m = fmaxf(fmaxf(att_mx[(h * 4 + 0) * RBM + q], att_mx[(h * 4 + 1) * RBM + q]),
                                      fmaxf(att_mx[(h * 4 + 2) * RBM + q], att_mx[(h * 4 + 3) * RBM + q]));
                float psum = 0.f;
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = (m == -INFINITY) ? 0.f : __expf(s[qf][f][r] - m);
                        s[qf][f][r] = pv;
                        psum += pv;
                    }
                psum += __shfl_xor(psum, 16, 64);
                psum += __shfl_xor(psum, 32, 64);
                if (g == 0) att_ls[wave * RBM + q] = psum;
                // ---- O^T partial of this quarter
                f32x4 acc_o[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const u32x4 ph = pack8<HT>(s[qf][2 * k2], s[qf][2 * k2 + 1]);
#pragma unroll
                    for (int fd = 0; fd < 4; ++fd) acc_o[fd] = HT::mfma(vf[k2][fd], ph, acc_o[fd]);
                }
#pragma unroll
                for (int fd = 0; fd < 4; ++fd) *(f32x4*)(opart + (kq * RBM + q) * XS_STRIDE + h * 64 + fd * 16 + 4 * g) = acc_o[fd];
            }
            __syncthreads();
            {   // all threads: context[r][c8 .. c8+7] = (sum of the four quarters) / l, rounded to the operand type, into the head GEMM's input tile
                const int r = tid >> 5, c8 = (tid & 31) * 8, hh = c8 >> 6;
                const float l_tot = (att_ls[(hh * 4 + 0) * RBM + r] + att_ls[(hh * 4 + 1) * RBM + r]) +
                                    (att_ls[(hh * 4 + 2) * RBM + r] + att_ls[(hh * 4 + 3) * RBM + r]);
                const float inv = l_tot > 0.f ? __builtin_amdgcn_rcpf(l_tot) : 0.f;
                f32x4 o0, o1;
                {
                    const float* p0 = opart + r * XS_STRIDE + c8;
                    constexpr int QS = RBM * XS_STRIDE;
                    o0 = (*(const f32x4*)p0 + *(const f32x4*)(p0 + QS)) + (*(const f32x4*)(p0 + 2 * QS) + *(const f32x4*)(p0 + 3 * QS));
                    o1 = (*(const f32x4*)(p0 + 4) + *(const f32x4*)(p0 + QS + 4)) + (*(const f32x4*)(p0 + 2 * QS + 4) + *(const f32x4*)(p0 + 3 * QS + 4));
                }
                *(u32x4*)(xn + r * XN_STRIDE + c8) = pack8<HT>(o0 * inv, o1 * inv);
            }
#pragma unroll
            for (int kk = 0; kk < KS1; ++kk) wr[kk] = wp[((int64_t)clampf(frag0(0)) * KS1 + kk) * 64];   // (earlier, they spill: 128 VGPRs)
        } else {
            // stage the 16-bit input tile (rows clamped), zero-padded to KP columns
            for (int id = tid; id < RBM * CPRW; id += NT) {
                const int r = id / CPRW, c = id % CPRW;
                int64_t grow = row0 + r;
                grow = grow < Mlim ? grow : Mlim - 1;
                const u32x4 v = c * 8 < D ? *(const u32x4*)(a.head_a + grow * D + c * 8) : (u32x4){0u, 0u, 0u, 0u};
                *(u32x4*)(xn + r * XN_STRIDE + c * 8) = v;
            }
        }
        if constexpr (HDW) {                               // (after the depthwise stage: its register window leaves no room before)
            if constexpr (WIDE) {
#pragma unroll
                for (int t = 0; t < HRG; ++t) wr[t] = *head_ptr(t);
            } else {
#pragma unroll
                for (int kk = 0; kk < KS1; ++kk) wr[kk] = wp[((int64_t)clampf(frag0(0)) * KS1 + kk) * 64];
            }
        }
        __syncthreads();
        CFM_STAMP(1);
#pragma unroll
        for (int s = 0; s < HSTEPS; ++s) {
            const int f = frag0(s);
            const int nx[1] = {clampf(frag0(s + 1 < HSTEPS ? s + 1 : s))};
            // epilogue operands of this step, issued before its MFMAs
            const int col = clampf(f) * 16 + 4 * g;
            f32x4 rs[MF];
            bool keep[MF];
            int64_t grows[MF];
            const f32x4 bb = *(const f32x4*)(a.head_b + col);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int64_t gr = row0 + mf * 16 + l15;
                grows[mf] = gr < Mlim ? gr : Mlim - 1;
                keep[mf] = true;
                rs[mf] = *(const f32x4*)(a.head_res + grows[mf] * D + col);
            }
            if (a.head_mask) {
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) keep[mf] = a.head_mask[grows[mf]] != 0;
            }
            f32x4 acc[MF][1];
            if constexpr (WIDE) ring_step<HT, KS1, 1, HRG, XN_STRIDE>(xn, s, HSTEPS * KS1, head_ptr, wr, acc, g, l15);
            else linear_step<HT, KS1, 1, XN_STRIDE>(xn, wp, s + 1 < HSTEPS, nx, wr, acc, g, l15);
            if (f < NF2) {
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    f32x4 v = acc[mf][0] + bb;
                    if (!keep[mf]) v = zero4;
                    v += rs[mf];
                    if constexpr (HSPLIT) {                 // this workgroup's half of the columns is all there is: no row phase behind it
                        if (row0 + mf * 16 + l15 < Mlim) *(f32x4*)(a.out_f32 + grows[mf] * D + col) = v;
                    } else {
                        *(f32x4*)(xs + (mf * 16 + l15) * XS_STRIDE + col) = v;
                    }
                }
            }
        }
        if constexpr (HSPLIT) return;
        __syncthreads();
    }
    if constexpr (TAIL && !MID) tail_prefetch();           // lands during the LayerNorm phase
    CFM_STAMP(2);

    // ================= FFN weight stream (declared here: its first ring fill is issued before the LayerNorm phase) ======
    // phase 1: this wavefront owns hidden-fragment PAIRS q = i * NW + wave (i < P1); phase 2: output fragments 2 np, 2 np + 1
    // over the K half kh.  One stream of positions per wavefront,
    //     phase 1: pos = (i * KS1 + kk) * 2 + nf          phase 2: pos = NPOS1 + k * 2 + nf
    // a ring of RING registers holds the next RING fragments, slot = pos % RING, and the slot just consumed is refilled with
    // the fragment RING positions ahead.
    constexpr int NPAIR1 = FFH / 32, P1 = (NPAIR1 + NW - 1) / NW;          // per half of FF (FH = 1: the whole of it)
    constexpr int KPARTS = WIDE ? 1 : 2;                   // phase 2: wavefronts per output-fragment pair (parts of K)
    constexpr int KS2 = FF / 32, KS2H = FFH / 32, KH = KS2H / KPARTS;
    constexpr int FHL = FSPLIT ? 1 : FH;                   // halves of FF this workgroup runs (FSPLIT: one, chosen by fsel)
    constexpr int NPOS1 = P1 * KS1 * 2, NPOS2 = KH * 2, NPOSH = NPOS1 + NPOS2, NPOS = FHL * NPOSH;
    constexpr int RING = FSPLIT ? 14 : WIDE && TAIL ? 8 : 10;                           // (measured, config 2 step: 8 .. 12 slots within 0.3 %; 16 spills ~10 VGPRs at the 128-register budget)
    static_assert(!MID || ((NF2 + 1) / 2 <= NW / KPARTS && KS2H % KPARTS == 0), "phase 2 maps (fragment pair, K part) onto 16 wavefronts");
    static_assert(!(WIDE && MID) || NPAIR1 % NW == 0, "wide: whole rounds of hidden-fragment pairs");
    int np = KPARTS == 2 ? wave & 7 : wave, kh = KPARTS == 2 ? wave >> 3 : 0;
    const u32x4* w1p = (const u32x4*)a.w1f + lane;
    const u32x4* w2p = (const u32x4*)a.w2n + lane;
    const float *seg_b1 = a.b1, *seg_b2 = a.b2, *seg_ln1_g = a.ln1_g, *seg_ln1_b = a.ln1_b;
    float* seg_out = a.out_f32;
    float seg_alpha = a.alpha;
    f32x4 vcur[RPW][VPL];                                  // SEG2: the rows between the segments
    f32x4 nx_g[VPL], nx_b[VPL];                            // SEG2: the second segment's input-norm parameters, requested during the first
    u32x4 ring[RING];
    auto pair_of = [&](int i) { const int q = i * NW + wave; return q < NPAIR1 ? q : NPAIR1 - 1; };
    auto frag_ptr = [&](int gpos) {
        const int fh = FSPLIT ? fsel : gpos / NPOSH, pos = gpos % NPOSH;     // (FH = 1: fh = 0, pos = gpos)
        if (pos < NPOS1) {
            const int i = pos / (2 * KS1), kk = (pos / 2) % KS1, nf = pos & 1;
            return w1p + ((int64_t)(2 * (fh * NPAIR1 + pair_of(i)) + nf) * KS1 + kk) * 64;
        }
        const int k = (pos - NPOS1) / 2, nf = pos & 1;
        const int n = 2 * np + nf < NF2 ? 2 * np + nf : NF2 - 1;
        return w2p + ((int64_t)n * KS2 + fh * KS2H + kh * KH + k) * 64;
    };
    auto refill = [&](int pos) {
        if (pos + RING < NPOS) ring[pos % RING] = *frag_ptr(pos + RING);
    };
    // (SEG2, measured: letting the last RING refills of the first segment fetch the second segment's first W1 fragments -- one weight stream across
    // the boundary -- shortens the second segment's LayerNorm phase by 1.4 k cycles and lengthens the first segment's phase 2 by 3.7 k: 62.3 vs
    // 61.0 us per launch; bit-identical, not kept.)
    // parameters of the post norms: requested during the FFN, used after it
    f32x4 pn_b2[VPL], pn_g1[VPL], pn_b1[VPL], pn_g2[VPL], pn_be2[VPL];
    f32x4 v[RPW][VPL];                                     // the rows after the feed-forward (+ LN1)
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {     // unrolled: with a run-time segment index the pointer selects spill (127 VGPRs, 173 SGPRs)
    const bool last_seg = sg == NSEG - 1;
    if constexpr (SEG2) {
        if (sg == 1) {
            // the second segment's fragment offsets are the first's: left visible, the compiler keeps all of them (and a lane-derived column index)
            // alive across the whole segment -- 102 SGPRs spilled to VGPR lanes + 9 VGPRs to scratch (40 B per lane, 10 MB per launch).
            // Opaque copies of the wavefront scalars make them new values here: no spills, .private_segment_fixed_size 0, step -1.8 %.
            asm volatile("" : "+s"(wave), "+s"(np), "+s"(kh));
            if constexpr (CIN) asm volatile("" : "+v"(lane));   // (and the lane: column offsets derived from it are otherwise kept from the first phase to the last store)
            w1p = (const u32x4*)a.s2_w1f + lane; w2p = (const u32x4*)a.s2_w2n + lane;
            seg_b1 = a.s2_b1; seg_b2 = a.s2_b2; seg_ln1_g = nullptr; seg_ln1_b = nullptr; seg_out = a.s2_out_f32; seg_alpha = a.s2_alpha;
        }
    }
    if constexpr (MID) {
#pragma unroll
        for (int t = 0; t < RING; ++t)
            if (t < NPOS) ring[t] = *frag_ptr(t);
    }

    // ================= rows -> LN_in -> xn ======================================================================
    if (SEG2 && sg == 1) {
        // second segment: the rows are in registers (vcur), no mask
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int it = 0; it < VPL; ++it) xres[rr][it] = vcur[rr][it];
        f32x4 w[RPW][VPL];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int it = 0; it < VPL; ++it) w[rr][it] = xres[rr][it];
        rows_layernorm<RPW, VPL, D>(w, nx_g, nx_b, a.eps, lane);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < KP) {
                    const f32x4 o = c < D ? w[rr][it] : zero4;
                    *(u32x2*)(xn + (wave * RPW + rr) * XN_STRIDE + c) = (u32x2){pack2<HT>(o.x, o.y), pack2<HT>(o.z, o.w)};
                }
            }
    } else {
        int64_t (&grows)[RPW] = ln_rows;
        f32x4 (&gam)[VPL] = ln_gam, (&bet)[VPL] = ln_bet;
        bool (&keep)[RPW] = ln_keep;
        if constexpr (HEAD) {                              // rows produced by the head GEMM: out of the f32 x tile in LDS
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    xres[rr][it] = c < D ? *(const f32x4*)(xs + (wave * RPW + rr) * XS_STRIDE + c) : zero4;
                }
        }
        // WIDE + feed-forward: the residual rows do not stay in registers across the feed-forward (16 VGPRs = 4 ring slots); rows made by the head
        // GEMM are parked in out_f32 (this thread's own elements, overwritten with the result at the end), rows from memory are read again
        if constexpr (!MID || (WIDE && HEAD)) {            // no FFN here: the rows ARE the new residual stream
            if (a.out_f32 && (!TSPLIT || tsel == 0)) {
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                    for (int it = 0; it < VPL; ++it) {
                        const int c = (lane + 64 * it) * 4;
                        if (c < D && row0 + wave * RPW + rr < Mlim) *(f32x4*)(a.out_f32 + (row0 + wave * RPW + rr) * D + c) = xres[rr][it];
                    }
            }
        }
        f32x4 w[RPW][VPL];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int it = 0; it < VPL; ++it) w[rr][it] = xres[rr][it];
        if (a.ln_g) rows_layernorm<RPW, VPL, D>(w, gam, bet, a.eps, lane);
        if constexpr (WIDE && !MID && !TAIL) {             // a bare rows chain (the reduce behind a pair-split final feed-forward): the normalised rows ARE the output
            if (a.out2_f32) {
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                    for (int it = 0; it < VPL; ++it) {
                        const int c = (lane + 64 * it) * 4;
                        if (c < D && row0 + wave * RPW + rr < Mlim) *(f32x4*)(a.out2_f32 + (row0 + wave * RPW + rr) * D + c) = w[rr][it];
                    }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < KP) {
                    const f32x4 o = (c < D && keep[rr]) ? w[rr][it] : zero4;
                    *(u32x2*)(xn + (wave * RPW + rr) * XN_STRIDE + c) = (u32x2){pack2<HT>(o.x, o.y), pack2<HT>(o.z, o.w)};
                }
            }
    }
    __syncthreads();                                       // xn complete; every read of the x tile in region A is done
    CFM_STAMP(sg == 0 ? 3 : 10);

    // ================= MID: feed-forward in two phases around the hidden tile ====================================
    if constexpr (MID) {
        f32x4 acc2[MF][2];
#pragma unroll
        for (int fhi = 0; fhi < FHL; ++fhi) {               // (FH = 1 except WIDE)
        const int fh = FSPLIT ? fsel : fhi;
        const int pbase = fhi * NPOSH;
        if (fhi > 0) {
            __syncthreads();                                // every wavefront is done reading the previous half's hidden tile
            asm volatile("" : "+s"(wave));                  // (new values per half: see the note at the segment boundary)
            CFM_STAMP(15);
        }
        // ---- phase 1: hidden = SiLU(xn . W1^T + b1) -> region A, 16 bit
#pragma unroll
        for (int i = 0; i < P1; ++i) {
            const int q = i * NW + wave;
            const bool valid = NPAIR1 % NW == 0 ? true : q < NPAIR1;
            const int qc = pair_of(i);
            const f32x4 bb0 = *(const f32x4*)(seg_b1 + (2 * (fh * NPAIR1 + qc)) * 16 + 4 * g);      // used after this pair's MFMAs
            const f32x4 bb1 = *(const f32x4*)(seg_b1 + (2 * (fh * NPAIR1 + qc) + 1) * 16 + 4 * g);
            auto xfrag = [&](int mf, int kk) { return *(const u32x4*)(xn + (mf * 16 + l15) * XN_STRIDE + kk * 32 + 8 * g); };
            f32x4 acc1[MF][2];
            u32x4 xf[2][MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                xf[0][mf] = xfrag(mf, 0);
                acc1[mf][0] = zero4;
                acc1[mf][1] = zero4;
            }
#pragma unroll
            for (int kk = 0; kk < KS1; ++kk) {
                const int pos = pbase + (i * KS1 + kk) * 2;
                if (kk + 1 < KS1) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) xf[(kk + 1) & 1][mf] = xfrag(mf, kk + 1);
                }
#pragma unroll
                for (int nf = 0; nf < 2; ++nf)
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) acc1[mf][nf] = HT::mfma(ring[(pos + nf) % RING], xf[kk & 1][mf], acc1[mf][nf]);
                refill(pos);
                refill(pos + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (valid) {
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    f32x4 h0 = acc1[mf][0] + bb0, h1 = acc1[mf][1] + bb1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        h0[r] = siluf_(h0[r]);
                        h1[r] = siluf_(h1[r]);
                    }
                    u16* hp = hid + (mf * 16 + l15) * HS + (2 * q) * 16 + 4 * g;
                    *(u32x2*)hp = (u32x2){pack2<HT>(h0.x, h0.y), pack2<HT>(h0.z, h0.w)};
                    *(u32x2*)(hp + 16) = (u32x2){pack2<HT>(h1.x, h1.y), pack2<HT>(h1.z, h1.w)};
                }
            }
        }
        __syncthreads();
        CFM_STAMP(fhi > 0 ? 14 : sg == 0 ? 4 : 11);

        // ---- phase 2: y = hidden . W2^T, fragments (2 np, 2 np + 1), K half kh
        asm volatile("" : "+s"(np), "+s"(kh));             // same reason: phase 2's 64 scalar offsets are otherwise computed (and spilled) at kernel entry
        {
            auto hfrag = [&](int mf, int k) { return *(const u32x4*)(hid + (mf * 16 + l15) * HS + (kh * KH + k) * 32 + 8 * g); };
            u32x4 hf[2][MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                hf[0][mf] = hfrag(mf, 0);
                if (fhi == 0) {
                    acc2[mf][0] = zero4;
                    acc2[mf][1] = zero4;
                }
            }
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int pos = pbase + NPOS1 + k * 2;
                if (k + 1 < KH) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) hf[(k + 1) & 1][mf] = hfrag(mf, k + 1);
                }
#pragma unroll
                for (int nf = 0; nf < 2; ++nf)
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) acc2[mf][nf] = HT::mfma(ring[(pos + nf) % RING], hf[k & 1][mf], acc2[mf][nf]);
                refill(pos);
                refill(pos + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (FHL > 1) {
            // the accumulators are next read after the LAST half: without an anchor here the compiler sinks this half's MFMAs below the next half's
            // barrier and keeps every fragment they read alive (spilled) until then
            asm volatile("" : "+v"(acc2[0][0]), "+v"(acc2[0][1]), "+v"(acc2[1][0]), "+v"(acc2[1][1]));
        }
        }   // FF halves
        // requests that land during the tile exchange below: post-norm parameters and the tail's first weights
        if constexpr (!FSPLIT) {
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            const int c = (lane + 64 * it) * 4;
            pn_b2[it] = c < D ? *(const f32x4*)(seg_b2 + c) : zero4;
        }
        if constexpr (WIDE) {
            const float* rsrc = HEAD ? a.out_f32 : a.x;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    xres[rr][it] = c < D ? *(const f32x4*)(rsrc + ln_rows[rr] * D + c) : zero4;
                }
        }
        if (seg_ln1_g) {
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                pn_g1[it] = c < D ? *(const f32x4*)(seg_ln1_g + c) : zero4;
                pn_b1[it] = c < D ? *(const f32x4*)(seg_ln1_b + c) : zero4;
            }
        }
        if constexpr (SEG2) {
            if (sg == 0) {
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    nx_g[it] = c < D ? *(const f32x4*)(a.s2_ln_g + c) : zero4;
                    nx_b[it] = c < D ? *(const f32x4*)(a.s2_ln_b + c) : zero4;
                }
            }
        }
        if (a.ln2_g && last_seg) {
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                pn_g2[it] = c < D ? *(const f32x4*)(a.ln2_g + c) : zero4;
                pn_be2[it] = c < D ? *(const f32x4*)(a.ln2_b + c) : zero4;
            }
        }
        }   // !FSPLIT
        if constexpr (TAIL && !WIDE) {
            if (last_seg) tail_prefetch();
        }
        __syncthreads();                                   // every wavefront is done reading the hidden tile
        // the two K halves meet in the f32 y tile (region A again), fixed order: half 0 stores, half 1 adds
        auto ytile = [&](int mf, int nf) { return xs + (mf * 16 + l15) * XS_STRIDE + (2 * np + nf) * 16 + 4 * g; };
        if (kh == 0) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf)
                    if (2 * np + nf < NF2) *(f32x4*)ytile(mf, nf) = acc2[mf][nf];
        }
        __syncthreads();
        if constexpr (KPARTS == 2) {
            if (kh == 1) {
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                    for (int nf = 0; nf < 2; ++nf)
                        if (2 * np + nf < NF2) *(f32x4*)ytile(mf, nf) = *(const f32x4*)ytile(mf, nf) + acc2[mf][nf];
            }
            __syncthreads();
        }
    }
    CFM_STAMP(sg == 0 ? 5 : 12);

    // ================= post norms: y1 -> out_f32, y2 -> out16 / next LDS tile =====================================
    if constexpr (FSPLIT) {
        // this half's partial sums leave as they are (no bias, no residual): full rows, 16 bytes per lane
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
            const int64_t grow = row0 + r;
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < D && grow < Mlim) *(f32x4*)(a.psum_out + ((int64_t)fsel * a.M + grow) * D + c) = *(const f32x4*)(xs + r * XS_STRIDE + c);
            }
        }
        return;
    }
    if constexpr (MID) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                v[rr][it] = zero4;
                if (c < D) v[rr][it] = seg_alpha * (*(const f32x4*)(xs + r * XS_STRIDE + c) + pn_b2[it]) + xres[rr][it];
            }
        }
        if (seg_ln1_g) rows_layernorm<RPW, VPL, D>(v, pn_g1, pn_b1, a.eps, lane);
        if (seg_out) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    const int64_t grow = row0 + wave * RPW + rr;
                    if (c < D && grow < Mlim) *(f32x4*)(seg_out + grow * D + c) = v[rr][it];     // (non-temporal here and on the tail: kernel -1 us, step unchanged)
                }
        }
        if constexpr (SEG2) {
            if (!last_seg) {
                CFM_STAMP(13);
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                    for (int it = 0; it < VPL; ++it) vcur[rr][it] = v[rr][it];
                continue;                                  // (the next segment's barrier after its LayerNorm also ends this one's reads of the y tile)
            }
        }
        if (a.ln2_g) {
            rows_layernorm<RPW, VPL, D>(v, pn_g2, pn_be2, a.eps, lane);
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    const int r = wave * RPW + rr;
                    const int64_t grow = row0 + r;
                    if (c < KP) {
                        const f32x4 o = c < D ? v[rr][it] : zero4;
                        const u32x2 pk = (u32x2){pack2<HT>(o.x, o.y), pack2<HT>(o.z, o.w)};
                        if constexpr (TAIL) *(u32x2*)(xn + r * XN_STRIDE + c) = pk;   // the tail's input tile
                        if (c < D && grow < Mlim && a.out16) *(u32x2*)((u16*)a.out16 + grow * D + c) = pk;
                        if (c < D && grow < Mlim && a.out2_f32) *(f32x4*)(a.out2_f32 + grow * D + c) = o;
                    }
                }
        }
        if constexpr (TAIL) __syncthreads();
    }
    }   // segments
    CFM_STAMP(6);

    // ================= TAIL: t = xn . Wt^T + bt (GLU optional), 16-bit store ===================================
    if constexpr (TAIL) {
        if constexpr (MID && WIDE) tail_prefetch();        // (not earlier: the ring would not fit beside the feed-forward's registers)
        const int ldo = TGLU ? a.tail_N / 2 : a.tail_N;
#pragma unroll
        for (int s = 0; s < TSTEPS; ++s) {
            const int f = t_frag0(s);
            const int fn = t_frag0(s + 1 < TSTEPS ? s + 1 : s);
            int nx[TFR];
            f32x4 bb[TFR];
#pragma unroll
            for (int nf = 0; nf < TFR; ++nf) {
                nx[nf] = t_clamp(fn + nf);
                bb[nf] = *(const f32x4*)(a.tail_b + t_clamp(f + nf) * 16 + 4 * g);      // issued before this step's MFMAs
            }
            f32x4 acc[MF][TFR];
            if constexpr (WIDE) ring_step<HT, KS1, TFR, TRG, XN_STRIDE>(xn, s, TSTEPS * KS1 * TFR, tail_ptr, twr, acc, g, l15);
            else linear_step<HT, KS1, TFR, XN_STRIDE>(xn, twp, s + 1 < TSTEPS, nx, twr, acc, g, l15);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int64_t grow = row0 + mf * 16 + l15;
                if (grow >= Mlim) continue;
                if constexpr (TGLU) {
                    if (f + 1 < t_nfrags) {                   // (value, gate) fragment pairs: tail_N % 32 == 0
                        f32x4 v0 = acc[mf][0] + bb[0];
                        const f32x4 v1 = acc[mf][TFR - 1] + bb[TFR - 1];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v0[r] *= sigmoidf_(v1[r]);
                        const int64_t o = grow * ldo + (f >> 1) * 16 + 4 * g;
                        *(u32x2*)((u16*)a.tail_out + o) = (u32x2){pack2<HT>(v0.x, v0.y), pack2<HT>(v0.z, v0.w)};
                    }
                } else {
                    if (f < t_nfrags) {
                        const f32x4 v0 = acc[mf][0] + bb[0];
                        if (TVT && f * 16 >= 2 * D) {             // wave-uniform: a value fragment goes out transposed per head
                            const int bb_ = (int)(grow / a.vt_T), t = (int)(grow - (int64_t)bb_ * a.vt_T);
                            const int c = f * 16 + 4 * g - 2 * D;  // head (c >> 6), d = c & 63 .. + 3
                            u16* vp = a.tail_vt + ((int64_t)bb_ * D + c) * a.vt_ld + t;
                            vp[0] = HT::from_f32(v0.x);
                            vp[a.vt_ld] = HT::from_f32(v0.y);
                            vp[2 * (int64_t)a.vt_ld] = HT::from_f32(v0.z);
                            vp[3 * (int64_t)a.vt_ld] = HT::from_f32(v0.w);
                        } else {
                            const int64_t o = grow * ldo + f * 16 + 4 * g;
                            *(u32x2*)((u16*)a.tail_out + o) = (u32x2){pack2<HT>(v0.x, v0.y), pack2<HT>(v0.z, v0.w)};
                        }
                    }
                }
            }
        }
    }
    CFM_STAMP(7);
}

template <typename HT, int D, int FF, int HS, bool HDW, bool MID, int TS, bool TGLU, bool HATT = false, bool TVT = false, bool SEG2 = false, bool FSPLIT = false, bool TSPLIT = false, bool CIN = false>
int launch_chain(const ChainArgs& a, hipStream_t s, const char* name, double flops) {
    // HATT: tiles do not cross utterances (B x ceil(T / 32) workgroups); FSPLIT: groups of 8 workgroups = 4 row tiles x 2 halves of FF
    const unsigned tiles = (unsigned)((a.M + RBM - 1) / RBM);
    const unsigned grid = HATT ? (unsigned)((((a.M / a.att_T) + 7) / 8) * 8 * ((a.att_T + RBM - 1) / RBM)) : (FSPLIT || TSPLIT) ? ((tiles + 3) / 4) * 8 : tiles;
    CfmProfScope prof(name, s, flops, (double)a.M * D * 8);
    CFM_LAUNCH((cfm_rowchain_kernel<HT, D, FF, HS, HDW, MID, TS, TGLU, HATT, TVT, SEG2, FSPLIT, TSPLIT, CIN>), dim3(grid), dim3(NT), 0, s, a);
    return cfm_launch_status(name);
}

}  // namespace

extern "C" int cfm_rowchain_supported(int32_t D, int32_t FF) { return (D == 256 && FF == 2048) || (D == 144 && FF == 576) || (D == 512 && FF == 2048); }
extern "C" int cfm_rowchain_dw_supported(int32_t D) { return D <= 256; }
extern "C" int cfm_rowchain_pair_supported(int32_t D, int32_t FF) { return D == 512 && FF == 2048; }   // the depthwise input stage (one (channel pair, frame group) per thread)

extern "C" int cfm_rowchain(const cfm_rowchain_desc* d, cfm_stream_t stream) {
    CFM_CHECK_ARG(d, "cfm_rowchain: null descriptor");
    const bool att = d->att_qkv != nullptr, tvt = d->tail_vt != nullptr;
    const bool head = d->head_a != nullptr || att, mid = d->w1f != nullptr, tail = d->tail_w != nullptr, dw = d->dw_w != nullptr;
    CFM_CHECK_ARG(!att || (!d->head_a && !mid && tail && d->tail_glu && !dw && d->D == 256 && d->att_H == 4 && d->att_vt && d->att_T > 0 && d->att_T <= 256 &&
                           d->M % d->att_T == 0 && d->att_vt_ld >= 256 && d->att_vt_ld % 4 == 0 && (!d->att_p || (d->att_bias_u && d->att_bias_v))),
                  "cfm_rowchain: the attention input stage needs the conv-in chain at D = 256 with 4 heads, T <= 256, M %% T == 0, transposed values with "
                  "a row stride >= 256 (multiple of 4) and, with positions, both biases");
    CFM_CHECK_ARG(!tvt || (tail && !d->tail_glu && d->D == 256 && d->tail_N == 768 && d->vt_T > 0 && d->M % d->vt_T == 0 && d->vt_ld >= d->vt_T),
                  "cfm_rowchain: transposed values need the fused-QKV tail at D = 256, M %% vt_T == 0 and vt_ld >= vt_T");
    const bool seg2 = d->s2_w1f != nullptr;
    CFM_CHECK_ARG(!seg2 || (mid && d->s2_w2n && d->s2_b1 && d->s2_b2 && d->s2_ln_g && d->s2_ln_b && d->s2_out_f32 && d->ln2_g),
                  "cfm_rowchain: a second segment needs a first feed-forward, its own weights / biases / input norm / output and ln2");
    CFM_CHECK_ARG(!dw || (head && (mid ? (!tail || seg2) : !tail) && d->dw_b && d->dw_scale && d->dw_shift && d->dw_K == 15 && d->dw_T > 0 && d->M % d->dw_T == 0),
                  "cfm_rowchain: the depthwise input stage needs a head (+ feed-forward) chain without a tail, bias/scale/shift, 15 taps and M %% dw_T == 0");
    CFM_CHECK_ARG(d->M > 0 && (d->D == 144 || d->D == 256 || d->D == 512), "cfm_rowchain: D=%d has no instance (144, 256, 512)", d->D);
    CFM_CHECK_ARG(d->D != 512 || (!att && !tvt && !seg2 && (!dw || (d->tail_pair && !mid && !tail))), "cfm_rowchain: D = 512 runs the plain macaron / conv-in / final chains (and the paired depthwise head) only");
    CFM_CHECK_ARG(d->w_dtype == CFM_BF16 || d->w_dtype == CFM_F16, "cfm_rowchain: w_dtype must be bf16 or fp16");
    CFM_CHECK_ARG(head || d->x, "cfm_rowchain: need x or a head input");
    CFM_CHECK_ARG(!head || (d->head_w && d->head_b && d->head_res), "cfm_rowchain: head needs weights, bias and residual");
    CFM_CHECK_ARG(!mid || (d->w2n && d->b1 && d->b2 && d->FF > 0), "cfm_rowchain: the feed-forward needs w1f, w2n, b1, b2 and FF");
    CFM_CHECK_ARG(!tail || (d->tail_b && d->tail_out && d->tail_N % 16 == 0 && d->tail_N > 0 && (!d->tail_glu || d->tail_N % 32 == 0)),
                  "cfm_rowchain: tail needs bias, output and N %% 16 == 0 (GLU: N %% 32 == 0)");
    CFM_CHECK_ARG(!tail || (mid ? d->ln2_g != nullptr : true), "cfm_rowchain: a tail after the FFN takes its input from the second LayerNorm");
    const bool ps_out = d->psum_out != nullptr, ps_in = d->psum_in != nullptr;
    CFM_CHECK_ARG(!ps_out || (mid && !tail && !seg2 && !dw && !att && !d->ln1_g && !d->ln2_g && cfm_rowchain_pair_supported(d->D, d->FF) &&
                              (!head || (d->out_f32 && d->out_f32 != d->head_res))),
                  "cfm_rowchain: psum_out needs a feed-forward chain at D = 512 without tail / post norms, and (with a head) out_f32 distinct from head_res");
    CFM_CHECK_ARG(!ps_in || (!head && !mid && d->D == 512 && d->psum_b2 && d->x), "cfm_rowchain: psum_in needs a rows chain (no head, no feed-forward) at D = 512, x and psum_b2");
    CFM_CHECK_ARG(!d->tail_pair || (!mid && d->D == 512 && (tail ? (d->tail_N / 16) % 2 == 0 : (head && dw && d->out_f32 && !d->ln_g)) && (!head || d->out_f32 != d->head_res)),
                  "cfm_rowchain: tail_pair needs a chain without a feed-forward at D = 512 -- with a tail, or the depthwise head alone (no LayerNorm, rows to out_f32) -- "
                  "and (with a head) out_f32 distinct from head_res");
    ChainArgs a;
    a.psum_out = d->psum_out; a.psum_in = d->psum_in; a.psum_b2 = d->psum_b2; a.psum_alpha = d->psum_alpha;
    const bool cin = d->cin_a != nullptr;
    CFM_CHECK_ARG(!cin || (seg2 && dw && d->D == 256 && d->cin_w && d->cin_b && d->cin_res && d->cin_out && d->cin_ln_g && d->cin_ln_b && d->cin_tail_w && d->cin_tail_b &&
                           d->head_res == d->cin_out && d->cin_out != d->cin_res),
                  "cfm_rowchain: the conv-in input stage (cin_*) needs the depthwise + final + next-macaron chain at D = 256, all of its operands, head_res == cin_out "
                  "and cin_out distinct from cin_res (halo rows of cin_res belong to other tiles)");
    a.cin_a = (const u16*)d->cin_a; a.cin_w = (const u16*)d->cin_w; a.cin_b = d->cin_b; a.cin_res = d->cin_res; a.cin_out = d->cin_out; a.cin_ln_g = d->cin_ln_g;
    a.cin_ln_b = d->cin_ln_b; a.cin_mask = d->cin_mask; a.cin_tw = (const u16*)d->cin_tail_w; a.cin_tb = d->cin_tail_b;
    a.x = d->x; a.head_a = (const u16*)d->head_a; a.head_w = (const u16*)d->head_w; a.head_b = d->head_b; a.head_res = d->head_res;
    a.dw_w = d->dw_w; a.dw_b = d->dw_b; a.dw_scale = d->dw_scale; a.dw_shift = d->dw_shift; a.dw_T = d->dw_T;
    a.head_mask = d->head_mask; a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.ln_mask = d->ln_mask; a.w1f = (const u16*)d->w1f; a.w2n = (const u16*)d->w2n;
    a.b1 = d->b1; a.b2 = d->b2; a.ln1_g = d->ln1_g; a.ln1_b = d->ln1_b; a.ln2_g = d->ln2_g; a.ln2_b = d->ln2_b; a.out_f32 = d->out_f32;
    a.out16 = d->out16; a.out2_f32 = d->out2_f32; a.tail_w = (const u16*)d->tail_w; a.tail_b = d->tail_b; a.tail_out = d->tail_out; a.M = d->M;
    a.tail_N = d->tail_N; a.out16_dtype = d->w_dtype; a.alpha = d->alpha; a.eps = d->eps;
    a.tail_vt = (u16*)d->tail_vt; a.vt_T = d->vt_T; a.vt_ld = d->vt_ld;
    a.att_qkv = (const u16*)d->att_qkv; a.att_vt = (const u16*)d->att_vt; a.att_p = (const u16*)d->att_p; a.att_bias_u = d->att_bias_u;
    a.att_bias_v = d->att_bias_v; a.att_mask = d->att_mask; a.att_p_sb = d->att_p_sb; a.att_m_sb = d->att_m_sb; a.att_T = d->att_T;
    a.att_vt_ld = d->att_vt_ld; a.att_scale = d->att_scale;
    a.s2_ln_g = d->s2_ln_g; a.s2_ln_b = d->s2_ln_b; a.s2_w1f = (const u16*)d->s2_w1f; a.s2_w2n = (const u16*)d->s2_w2n; a.s2_b1 = d->s2_b1;
    a.s2_b2 = d->s2_b2; a.s2_out_f32 = d->s2_out_f32; a.s2_alpha = d->s2_alpha;
    hipStream_t s = (hipStream_t)stream;
    const bool bf = d->w_dtype == CFM_BF16;
    const int tfrags = tail ? d->tail_N / 16 : 0;
    const int tsteps = !tail ? 0 : d->tail_glu ? (tfrags / 2 + NW - 1) / NW : (tfrags + NW - 1) / NW;
    const double M = (double)d->M;
    const double fl_head = head ? 2.0 * M * d->D * d->D : 0.0, fl_mid = mid ? 4.0 * M * d->D * d->FF : 0.0,
                 fl_tail = tail ? 2.0 * M * d->D * d->tail_N : 0.0;
    const double fl = fl_head + fl_mid * (seg2 ? 2.0 : 1.0) + fl_tail + (dw ? 2.0 * M * d->D * 15 : 0.0) + (att ? 4.0 * M * d->att_T * d->D : 0.0);
#define CFM_RC(HT, DD, FFV, HS, MIDV, TS, GLU, NAME) return launch_chain<HT, DD, FFV, HS, false, MIDV, TS, GLU>(a, s, NAME, fl)
#define CFM_RCDW(HT, DD, FFV, NAME) return launch_chain<HT, DD, FFV, 1, true, true, 0, false>(a, s, NAME, fl)
    // the three roles of a conformer block (+ a bare QKV projection and a bare LayerNorm), for D = 256 (ff 2048) and D = 144 (ff 576)
    if (d->D == 256 && (!mid || d->FF == 2048)) {
        if (att) { if (bf) return launch_chain<BF16, 256, 64, 1, false, false, 1, true, true, false>(a, s, "chain_attconvin_bf16_d256", fl); else return launch_chain<F16, 256, 64, 1, false, false, 1, true, true, false>(a, s, "chain_attconvin_f16_d256", fl); }
        if (tvt && !head && mid && tsteps == 3) { if (bf) return launch_chain<BF16, 256, 2048, 0, false, true, 3, false, false, true>(a, s, "chain_macaron_vt_bf16_d256", fl); else return launch_chain<F16, 256, 2048, 0, false, true, 3, false, false, true>(a, s, "chain_macaron_vt_f16_d256", fl); }
        if (tvt) return cfm_fail(CFM_ERR_UNSUPPORTED, "cfm_rowchain: transposed values only on the macaron chain");
        if (cin && seg2 && head && dw && tail && !d->tail_glu && tsteps == 3) { if (bf) return launch_chain<BF16, 256, 2048, 1, true, true, 3, false, false, false, true, false, false, true>(a, s, "chain_convin_dwfinal_macaron_bf16_d256", fl + 2.0 * M * d->D * 3 * d->D); else return launch_chain<F16, 256, 2048, 1, true, true, 3, false, false, false, true, false, false, true>(a, s, "chain_convin_dwfinal_macaron_f16_d256", fl + 2.0 * M * d->D * 3 * d->D); }
        if (seg2 && head && dw && tail && !d->tail_glu && tsteps == 3) { if (bf) return launch_chain<BF16, 256, 2048, 1, true, true, 3, false, false, false, true>(a, s, "chain_dwfinal_macaron_bf16_d256", fl); else return launch_chain<F16, 256, 2048, 1, true, true, 3, false, false, false, true>(a, s, "chain_dwfinal_macaron_f16_d256", fl); }
        if (seg2) return cfm_fail(CFM_ERR_UNSUPPORTED, "cfm_rowchain: a second segment only on the depthwise + final chain followed by a QKV tail (D = 256)");
        if (!head && mid && tail && !d->tail_glu && tsteps == 3) { if (bf) CFM_RC(BF16, 256, 2048, 0, true, 3, false, "chain_macaron_bf16_d256"); else CFM_RC(F16, 256, 2048, 0, true, 3, false, "chain_macaron_f16_d256"); }
        if (head && !mid && tail && d->tail_glu && tsteps == 1) { if (bf) CFM_RC(BF16, 256, 64, 1, false, 1, true, "chain_convin_bf16_d256"); else CFM_RC(F16, 256, 64, 1, false, 1, true, "chain_convin_f16_d256"); }
        if (head && mid && !tail && dw) { if (bf) CFM_RCDW(BF16, 256, 2048, "chain_dwfinal_bf16_d256"); else CFM_RCDW(F16, 256, 2048, "chain_dwfinal_f16_d256"); }
        // depthwise + BatchNorm + SiLU -> pointwise-conv-2 + pad mask + residual only (the split feed-forward path runs the rest: ffnsplit.hip)
        if (head && !mid && !tail && dw) { if (bf) return launch_chain<BF16, 256, 64, 1, true, false, 0, false>(a, s, "chain_dwhead_bf16_d256", fl); else return launch_chain<F16, 256, 64, 1, true, false, 0, false>(a, s, "chain_dwhead_f16_d256", fl); }
        if (head && mid && !tail) { if (bf) CFM_RC(BF16, 256, 2048, 1, true, 0, false, "chain_final_bf16_d256"); else CFM_RC(F16, 256, 2048, 1, true, 0, false, "chain_final_f16_d256"); }
        if (!head && !mid && tail && !d->tail_glu && tsteps == 3) { if (bf) CFM_RC(BF16, 256, 64, 0, false, 3, false, "chain_qkv_bf16_d256"); else CFM_RC(F16, 256, 64, 0, false, 3, false, "chain_qkv_f16_d256"); }
        if (!head && !mid && !tail) { if (bf) CFM_RC(BF16, 256, 64, 0, false, 0, false, "chain_rows_bf16_d256"); else CFM_RC(F16, 256, 64, 0, false, 0, false, "chain_rows_f16_d256"); }
    } else if (d->D == 144 && (!mid || d->FF == 576)) {
        if (!head && mid && tail && !d->tail_glu && tsteps == 2) { if (bf) CFM_RC(BF16, 144, 576, 0, true, 2, false, "chain_macaron_bf16_d144"); else CFM_RC(F16, 144, 576, 0, true, 2, false, "chain_macaron_f16_d144"); }
        if (head && !mid && tail && d->tail_glu && tsteps == 1) { if (bf) CFM_RC(BF16, 144, 64, 1, false, 1, true, "chain_convin_bf16_d144"); else CFM_RC(F16, 144, 64, 1, false, 1, true, "chain_convin_f16_d144"); }
        if (head && mid && !tail && dw) { if (bf) CFM_RCDW(BF16, 144, 576, "chain_dwfinal_bf16_d144"); else CFM_RCDW(F16, 144, 576, "chain_dwfinal_f16_d144"); }
        if (head && mid && !tail) { if (bf) CFM_RC(BF16, 144, 576, 1, true, 0, false, "chain_final_bf16_d144"); else CFM_RC(F16, 144, 576, 1, true, 0, false, "chain_final_f16_d144"); }
        if (!head && !mid && tail && !d->tail_glu && tsteps == 2) { if (bf) CFM_RC(BF16, 144, 64, 0, false, 2, false, "chain_qkv_bf16_d144"); else CFM_RC(F16, 144, 64, 0, false, 2, false, "chain_qkv_f16_d144"); }
        if (!head && !mid && !tail) { if (bf) CFM_RC(BF16, 144, 64, 0, false, 0, false, "chain_rows_bf16_d144"); else CFM_RC(F16, 144, 64, 0, false, 0, false, "chain_rows_f16_d144"); }
    } else if (d->D == 512 && (!mid || d->FF == 2048)) {
        if (d->tail_pair && !head && !d->tail_glu && tsteps == 6) { if (bf) return launch_chain<BF16, 512, 64, 0, false, false, 3, false, false, false, false, false, true>(a, s, "chain_qkv_pair_bf16_d512", fl); else return launch_chain<F16, 512, 64, 0, false, false, 3, false, false, false, false, false, true>(a, s, "chain_qkv_pair_f16_d512", fl); }
        if (d->tail_pair && head && d->tail_glu && tsteps == 2) { if (bf) return launch_chain<BF16, 512, 64, 2, false, false, 1, true, false, false, false, false, true>(a, s, "chain_convin_pair_bf16_d512", fl); else return launch_chain<F16, 512, 64, 2, false, false, 1, true, false, false, false, false, true>(a, s, "chain_convin_pair_f16_d512", fl); }
        if (d->tail_pair && head && dw && !tail) { if (bf) return launch_chain<BF16, 512, 64, 1, true, false, 0, false, false, false, false, false, true>(a, s, "chain_dwhead_pair_bf16_d512", fl); else return launch_chain<F16, 512, 64, 1, true, false, 0, false, false, false, false, false, true>(a, s, "chain_dwhead_pair_f16_d512", fl); }
        if (d->tail_pair) return cfm_fail(CFM_ERR_UNSUPPORTED, "cfm_rowchain: tail_pair on the q|k|v rows chain, the conv-in chain and the depthwise head only");
        if (!head && mid && tail && !d->tail_glu && tsteps == 6) { if (bf) CFM_RC(BF16, 512, 2048, 0, true, 6, false, "chain_macaron_bf16_d512"); else CFM_RC(F16, 512, 2048, 0, true, 6, false, "chain_macaron_f16_d512"); }
        if (head && !mid && tail && d->tail_glu && tsteps == 2) { if (bf) CFM_RC(BF16, 512, 64, 2, false, 2, true, "chain_convin_bf16_d512"); else CFM_RC(F16, 512, 64, 2, false, 2, true, "chain_convin_f16_d512"); }
        if (ps_out && head) { if (bf) return launch_chain<BF16, 512, 2048, 2, false, true, 0, false, false, false, false, true>(a, s, "chain_final_half_bf16_d512", fl); else return launch_chain<F16, 512, 2048, 2, false, true, 0, false, false, false, false, true>(a, s, "chain_final_half_f16_d512", fl); }
        if (ps_out) { if (bf) return launch_chain<BF16, 512, 2048, 0, false, true, 0, false, false, false, false, true>(a, s, "chain_macaron_half_bf16_d512", fl); else return launch_chain<F16, 512, 2048, 0, false, true, 0, false, false, false, false, true>(a, s, "chain_macaron_half_f16_d512", fl); }
        if (head && mid && !tail) { if (bf) CFM_RC(BF16, 512, 2048, 2, true, 0, false, "chain_final_bf16_d512"); else CFM_RC(F16, 512, 2048, 2, true, 0, false, "chain_final_f16_d512"); }
        if (!head && !mid && tail && !d->tail_glu && tsteps == 6) { if (bf) CFM_RC(BF16, 512, 64, 0, false, 6, false, "chain_qkv_bf16_d512"); else CFM_RC(F16, 512, 64, 0, false, 6, false, "chain_qkv_f16_d512"); }
        if (!head && !mid && !tail) { if (bf) CFM_RC(BF16, 512, 64, 0, false, 0, false, "chain_rows_bf16_d512"); else CFM_RC(F16, 512, 64, 0, false, 0, false, "chain_rows_f16_d512"); }
    }
#undef CFM_RC
#undef CFM_RCDW
    return cfm_fail(CFM_ERR_UNSUPPORTED, "cfm_rowchain: no instance for D=%d FF=%d tail_N=%d head=%d mid=%d tail=%d glu=%d", d->D, d->FF,
                    d->tail_N, (int)head, (int)mid, (int)tail, d->tail_glu);
}
