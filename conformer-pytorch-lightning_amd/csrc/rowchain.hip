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
//     [MID]   y  = x + alpha * ( W2 . act( W1 . xn + b1 ) + b2 )                    (the fused FFN of ffn.hip)
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
// All linears use the machinery proven in ffn.hip: fragment-major weights (one wavefront-load = 1 KB = one MFMA A
// fragment), streamed L2 -> VGPR by a pinned rolling ring, straight-line unrolled steps (no branches: see ffn.hip for
// why), swapped MFMA roles (a lane owns 4 consecutive output columns of one row), NW wavefronts that each own distinct
// output columns (HEAD/TAIL: no cross-wavefront reduction) or distinct FF slices (MID: fixed-order LDS reduction).
//
// NW = 8 (two wavefronts per SIMD) for D = 256.  A wavefront pays ~60-100 clk to ISSUE one 1 KB weight load and issues in
// order, so with one wavefront per SIMD the 32 loads of an FFN step (~3 k clk) serialise with its 64 MFMAs (1 k clk) -- measured
// 34 us for one workgroup alone, nowhere near either the L2 stream rate (scripts/ubench.hip: 57 B/clk/CU at 4 wavefronts,
// 109 at 8) or the MFMA rate.  With two wavefronts per SIMD one issues loads while the other runs MFMAs, and each wavefront
// has half as many loads to issue.  256 VGPRs per wavefront then: 128 accumulators + a 12-fragment unified weight ring.
#include <string>
#include <type_traits>

#include "cfm_common.h"

struct ChainArgs {
    const float* x;           // f32 rows (no HEAD)
    const float *py0, *py1, *pb2, *pln_g, *pln_b;   // optional reduce input (pending partial FFN of the previous kernel)
    float palpha;
    const u16* head_a;        // 16-bit [M,D] (HEAD)
    const u16* head_w;        // fragment-major [D/16][KS][64][8]
    const float* head_b;
    const float* head_res;    // f32 [M,D]
    const uint8_t* head_mask; // zero the head OUTPUT row where 0 (before the residual add)
    const float *ln_g, *ln_b;
    const uint8_t* ln_mask;   // zero the normalised row where 0
    const u16 *w1f, *w2f;
    const float *b1, *b2;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float* out_f32;
    void* out16;
    const u16* tail_w;        // fragment-major [tail_N/16][KS][64][8]
    const float* tail_b;
    void* tail_out;           // 16-bit [M, tail_N] (GLU: [M, tail_N/2])
    int64_t M;
    int FF, tail_N;
    int out16_dtype;
    float alpha, eps;
};

// Phase stamps for scripts/probe_chain.hip (built with -DCFM_CHAIN_STAMPS; never defined in the product build): thread 0 of each
// workgroup records the shader clock at every phase boundary.
#ifdef CFM_CHAIN_STAMPS
__device__ long long cfm_chain_stamps[1024 * 16];
#define CFM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) cfm_chain_stamps[blockIdx.x * 16 + (i)] = clock64(); } while (0)
#else
#define CFM_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int RBM = 32;

// One "linear step": this wavefront's two 16-column fragments (n-fragments f0, f0+1) over the whole K of the LDS tile.
// With `refill`, weights for the NEXT step (n-fragments nxa, nxb -- clamped by the caller when out of range) replace the
// current ones in the ring as they are consumed; the last step of a phase passes false (a compile-time constant after
// unrolling) and issues no loads.
template <typename HT, int KS1, int XN_STRIDE>
__device__ __forceinline__ void linear_step(const u16* xn, const u32x4* wp, bool refill, int nxa, int nxb, u32x4 (&wr)[2 * KS1],
                                            f32x4 (&acc)[2][2], int g, int l15) {
    u32x4 xf[2][KS1];
#pragma unroll
    for (int kk = 0; kk < KS1; ++kk)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) xf[mf][kk] = *(const u32x4*)(xn + (mf * 16 + l15) * XN_STRIDE + kk * 32 + 8 * g);
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS1; ++kk) {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) acc[mf][nf] = HT::mfma(wr[nf * KS1 + kk], xf[mf][kk], acc[mf][nf]);
        if (refill) {
            wr[kk] = wp[((int64_t)nxa * KS1 + kk) * 64];
            wr[KS1 + kk] = wp[((int64_t)nxb * KS1 + kk) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// s_setprio takes an immediate; the argument is a constant after unrolling and the switch folds away.
__device__ __forceinline__ void set_wave_priority(int p) {
    switch (p) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// LayerNorm of ROWS rows held one row per wavefront pass (lane owns columns (lane + 64 it) * 4 ..+3), in place.
template <int ROWS, int VPL, int D>
__device__ __forceinline__ void rows_layernorm(f32x4 (&v)[ROWS][VPL], const f32x4 (&gam)[VPL], const f32x4 (&bet)[VPL], float eps, int lane) {
    float mean[ROWS], rstd[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < VPL; ++it)
            if ((lane + 64 * it) * 4 < D) s += (v[rr][it].x + v[rr][it].y) + (v[rr][it].z + v[rr][it].w);
        mean[rr] = wave_sum(s) / (float)D;
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
        rstd[rr] = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
        for (int it = 0; it < VPL; ++it)
            if ((lane + 64 * it) * 4 < D) v[rr][it] = (v[rr][it] - mean[rr]) * rstd[rr] * gam[it] + bet[it];
}

template <typename HT, int D, int NW, int HSTEPS, int FSTEPS, int TSTEPS, bool TGLU>
__global__ __launch_bounds__(64 * NW) void cfm_rowchain_kernel(const ChainArgs a) {
    constexpr bool HEAD = HSTEPS > 0, MID = FSTEPS > 0, TAIL = TSTEPS > 0;
    constexpr int NT = 64 * NW;
    constexpr int RPW = RBM / NW;                          // rows per wavefront in the row-wise (LayerNorm) phases
    static_assert(NW == 4 || NW == 8, "4 or 8 wavefronts");
    constexpr int KS1 = (D + 31) / 32;
    constexpr int KP = KS1 * 32;
    constexpr int NF2 = D / 16;
    constexpr int MF = RBM / 16;
    constexpr int XS_STRIDE = D + 4;
    constexpr int XN_STRIDE = KP + 8;
    constexpr int VPL = (D + 255) / 256;
    static_assert(D % 16 == 0 && D <= 256 && MF == 2, "row chain supports D % 16 == 0, D <= 256");

    __shared__ __attribute__((aligned(16))) float xs[RBM * XS_STRIDE];
    __shared__ __attribute__((aligned(16))) float slab[(MID ? 2 : 0) * RBM * XS_STRIDE + 4];
    __shared__ __attribute__((aligned(16))) u16 xn[RBM * XN_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * RBM;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    CFM_STAMP(0);

    // The tail's weight ring is declared here so that its first fill can be issued a phase or two before the tail runs
    // (with no FFN in between: at kernel start; otherwise right after the FFN steps, to land during the reduction).
    const int t_nfrags = TAIL ? a.tail_N / 16 : 1;
    auto t_frag0 = [&](int s) { return (s * NW + wave) * 2; };
    auto t_clamp = [&](int f) { return f < t_nfrags ? f : t_nfrags - 1; };
    const u32x4* twp = (const u32x4*)a.tail_w + lane;
    u32x4 twr[2 * KS1];
    auto tail_prefetch = [&]() {
#pragma unroll
        for (int kk = 0; kk < KS1; ++kk) {
            twr[kk] = twp[((int64_t)t_clamp(t_frag0(0)) * KS1 + kk) * 64];
            twr[KS1 + kk] = twp[((int64_t)t_clamp(t_frag0(0) + 1) * KS1 + kk) * 64];
        }
    };

    // ================= HEAD: x = res + mask(A . Wh^T + bh) =====================================================
    if constexpr (HEAD) {
        const u32x4* wp = (const u32x4*)a.head_w + lane;
        auto frag0 = [&](int s) { return (s * NW + wave) * 2; };
        auto clampf = [&](int f) { return f < NF2 ? f : NF2 - 1; };   // out-of-range fragments re-read the last one (unused)
        u32x4 wr[2 * KS1];
#pragma unroll
        for (int kk = 0; kk < KS1; ++kk) {                 // weights first: their latency overlaps the tile staging below
            wr[kk] = wp[((int64_t)clampf(frag0(0)) * KS1 + kk) * 64];
            wr[KS1 + kk] = wp[((int64_t)clampf(frag0(0) + 1) * KS1 + kk) * 64];
        }
        if constexpr (TAIL && !MID) tail_prefetch();
        // stage the 16-bit input tile (rows clamped), zero-padded to KP columns
        constexpr int CPRW = KP / 8;                       // 16-byte chunks per row
        for (int id = tid; id < RBM * CPRW; id += NT) {
            const int r = id / CPRW, c = id % CPRW;
            int64_t grow = row0 + r;
            grow = grow < a.M ? grow : a.M - 1;
            const u32x4 v = c * 8 < D ? *(const u32x4*)(a.head_a + grow * D + c * 8) : (u32x4){0u, 0u, 0u, 0u};
            *(u32x4*)(xn + r * XN_STRIDE + c * 8) = v;
        }
        __syncthreads();
        CFM_STAMP(1);
#pragma unroll
        for (int s = 0; s < HSTEPS; ++s) {
            const int f = frag0(s);
            const int fn = frag0(s + 1 < HSTEPS ? s + 1 : s);
            // epilogue operands of this step, issued before its MFMAs
            f32x4 bb[2], rs[2][MF];
            bool keep[MF];
            int64_t grows[MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int64_t gr = row0 + mf * 16 + l15;
                grows[mf] = gr < a.M ? gr : a.M - 1;
                keep[mf] = true;
            }
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                const int col = clampf(f + nf) * 16 + 4 * g;
                bb[nf] = *(const f32x4*)(a.head_b + col);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) rs[nf][mf] = *(const f32x4*)(a.head_res + grows[mf] * D + col);
            }
            if (a.head_mask) {
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) keep[mf] = a.head_mask[grows[mf]] != 0;
            }
            f32x4 acc[2][2];
            linear_step<HT, KS1, XN_STRIDE>(xn, wp, s + 1 < HSTEPS, clampf(fn), clampf(fn + 1), wr, acc, g, l15);
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                const int col = (f + nf) * 16 + 4 * g;
                if (f + nf < NF2) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) {
                        f32x4 v = acc[mf][nf] + bb[nf];
                        if (!keep[mf]) v = zero4;
                        v += rs[nf][mf];
                        *(f32x4*)(xs + (mf * 16 + l15) * XS_STRIDE + col) = v;
                    }
                }
            }
        }
        __syncthreads();
    } else {
        if constexpr (TAIL && !MID) tail_prefetch();
    }
    CFM_STAMP(2);

    // ================= rows -> (x in LDS,) LN_in -> xn ===========================================================
    {
        f32x4 v[RPW][VPL];
        int64_t grows[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {                 // all of this wavefront's rows are requested before any is used
            const int r = wave * RPW + rr;
            const int64_t gr = row0 + r;
            grows[rr] = gr < a.M ? gr : a.M - 1;
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                if constexpr (HEAD) v[rr][it] = c < D ? *(const f32x4*)(xs + r * XS_STRIDE + c) : zero4;
                else v[rr][it] = c < D ? *(const f32x4*)(a.x + grows[rr] * D + c) : zero4;
            }
        }
        f32x4 gam[VPL], bet[VPL];
        bool keep[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) keep[rr] = true;
        if (a.ln_g) {
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                gam[it] = c < D ? *(const f32x4*)(a.ln_g + c) : zero4;
                bet[it] = c < D ? *(const f32x4*)(a.ln_b + c) : zero4;
            }
        }
        if (a.ln_mask) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) keep[rr] = a.ln_mask[grows[rr]] != 0;
        }
        if constexpr (!HEAD) {
            if (a.py0) {                                   // pending partial FFN of the previous kernel (+ its norm_final)
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                    for (int it = 0; it < VPL; ++it) {
                        const int c = (lane + 64 * it) * 4;
                        if (c < D) {
                            const f32x4 y = *(const f32x4*)(a.py0 + grows[rr] * D + c) + *(const f32x4*)(a.py1 + grows[rr] * D + c);
                            v[rr][it] += a.palpha * (y + *(const f32x4*)(a.pb2 + c));
                        }
                    }
                if (a.pln_g) {
                    f32x4 pg[VPL], pb[VPL];
#pragma unroll
                    for (int it = 0; it < VPL; ++it) {
                        const int c = (lane + 64 * it) * 4;
                        pg[it] = c < D ? *(const f32x4*)(a.pln_g + c) : zero4;
                        pb[it] = c < D ? *(const f32x4*)(a.pln_b + c) : zero4;
                    }
                    rows_layernorm<RPW, VPL, D>(v, pg, pb, a.eps, lane);
                }
            }
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    if (c < D) *(f32x4*)(xs + (wave * RPW + rr) * XS_STRIDE + c) = v[rr][it];
                }
        }
        if constexpr (!MID) {                              // no FFN here: the rows ARE the new residual stream
            if (a.out_f32) {
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                    for (int it = 0; it < VPL; ++it) {
                        const int c = (lane + 64 * it) * 4;
                        if (c < D && row0 + wave * RPW + rr < a.M) *(f32x4*)(a.out_f32 + (row0 + wave * RPW + rr) * D + c) = v[rr][it];
                    }
            }
        }
        if (a.ln_g) rows_layernorm<RPW, VPL, D>(v, gam, bet, a.eps, lane);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < KP) {
                    const f32x4 o = (c < D && keep[rr]) ? v[rr][it] : zero4;
                    *(u32x2*)(xn + (wave * RPW + rr) * XN_STRIDE + c) = (u32x2){pack2<HT>(o.x, o.y), pack2<HT>(o.z, o.w)};
                }
            }
    }
    __syncthreads();
    CFM_STAMP(3);

    // parameters of the post norms: requested after the FFN steps, used after the reduction
    f32x4 pn_b2[VPL], pn_g1[VPL], pn_b1[VPL], pn_g2[VPL], pn_be2[VPL];

    // ================= MID: fused feed-forward (see ffn.hip for the design notes) ================================
    if constexpr (MID) {
        auto xfrag = [&](int mf, int kk) { return *(const u32x4*)(xn + (mf * 16 + l15) * XN_STRIDE + kk * 32 + 8 * g); };
        const int nsteps_total = a.FF / 32;
        f32x4 acc2[MF][NF2];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF2; ++nf) acc2[mf][nf] = zero4;
        const u32x4* w1p = (const u32x4*)a.w1f + lane;
        const u32x4* w2p = (const u32x4*)a.w2f + lane;
        // One weight STREAM per wavefront: step s consumes NW1 fragments of W1 (in (kk, nf) order) and then NF2 fragments of
        // W2; a ring of RING registers holds the next RING fragments of that stream, slot = position % RING, and the slot
        // just consumed is refilled with the fragment RING positions ahead.  Everything about a position is a compile-time
        // constant after unrolling, so the ring lives in registers and the waits are exact vmcnt values.
        constexpr int NW1 = 2 * KS1, NFR = NW1 + NF2;
        constexpr int RING = NW == 4 ? 32 : 12;            // 16 spills at 256 VGPRs (128 of them accumulators)
        u32x4 ring[RING];
        f32x4 b1r[2];
        const int rot = (int)(blockIdx.x % FSTEPS);
        auto step_of = [&](int s) { int q = s + rot; q = q >= FSTEPS ? q - FSTEPS : q; return q * NW + wave; };
        auto step_clamped = [&](int s) { const int f = step_of(s); return f < nsteps_total ? f : nsteps_total - 1; };
        auto frag_ptr = [&](int fs, int p) {
            if (p < NW1) return w1p + ((int64_t)(2 * fs) * KS1 + (p & 1) * KS1 + (p >> 1)) * 64;   // (kk, nf) = (p >> 1, p & 1)
            return w2p + ((int64_t)fs * NF2 + (p - NW1)) * 64;
        };
        auto refill = [&](int s, int p) {                  // after consuming position (s, p)
            const int t = s * NFR + p + RING;
            const int s2 = t / NFR, p2 = t % NFR;
            if (s2 < FSTEPS) ring[(s * NFR + p) % RING] = *frag_ptr(step_clamped(s2), p2);
        };
        auto step = [&](int s) {
            const bool valid = step_of(s) < nsteps_total;
            const int nx = step_clamped(s + 1 < FSTEPS ? s + 1 : s);
            f32x4 acc1[MF][2];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) acc1[mf][nf] = zero4;
            u32x4 xf[2][MF];                               // activation fragments: this kk and the next (one-ahead LDS reads)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) xf[0][mf] = xfrag(mf, 0);
#pragma unroll
            for (int kk = 0; kk < KS1; ++kk) {
                if (kk + 1 < KS1) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) xf[(kk + 1) & 1][mf] = xfrag(mf, kk + 1);
                }
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
                        acc1[mf][nf] = HT::mfma(ring[(s * NFR + kk * 2 + nf) % RING], xf[kk & 1][mf], acc1[mf][nf]);
                }
                refill(s, kk * 2);
                refill(s, kk * 2 + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            const f32x4 bb0 = b1r[0], bb1 = b1r[1];
            if (s + 1 < FSTEPS) {
                b1r[0] = *(const f32x4*)(a.b1 + nx * 32 + 4 * g);
                b1r[1] = *(const f32x4*)(a.b1 + nx * 32 + 16 + 4 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
            u32x4 hf[MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                f32x4 h0 = acc1[mf][0] + bb0, h1 = acc1[mf][1] + bb1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h0[r] = siluf_(h0[r]);
                    h1[r] = siluf_(h1[r]);
                }
                hf[mf] = pack8<HT>(h0, h1);
                if (!valid) hf[mf] = (u32x4){0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int nf = 0; nf < NF2; ++nf) {
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) acc2[mf][nf] = HT::mfma(ring[(s * NFR + NW1 + nf) % RING], hf[mf], acc2[mf][nf]);
                refill(s, NW1 + nf);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {
            const int f0 = step_clamped(0);
            b1r[0] = *(const f32x4*)(a.b1 + f0 * 32 + 4 * g);
            b1r[1] = *(const f32x4*)(a.b1 + f0 * 32 + 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < RING; ++t)
                if (t / NFR < FSTEPS) ring[t] = *frag_ptr(step_clamped(t / NFR), t % NFR);
        }
#pragma unroll
        for (int s = 0; s < FSTEPS; ++s) {
            // the two wavefronts of a SIMD share its issue slots; the one that is behind gets priority, so that both finish
            // together instead of the younger one running its last steps alone (loads and MFMAs serialised again)
            if constexpr (NW == 8) set_wave_priority(3 - (s * 4) / FSTEPS);
            step(s);
        }
        if constexpr (NW == 8) set_wave_priority(0);
        CFM_STAMP(4);

        // requests that land during the reduction: post-norm parameters and the tail's first weights
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            const int c = (lane + 64 * it) * 4;
            pn_b2[it] = c < D ? *(const f32x4*)(a.b2 + c) : zero4;
        }
        if (a.ln1_g) {
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                pn_g1[it] = c < D ? *(const f32x4*)(a.ln1_g + c) : zero4;
                pn_b1[it] = c < D ? *(const f32x4*)(a.ln1_b + c) : zero4;
            }
        }
        if (a.ln2_g) {
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                pn_g2[it] = c < D ? *(const f32x4*)(a.ln2_g + c) : zero4;
                pn_be2[it] = c < D ? *(const f32x4*)(a.ln2_b + c) : zero4;
            }
        }
        if constexpr (TAIL) tail_prefetch();

        // cross-wavefront reduction through two LDS slabs, fixed order:
        //   NW = 4: (w0 + w2) + (w1 + w3)        NW = 8: ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7))
        auto slab_ptr = [&](int which, int mf, int nf) { return slab + which * RBM * XS_STRIDE + (mf * 16 + l15) * XS_STRIDE + nf * 16 + 4 * g; };
        auto dump = [&](int which) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF2; ++nf) *(f32x4*)slab_ptr(which, mf, nf) = acc2[mf][nf];
        };
        auto absorb = [&](int which) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF2; ++nf) acc2[mf][nf] += *(const f32x4*)slab_ptr(which, mf, nf);
        };
        if constexpr (NW == 8) {
            if (wave >= 6) dump(wave - 6);
            __syncthreads();
            if (wave == 2 || wave == 3) absorb(wave - 2);
            __syncthreads();
            if (wave == 4 || wave == 5) dump(wave - 4);
            __syncthreads();
            if (wave < 2) absorb(wave);
            __syncthreads();
        }
        if (wave == 2 || wave == 3) dump(wave - 2);
        __syncthreads();
        if (wave < 2) absorb(wave);
        __syncthreads();
        if (wave < 2) dump(wave);
        __syncthreads();
    }
    CFM_STAMP(5);

    // ================= post norms: y1 -> out_f32, y2 -> out16 / next LDS tile =====================================
    if constexpr (MID) {
        f32x4 v[RPW][VPL];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                v[rr][it] = zero4;
                if (c < D) {
                    const f32x4 y = *(const f32x4*)(slab + r * XS_STRIDE + c) + *(const f32x4*)(slab + RBM * XS_STRIDE + r * XS_STRIDE + c);
                    v[rr][it] = a.alpha * (y + pn_b2[it]) + *(const f32x4*)(xs + r * XS_STRIDE + c);
                }
            }
        }
        if (a.ln1_g) rows_layernorm<RPW, VPL, D>(v, pn_g1, pn_b1, a.eps, lane);
        if (a.out_f32) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    const int64_t grow = row0 + wave * RPW + rr;
                    if (c < D && grow < a.M) *(f32x4*)(a.out_f32 + grow * D + c) = v[rr][it];
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
                        if (c < D && grow < a.M && a.out16) *(u32x2*)((u16*)a.out16 + grow * D + c) = pk;
                    }
                }
        }
        if constexpr (TAIL) __syncthreads();
    }
    CFM_STAMP(6);

    // ================= TAIL: t = xn . Wt^T + bt (GLU optional), 16-bit store ===================================
    if constexpr (TAIL) {
        const int ldo = TGLU ? a.tail_N / 2 : a.tail_N;
#pragma unroll
        for (int s = 0; s < TSTEPS; ++s) {
            const int f = t_frag0(s);
            const int fn = t_frag0(s + 1 < TSTEPS ? s + 1 : s);
            const bool v0ok = f < t_nfrags, v1ok = f + 1 < t_nfrags;
            const f32x4 bb0 = *(const f32x4*)(a.tail_b + t_clamp(f) * 16 + 4 * g);      // issued before this step's MFMAs
            const f32x4 bb1 = *(const f32x4*)(a.tail_b + t_clamp(f + 1) * 16 + 4 * g);
            f32x4 acc[2][2];
            linear_step<HT, KS1, XN_STRIDE>(xn, twp, s + 1 < TSTEPS, t_clamp(fn), t_clamp(fn + 1), twr, acc, g, l15);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int64_t grow = row0 + mf * 16 + l15;
                if (grow >= a.M) continue;
                f32x4 v0 = acc[mf][0] + bb0, v1 = acc[mf][1] + bb1;
                if constexpr (TGLU) {
                    if (v1ok) {                               // (value, gate) fragment pairs: tail_N % 32 == 0
#pragma unroll
                        for (int r = 0; r < 4; ++r) v0[r] *= sigmoidf_(v1[r]);
                        const int64_t o = grow * ldo + (f >> 1) * 16 + 4 * g;
                        *(u32x2*)((u16*)a.tail_out + o) = (u32x2){pack2<HT>(v0.x, v0.y), pack2<HT>(v0.z, v0.w)};
                    }
                } else {
                    const int64_t o = grow * ldo + f * 16 + 4 * g;
                    if (v0ok) *(u32x2*)((u16*)a.tail_out + o) = (u32x2){pack2<HT>(v0.x, v0.y), pack2<HT>(v0.z, v0.w)};
                    if (v1ok) *(u32x2*)((u16*)a.tail_out + o + 16) = (u32x2){pack2<HT>(v1.x, v1.y), pack2<HT>(v1.z, v1.w)};
                }
            }
        }
    }
    CFM_STAMP(7);
}

template <typename HT, int D, int NW, int HS, int FS, int TS, bool TGLU>
int launch_chain(const ChainArgs& a, hipStream_t s, const char* name, double flops) {
    const unsigned grid = (unsigned)((a.M + RBM - 1) / RBM);
    CfmProfScope prof(name, s, flops, (double)a.M * D * 8);
    hipLaunchKernelGGL((cfm_rowchain_kernel<HT, D, NW, HS, FS, TS, TGLU>), dim3(grid), dim3(64 * NW), 0, s, a);
    return cfm_launch_status(name);
}

}  // namespace

extern "C" int cfm_rowchain_supported(int32_t D, int32_t FF) { return (D == 256 && FF == 2048) || (D == 144 && FF == 576); }

extern "C" int cfm_rowchain(const cfm_rowchain_desc* d, cfm_stream_t stream) {
    CFM_CHECK_ARG(d, "cfm_rowchain: null descriptor");
    const bool head = d->head_a != nullptr, mid = d->w1f != nullptr, tail = d->tail_w != nullptr;
    CFM_CHECK_ARG(d->M > 0 && (d->D == 144 || d->D == 256), "cfm_rowchain: D=%d has no instance (144, 256)", d->D);
    CFM_CHECK_ARG(d->w_dtype == CFM_BF16 || d->w_dtype == CFM_F16, "cfm_rowchain: w_dtype must be bf16 or fp16");
    CFM_CHECK_ARG(head || d->x, "cfm_rowchain: need x or a head input");
    CFM_CHECK_ARG(!head || (d->head_w && d->head_b && d->head_res), "cfm_rowchain: head needs weights, bias and residual");
    CFM_CHECK_ARG(!mid || (d->w2f && d->b1 && d->b2 && d->FF > 0 && d->FF % 32 == 0 && d->FF <= 2048), "cfm_rowchain: bad FFN arguments");
    CFM_CHECK_ARG(!tail || (d->tail_b && d->tail_out && d->tail_N % 16 == 0 && d->tail_N > 0 && (!d->tail_glu || d->tail_N % 32 == 0)),
                  "cfm_rowchain: tail needs bias, output and N %% 16 == 0 (GLU: N %% 32 == 0)");
    CFM_CHECK_ARG(!tail || (mid ? d->ln2_g != nullptr : true), "cfm_rowchain: a tail after the FFN takes its input from the second LayerNorm");
    ChainArgs a;
    a.x = d->x; a.head_a = (const u16*)d->head_a; a.head_w = (const u16*)d->head_w; a.head_b = d->head_b; a.head_res = d->head_res;
    a.head_mask = d->head_mask; a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.ln_mask = d->ln_mask; a.w1f = (const u16*)d->w1f; a.w2f = (const u16*)d->w2f;
    a.b1 = d->b1; a.b2 = d->b2; a.ln1_g = d->ln1_g; a.ln1_b = d->ln1_b; a.ln2_g = d->ln2_g; a.ln2_b = d->ln2_b; a.out_f32 = d->out_f32;
    a.out16 = d->out16; a.tail_w = (const u16*)d->tail_w; a.tail_b = d->tail_b; a.tail_out = d->tail_out; a.M = d->M; a.FF = d->FF;
    a.tail_N = d->tail_N; a.out16_dtype = d->w_dtype; a.alpha = d->alpha; a.eps = d->eps;
    a.py0 = d->py0; a.py1 = d->py1; a.pb2 = d->pb2; a.pln_g = d->pln_g; a.pln_b = d->pln_b; a.palpha = d->palpha;
    CFM_CHECK_ARG(!d->py0 || (d->py1 && d->pb2 && !head), "cfm_rowchain: the reduce input needs both slabs and the bias, and no head");
    hipStream_t s = (hipStream_t)stream;
    const bool bf = d->w_dtype == CFM_BF16;
    const int nw = d->D == 256 ? 8 : 4;                    // wavefronts per workgroup (see the header comment)
    const int fsteps = mid ? (d->FF / 32 + nw - 1) / nw : 0;
    const int tsteps = tail ? (d->tail_N / 16 + 2 * nw - 1) / (2 * nw) : 0;
    const double M = (double)d->M;
    const double fl_head = head ? 2.0 * M * d->D * d->D : 0.0, fl_mid = mid ? 4.0 * M * d->D * d->FF : 0.0,
                 fl_tail = tail ? 2.0 * M * d->D * d->tail_N : 0.0;
    const double fl = fl_head + fl_mid + fl_tail;
#define CFM_RC(HT, DD, NWV, HS, FS, TS, GLU, NAME) return launch_chain<HT, DD, NWV, HS, FS, TS, GLU>(a, s, NAME, fl)
    // the three roles of a conformer block, for D = 256 (ff 2048) and D = 144 (ff 576)
    if (d->D == 256) {
        if (!head && mid && tail && !d->tail_glu && fsteps == 8 && tsteps == 3) { if (bf) CFM_RC(BF16, 256, 8, 0, 8, 3, false, "chain_macaron_bf16_d256"); else CFM_RC(F16, 256, 8, 0, 8, 3, false, "chain_macaron_f16_d256"); }
        if (head && !mid && tail && d->tail_glu && tsteps == 2) { if (bf) CFM_RC(BF16, 256, 8, 1, 0, 2, true, "chain_convin_bf16_d256"); else CFM_RC(F16, 256, 8, 1, 0, 2, true, "chain_convin_f16_d256"); }
        if (head && mid && !tail && fsteps == 8) { if (bf) CFM_RC(BF16, 256, 8, 1, 8, 0, false, "chain_final_bf16_d256"); else CFM_RC(F16, 256, 8, 1, 8, 0, false, "chain_final_f16_d256"); }
        if (!head && !mid && tail && !d->tail_glu && tsteps == 3) { if (bf) CFM_RC(BF16, 256, 8, 0, 0, 3, false, "chain_qkv_bf16_d256"); else CFM_RC(F16, 256, 8, 0, 0, 3, false, "chain_qkv_f16_d256"); }
        if (!head && !mid && !tail) { if (bf) CFM_RC(BF16, 256, 8, 0, 0, 0, false, "chain_rows_bf16_d256"); else CFM_RC(F16, 256, 8, 0, 0, 0, false, "chain_rows_f16_d256"); }
    } else {
        if (!head && mid && tail && !d->tail_glu && fsteps == 5 && tsteps == 4) { if (bf) CFM_RC(BF16, 144, 4, 0, 5, 4, false, "chain_macaron_bf16_d144"); else CFM_RC(F16, 144, 4, 0, 5, 4, false, "chain_macaron_f16_d144"); }
        if (head && !mid && tail && d->tail_glu && tsteps == 3) { if (bf) CFM_RC(BF16, 144, 4, 2, 0, 3, true, "chain_convin_bf16_d144"); else CFM_RC(F16, 144, 4, 2, 0, 3, true, "chain_convin_f16_d144"); }
        if (head && mid && !tail && fsteps == 5) { if (bf) CFM_RC(BF16, 144, 4, 2, 5, 0, false, "chain_final_bf16_d144"); else CFM_RC(F16, 144, 4, 2, 5, 0, false, "chain_final_f16_d144"); }
        if (!head && !mid && tail && !d->tail_glu && tsteps == 4) { if (bf) CFM_RC(BF16, 144, 4, 0, 0, 4, false, "chain_qkv_bf16_d144"); else CFM_RC(F16, 144, 4, 0, 0, 4, false, "chain_qkv_f16_d144"); }
        if (!head && !mid && !tail) { if (bf) CFM_RC(BF16, 144, 4, 0, 0, 0, false, "chain_rows_bf16_d144"); else CFM_RC(F16, 144, 4, 0, 0, 0, false, "chain_rows_f16_d144"); }
    }
#undef CFM_RC
    return cfm_fail(CFM_ERR_UNSUPPORTED, "cfm_rowchain: no instance for D=%d FF=%d tail_N=%d head=%d mid=%d tail=%d glu=%d", d->D, d->FF,
                    d->tail_N, (int)head, (int)mid, (int)tail, d->tail_glu);
}
