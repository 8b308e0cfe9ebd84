// ffnpart.hip -- feed-forward block as PARTIAL sums over FF halves on 64-row tiles (gfx950).
//
// EXPERIMENTAL, opt-in (CFM_PARTIAL_FFN=1), slower than the row chains of rowchain.hip -- kept with its tests as the plumbing for
// a partial-sum pipeline.  Motivation, as understood now: with one 32-row tile per
// CU every CU pulls all of W1 and W2 (2 MB) through its own vector-memory path (64 B/clk/CU), and that path bounds the
// feed-forward.  The way down is that each CU pulls a DIFFERENT part of the stream for MORE rows:
//
//     workgroup (tile t, half s):   Y_s[64 rows, D] = act( LN(x_t) . W1[half s]^T + b1 ) . W2[:, half s]^T        (1 MB stream)
//
// 2 x ceil(M/64) workgroups; the two partial slabs Y_0, Y_1 are combined (with bias, alpha, residual and the following
// LayerNorm) by the INPUT STAGE of whichever kernel runs next (this one, or rowchain.hip) -- a launch boundary, so no
// inter-workgroup protocol, and a fixed summation order (bitwise reproducible).
//
// 64 rows per workgroup do not fit the ffn.hip decomposition (a wavefront would hold 64 x 256 accumulators), so the second
// product is split by OUTPUT COLUMNS instead: per 128-column step of FF
//     product 1   wavefront w (of 8) computes H[64, its 32 columns] (rolling-ring weight stream from L2, as ffn.hip),
//     exchange    bias + SiLU, converted to the second product's B-operand fragments and parked in LDS fragment-major
//                 (one barrier per step, two buffers),
//     product 2   wavefront w accumulates Y[64, its 32 output columns] over all 256 hidden columns of the step.
// 8 wavefronts (2 per SIMD, <= 256 VGPRs each): a wavefront pays ~64 clk to issue a 1 KB load, so load throughput scales with
// the number of wavefronts issuing (see PNW below).
// Input stages (template INMODE):  0 = f32 rows;  1 = rows reduced from the previous block's partial slabs
// (x = LN?(res + alpha (Y0 + Y1 + b2)), written back by the half-0 workgroup);  2 = rows produced by a head GEMM on a
// 16-bit tile (pointwise-conv-2 + pad mask + residual), also written back.
#include <string>
#include <type_traits>

#include "cfm_common.h"

struct PartArgs {
    const float* x;             // INMODE 0: rows | 1: residual rows | 2: head residual rows
    const float *py0, *py1;     // INMODE 1: previous partial slabs
    const float* pb2;           //           previous second bias
    const float *pln_g, *pln_b; //           optional LayerNorm on the reduced rows (norm_final)
    const u16* head_a;          // INMODE 2: 16-bit tile source [M,D]
    const u16* head_w;          //           fragment-major [D/16][KS][64][8]
    const float* head_b;
    const uint8_t* head_mask;
    float* x_out;               // INMODE 1/2: the rows this kernel computed, written by the half-0 workgroup (may alias x)
    const float *ln_g, *ln_b;   // LayerNorm feeding the FFN
    const u16 *w1f, *w2f;
    const float* b1;
    float *y0, *y1;             // partial slabs out, f32 [M,D]
    int64_t M;
    int FF;
    float palpha, eps;
};

namespace {

constexpr int PBM = 64;
constexpr int PNW = 8;    // wavefronts per workgroup: a wavefront issues at most one 1 KB load per ~64 clk (16 B/clk), so per-CU
                          // streaming scales with the wavefronts issuing loads, up to the CU's 64 B/clk vector-memory path

template <typename HT, int D, int NSTEPS, int INMODE>
__global__ __launch_bounds__(PNW * 64) void cfm_ffnpart_kernel(const PartArgs a) {
    constexpr int KS1 = (D + 31) / 32;
    constexpr int KP = KS1 * 32;
    constexpr int NF2 = D / 16;              // output column fragments in total
    constexpr int NFW = (NF2 + PNW - 1) / PNW; // ... per wavefront (column split of the second product)
    constexpr int NT = PNW * 64;
    constexpr int MF = PBM / 16;             // 4 row fragments
    constexpr int XS_STRIDE = D + 4;
    constexpr int XN_STRIDE = KP + 8;
    constexpr int VPL = (D + 255) / 256;
    static_assert(D % 16 == 0 && D <= 256, "D % 16 == 0, D <= 256");

    __shared__ __attribute__((aligned(16))) u16 xn[PBM * XN_STRIDE];
    constexpr int HB_U4 = 2 * PNW * MF * 64;                              // [buffer][k-step = producing wavefront][mf][lane]
    constexpr int XS_U4 = INMODE == 2 ? (PBM * XS_STRIDE + 3) / 4 : 0;
    __shared__ u32x4 hx[HB_U4 > XS_U4 ? HB_U4 : XS_U4];                  // hidden-tile exchange; the head stage's f32 rows alias it
    u32x4* hbuf = hx;
    float* xs = (float*)hx;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int half = blockIdx.x & 1;
    const int64_t row0 = (int64_t)(blockIdx.x >> 1) * PBM;

    // ================= weight ring: declared and PRIMED FIRST, so its L2 latency overlaps the whole input stage ===============
    const int fs_half = a.FF / 64;                       // 32-column steps in one half
    const int fs_base = half * fs_half;                  // first 32-column step of this half
    f32x4 acc2[MF][NFW];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) acc2[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4* w1p = (const u32x4*)a.w1f + lane;       // fragment (ffb, kk) at ((ffb*KS1 + kk) * 64 + lane)
    const u32x4* w2p = (const u32x4*)a.w2f + lane;       // fragment (fs, nf2) at ((fs*NF2 + nf2) * 64 + lane)
    u32x4 w1r[KS1];                                      // ring over the wavefront's W1 fragments in consumption order
                                                         // (step, hidden fragment nf, kk): a refill targets the fragment KS1 ahead
    u32x4 w2r[PNW * NFW];                                // [k-step of the workgroup step][own output fragment]
    f32x4 b1r[2];
    auto clamp_fs = [&](int fs) { return fs < fs_half ? fs_base + fs : fs_base + fs_half - 1; };
    auto ocol = [&](int nf) { const int f = wave * NFW + nf; return f < NF2 ? f : NF2 - 1; };
    // (s, wave) -> the wavefront's 32-column step inside the half; step s of the workgroup covers 32-column steps PNW*s ..
    auto load_w1 = [&](int s, int i) { return w1p[((int64_t)(2 * clamp_fs(PNW * s + wave)) * KS1 + i) * 64]; };
    auto load_w2 = [&](int s, int ks, int nf) { return w2p[((int64_t)clamp_fs(PNW * s + ks) * NF2 + ocol(nf)) * 64]; };
#pragma unroll
    for (int i = 0; i < KS1; ++i) w1r[i] = load_w1(0, i);
    b1r[0] = *(const f32x4*)(a.b1 + clamp_fs(wave) * 32 + 4 * g);
    b1r[1] = *(const f32x4*)(a.b1 + clamp_fs(wave) * 32 + 16 + 4 * g);
#pragma unroll
    for (int ks = 0; ks < PNW; ++ks)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) w2r[ks * NFW + nf] = load_w2(0, ks, nf);


    // ================= input stage 2: head GEMM on a 16-bit tile -> xs ===================================================
    if constexpr (INMODE == 2) {
        constexpr int CPRW = KP / 8;
        for (int id = tid; id < PBM * CPRW; id += NT) {
            const int r = id / CPRW, c = id % CPRW;
            int64_t grow = row0 + r;
            grow = grow < a.M ? grow : a.M - 1;
            *(u32x4*)(xn + r * XN_STRIDE + c * 8) = c * 8 < D ? *(const u32x4*)(a.head_a + grow * D + c * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
        __syncthreads();
        const u32x4* wp = (const u32x4*)a.head_w + lane;
        // wavefront w owns output fragments w*NFW .. w*NFW+NFW-1, two per pass
#pragma unroll
        for (int pss = 0; pss < (NFW + 1) / 2; ++pss) {
            const int f = wave * NFW + pss * 2;
            const int fa = f < NF2 ? f : NF2 - 1, fb = f + 1 < NF2 ? f + 1 : NF2 - 1;
            f32x4 acc[MF][2];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) acc[mf][0] = acc[mf][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS1; ++kk) {
                const u32x4 wa = wp[((int64_t)fa * KS1 + kk) * 64], wb = wp[((int64_t)fb * KS1 + kk) * 64];
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    const u32x4 xf = *(const u32x4*)(xn + (mf * 16 + l15) * XN_STRIDE + kk * 32 + 8 * g);
                    acc[mf][0] = HT::mfma(wa, xf, acc[mf][0]);
                    acc[mf][1] = HT::mfma(wb, xf, acc[mf][1]);
                }
            }
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                const bool in_wave = pss * 2 + nf < NFW;
                if (f + nf < NF2 && in_wave) {
                    const int col = (f + nf) * 16 + 4 * g;
                    const f32x4 bb = *(const f32x4*)(a.head_b + col);
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) {
                        const int r = mf * 16 + l15;
                        int64_t grow = row0 + r;
                        grow = grow < a.M ? grow : a.M - 1;
                        f32x4 v = acc[mf][nf] + bb;
                        if (a.head_mask && a.head_mask[grow] == 0) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                        v += *(const f32x4*)(a.x + grow * D + col);
                        *(f32x4*)(xs + r * XS_STRIDE + col) = v;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ================= rows: (reduce | load) -> optional norm_final -> write back -> LN -> xn ===========================
    constexpr int RPW = PBM / PNW;                       // rows per wavefront
    f32x4 rowv[RPW][VPL];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {                   // all global row loads in flight together (one exposed latency, not RPW)
        const int r = wave * RPW + rr;
        int64_t grow = row0 + r;
        grow = grow < a.M ? grow : a.M - 1;
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            const int c = (lane + 64 * it) * 4;
            rowv[rr][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                if constexpr (INMODE == 2) {
                    rowv[rr][it] = *(const f32x4*)(xs + r * XS_STRIDE + c);
                } else if constexpr (INMODE == 1) {
                    const f32x4 y = *(const f32x4*)(a.py0 + grow * D + c) + *(const f32x4*)(a.py1 + grow * D + c);
                    rowv[rr][it] = *(const f32x4*)(a.x + grow * D + c) + a.palpha * (y + *(const f32x4*)(a.pb2 + c));
                } else {
                    rowv[rr][it] = *(const f32x4*)(a.x + grow * D + c);
                }
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        int64_t grow = row0 + r;
        const bool live = grow < a.M;
        grow = live ? grow : a.M - 1;
        f32x4 v[VPL];
#pragma unroll
        for (int it = 0; it < VPL; ++it) v[it] = rowv[rr][it];
        auto norm = [&](const float* gam, const float* bet) {
            float s = 0.f;
#pragma unroll
            for (int it = 0; it < VPL; ++it)
                if ((lane + 64 * it) * 4 < D) s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
            const float mean = wave_sum(s) / (float)D;
            float q = 0.f;
#pragma unroll
            for (int it = 0; it < VPL; ++it)
                if ((lane + 64 * it) * 4 < D) {
                    const f32x4 d = v[it] - mean;
                    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + a.eps);
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < D) v[it] = (v[it] - mean) * rstd * *(const f32x4*)(gam + c) + *(const f32x4*)(bet + c);
            }
        };
        if constexpr (INMODE == 1) {
            if (a.pln_g) norm(a.pln_g, a.pln_b);
        }
        if constexpr (INMODE != 0) {
            if (a.x_out && half == 0 && live) {
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    if (c < D) *(f32x4*)(a.x_out + grow * D + c) = v[it];
                }
            }
        }
        norm(a.ln_g, a.ln_b);
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            const int c = (lane + 64 * it) * 4;
            if (c < KP) {
                const f32x4 o = c < D ? v[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
                *(u32x2*)(xn + r * XN_STRIDE + c) = (u32x2){pack2<HT>(o.x, o.y), pack2<HT>(o.z, o.w)};
            }
        }
    }
    __syncthreads();

    // ================= main loop over this half of FF (ring primed at kernel entry) =============================================
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
        const int sn = s + 1 < NSTEPS ? s + 1 : s;       // refill source (last step re-fetches itself: result unused)
        const bool valid = PNW * s + wave < fs_half;     // this wavefront's 32 hidden columns exist
        // ---- product 1: H[64, this wavefront's 32 columns], one 16-column fragment after the other ---------------------------
        f32x4 acc1[MF][2];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc1[mf][0] = acc1[mf][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            u32x4 xf[2][MF];                                 // double-buffered: the reads of group i+1 are issued before the MFMAs of group i
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) xf[0][mf] = *(const u32x4*)(xn + (mf * 16 + l15) * XN_STRIDE + 8 * g);
#pragma unroll
            for (int grp = 0; grp < 2 * KS1; ++grp) {
                const int nf = grp / KS1, kk = grp % KS1;
                const int nkk = (grp + 1) % KS1;
                if (grp + 1 < 2 * KS1) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) xf[(grp + 1) & 1][mf] = *(const u32x4*)(xn + (mf * 16 + l15) * XN_STRIDE + nkk * 32 + 8 * g);
                }
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) acc1[mf][nf] = HT::mfma(w1r[kk], xf[grp & 1][mf], acc1[mf][nf]);
                w1r[kk] = nf == 0 ? load_w1(s, KS1 + kk) : load_w1(sn, kk);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const f32x4 bb0 = b1r[0], bb1 = b1r[1];
        b1r[0] = *(const f32x4*)(a.b1 + clamp_fs(PNW * sn + wave) * 32 + 4 * g);
        b1r[1] = *(const f32x4*)(a.b1 + clamp_fs(PNW * sn + wave) * 32 + 16 + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
        // ---- bias + SiLU -> B-operand fragments -> LDS (fragment-major: a plain 16-byte store per lane) ------------------------
        u32x4* hb = hbuf + (s & 1) * (PNW * MF * 64);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            f32x4 h0 = acc1[mf][0] + bb0, h1 = acc1[mf][1] + bb1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                h0[r] = siluf_(h0[r]);
                h1[r] = siluf_(h1[r]);
            }
            u32x4 hf = pack8<HT>(h0, h1);
            if (!valid) hf = (u32x4){0u, 0u, 0u, 0u};
            hb[(wave * MF + mf) * 64 + lane] = hf;
        }
        __syncthreads();
        // ---- product 2: Y[64, this wavefront's output columns] += H[64, 128] . W2 ----------------------------------------------
        {
            u32x4 hf[2][MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) hf[0][mf] = hb[(0 * MF + mf) * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < PNW; ++ks) {
                if (ks + 1 < PNW) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) hf[(ks + 1) & 1][mf] = hb[((ks + 1) * MF + mf) * 64 + lane];
                }
#pragma unroll
                for (int nf = 0; nf < NFW; ++nf) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) acc2[mf][nf] = HT::mfma(w2r[ks * NFW + nf], hf[ks & 1][mf], acc2[mf][nf]);
                    w2r[ks * NFW + nf] = load_w2(sn, ks, nf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // ================= partial slab out ===============================================================================================
    float* ys = half ? a.y1 : a.y0;
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
        const int f = wave * NFW + nf;
        if (f >= NF2) continue;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int64_t grow = row0 + mf * 16 + l15;
            if (grow < a.M) *(f32x4*)(ys + grow * D + f * 16 + 4 * g) = acc2[mf][nf];
        }
    }
}

template <typename HT, int D, int NS>
int launch_part(const PartArgs& a, int inmode, hipStream_t s, const char* name) {
    const unsigned grid = 2u * (unsigned)((a.M + PBM - 1) / PBM);
    const double flops = 4.0 * (double)a.M * D * a.FF + (inmode == 2 ? 2.0 * (double)a.M * D * D : 0.0);
    CfmProfScope prof(name, s, flops, (double)a.M * D * 16);
    if (inmode == 0) CFM_LAUNCH((cfm_ffnpart_kernel<HT, D, NS, 0>), dim3(grid), dim3(PNW * 64), 0, s, a);
    else if (inmode == 1) CFM_LAUNCH((cfm_ffnpart_kernel<HT, D, NS, 1>), dim3(grid), dim3(PNW * 64), 0, s, a);
    else CFM_LAUNCH((cfm_ffnpart_kernel<HT, D, NS, 2>), dim3(grid), dim3(PNW * 64), 0, s, a);
    return cfm_launch_status(name);
}

}  // namespace

extern "C" int cfm_ffn_partial_supported(int32_t D, int32_t FF) { return (D == 256 && FF == 2048) || (D == 144 && FF == 576); }

extern "C" int cfm_ffn_partial(const cfm_ffn_partial_desc* d, cfm_stream_t stream) {
    CFM_CHECK_ARG(d && d->x && d->ln_g && d->ln_b && d->w1f && d->w2f && d->b1 && d->y0 && d->y1, "cfm_ffn_partial: null pointer");
    CFM_CHECK_ARG(d->M > 0 && cfm_ffn_partial_supported(d->D, d->FF), "cfm_ffn_partial: no instance for D=%d FF=%d", d->D, d->FF);
    CFM_CHECK_ARG(d->w_dtype == CFM_BF16 || d->w_dtype == CFM_F16, "cfm_ffn_partial: w_dtype must be bf16 or fp16");
    const int inmode = d->head_a ? 2 : (d->py0 ? 1 : 0);
    CFM_CHECK_ARG(inmode != 1 || (d->py1 && d->pb2), "cfm_ffn_partial: the reduce input needs both slabs and the bias");
    CFM_CHECK_ARG(inmode != 2 || (d->head_w && d->head_b), "cfm_ffn_partial: the head input needs weights and bias");
    CFM_CHECK_ARG((d->pln_g == nullptr) == (d->pln_b == nullptr), "cfm_ffn_partial: LayerNorm gain/bias must come in pairs");
    CFM_CHECK_ARG(inmode == 0 || d->x_out == nullptr || d->x_out != d->x, "cfm_ffn_partial: x_out must not alias x (two workgroups read each row tile)");
    PartArgs a;
    a.x = d->x; a.py0 = d->py0; a.py1 = d->py1; a.pb2 = d->pb2; a.pln_g = d->pln_g; a.pln_b = d->pln_b; a.head_a = (const u16*)d->head_a;
    a.head_w = (const u16*)d->head_w; a.head_b = d->head_b; a.head_mask = d->head_mask; a.x_out = d->x_out; a.ln_g = d->ln_g; a.ln_b = d->ln_b;
    a.w1f = (const u16*)d->w1f; a.w2f = (const u16*)d->w2f; a.b1 = d->b1; a.y0 = d->y0; a.y1 = d->y1; a.M = d->M; a.FF = d->FF;
    a.palpha = d->palpha; a.eps = d->eps;
    hipStream_t s = (hipStream_t)stream;
    const bool bf = d->w_dtype == CFM_BF16;
    // workgroup steps of PNW*32 = 256 hidden columns per half: FF=2048 -> 4; FF=576 -> 288 per half = 9 slices of 32 -> 2 (partly masked)
    if (d->D == 256) return bf ? launch_part<BF16, 256, 4>(a, inmode, s, "ffn_partial_bf16_d256") : launch_part<F16, 256, 4>(a, inmode, s, "ffn_partial_f16_d256");
    return bf ? launch_part<BF16, 144, 2>(a, inmode, s, "ffn_partial_bf16_d144") : launch_part<F16, 144, 2>(a, inmode, s, "ffn_partial_f16_d144");
}
