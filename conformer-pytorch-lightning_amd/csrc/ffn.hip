// ffn.hip -- fused position-wise feed-forward block for gfx950:
//
//     y  = x + alpha * ( W2 . act( W1 . LN(x) + b1 ) + b2 )           (encoder_layer.py:56-58 / :67-69, feedforward.py:16-21)
//     y1 = LN1(y)  (optional, e.g. norm_final)      -> out_f32
//     y2 = LN2(y1) (optional, next GEMM's operand)  -> out16
//
// in ONE launch.  The [M, FF] hidden activation never exists in memory: it goes from the accumulators of the first
// product straight into the B operand of the second.
//
// Why this shape.  At config 2 there are M = 7968 rows for 256 CUs: ~31 rows per CU.  Two separate GEMMs spend their time
// in per-launch prologues/epilogues and in writing + re-reading the 32 MB hidden tensor.  Here a workgroup (4 wavefronts,
// one per SIMD) owns a 32-row tile for the whole block:
//   prologue   the tile's rows are loaded once (f32), LayerNorm'ed with wavefront shuffles, rounded to 16 bit and parked in
//              LDS (read-only from then on: no barrier in the main loop) as the B operand of the first product.
//   main loop  NO LDS and NO BARRIERS.  FF is cut into steps of 128 columns, 32 per wavefront.  Per step a wavefront
//              streams its own 16 KB of W1 and 16 KB of W2 directly global(L2) -> VGPR.  The host packs both matrices
//              FRAGMENT-MAJOR, so every wavefront-load is 1 KB contiguous (lane*16 B) and already in MFMA A-operand
//              order; W2's k order inside each 32-block is permuted on the host to the order in which the first
//              product's accumulators hold the hidden values (4 consecutive columns per lane per fragment), so
//              SiLU(acc1) is converted in registers and fed to the second MFMA with no shuffle.  Each wavefront
//              accumulates a PARTIAL y over its own FF columns for all 256 outputs (128 accumulator registers).
//              Each fragment's registers are refilled with the next step's fragment right after the MFMAs that consume
//              them (rolling ring: a full step of lead, one register set); the step loop is fully unrolled so that the
//              compiler's vmcnt bookkeeping is exact (it drains everything at a loop header).
//   epilogue   the 4 partials are summed through LDS in a FIXED order ((w0+w2)+(w1+w3): bitwise reproducible), then each
//              wavefront owns 8 complete rows: bias, alpha, residual, up to two chained LayerNorms, full-row 16-byte
//              stores (1 KB contiguous per row).
// The kernel is bound by the L2 -> CU weight stream (2 MB per workgroup), not by MFMA issue: see DESIGN.md.
#include <string>
#include <type_traits>

#include "cfm_common.h"

struct FfnArgs {
    const float* x;
    const float *ln_g, *ln_b;
    const u16 *w1f, *w2f;
    const float *b1, *b2;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float* out_f32;
    void* out16;
    int64_t M;
    int FF;
    int act, out16_dtype, add_x;
    float alpha, eps;
    // TRAIN (cfm_ffn_train_forward): what the backward needs is written on the way -- the LayerNorm output (operand of dW1), the pre-activation
    // (silu'), the hidden activation after its dropout (operand of dW2) -- and both dropout sites of feedforward.py:19 / encoder_layer.py:58,69 are
    // applied here with the element indices the unfused products use (row * FF + column, row * D + column: the backward regenerates the masks)
    u16 *xn_out, *z_out, *h_out;
    CfmDrop drop_h, drop_o;
};

namespace {

constexpr int FBM = 32;  // rows per workgroup

template <typename HT, int D, int NSTEPS, int ACT, bool TRAIN = false>
__global__ __launch_bounds__(256) void cfm_ffn_kernel(const FfnArgs a) {
    constexpr int KS1 = (D + 31) / 32;   // k-steps of the first product (K zero-padded to a multiple of 32)
    constexpr int KP = KS1 * 32;
    constexpr int NF2 = D / 16;          // 16-column output fragments of the second product
    constexpr int MF = FBM / 16;         // 16-row fragments per tile
    constexpr int XS_STRIDE = D + 4;     // f32 row stride of the x tile / reduction slabs (+16 B: conflict-free b128)
    constexpr int XN_STRIDE = KP + 8;    // 16-bit row stride of the normalised tile
    constexpr int VPL = (D + 255) / 256; // f32x4 per lane per row in the row-wise phases
    static_assert(D % 16 == 0 && D <= 256, "fused FFN supports D % 16 == 0, D <= 256");

    __shared__ __attribute__((aligned(16))) float xs[FBM * XS_STRIDE];        // residual rows, later the final rows
    __shared__ __attribute__((aligned(16))) float slab[2 * FBM * XS_STRIDE];  // cross-wavefront reduction
    __shared__ __attribute__((aligned(16))) u16 xn[FBM * XN_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches, not exec masks
    const int g = lane >> 4, l15 = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * FBM;

    // ---- prologue: load 8 rows per wavefront, LayerNorm, keep f32 rows (residual) and 16-bit rows in LDS ----------------
#pragma unroll
    for (int rr = 0; rr < FBM / 4; ++rr) {
        const int r = wave * (FBM / 4) + rr;
        int64_t grow = row0 + r;
        grow = grow < a.M ? grow : a.M - 1;
        f32x4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            const int c = (lane + 64 * it) * 4;
            v[it] = c < D ? *(const f32x4*)(a.x + grow * D + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                *(f32x4*)(xs + r * XS_STRIDE + c) = v[it];
                s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
            }
        }
        if (a.ln_g) {
            const float mean = wave_sum(s) / (float)D;
            float q = 0.f;
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < D) {
                    const f32x4 d = v[it] - mean;
                    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
            }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + a.eps);
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < D) v[it] = (v[it] - mean) * rstd * *(const f32x4*)(a.ln_g + c) + *(const f32x4*)(a.ln_b + c);
            }
        }
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            const int c = (lane + 64 * it) * 4;
            if (c < KP) {
                const f32x4 o = c < D ? v[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
                const u32x2 pk = (u32x2){pack2<HT>(o.x, o.y), pack2<HT>(o.z, o.w)};
                *(u32x2*)(xn + r * XN_STRIDE + c) = pk;
                if constexpr (TRAIN) {
                    if (c < D && row0 + r < a.M) *(u32x2*)(a.xn_out + (row0 + r) * D + c) = pk;
                }
            }
        }
    }
    __syncthreads();
    // B-operand fragments of the normalised tile (row mf*16 + l15, k = kk*32 + 8g..+8) are re-read from LDS at every step:
    // 16 conflict-free ds_read_b128 per step are cheap, and the 64 VGPRs they would pin are spent on the weight double
    // buffer below instead.
    auto xfrag = [&](int mf, int kk) { return *(const u32x4*)(xn + (mf * 16 + l15) * XN_STRIDE + kk * 32 + 8 * g); };

    // ---- main loop: barrier-free weight streaming --------------------------------------------------------------------
    const int nsteps_total = a.FF / 32;                 // 32 FF columns per wavefront-step
    f32x4 acc2[MF][NF2];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF2; ++nf) acc2[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const u32x4* w1p = (const u32x4*)a.w1f + lane;      // fragment (ffb, kk) at ((ffb*KS1 + kk) * 64 + lane)
    const u32x4* w2p = (const u32x4*)a.w2f + lane;      // fragment (fs, nf2) at ((fs*NF2 + nf2) * 64 + lane)
    // Rolling ring, ONE register set per matrix: the registers of a weight fragment are refilled with the NEXT step's
    // fragment right after the MFMAs that consume them, so every load leads its use by a full step (~64 MFMAs) at no
    // extra register cost.  (Refilling a whole matrix after its product gave only half a step of lead and two L2-latency
    // stalls per step: 44 us per call; a full double buffer needs 512+ VGPRs and spills.)
    u32x4 w1r[2 * KS1], w2r[NF2];
    f32x4 b1r[2];
    auto w1_addr = [&](int fs, int i) { return w1p + ((int64_t)(2 * fs) * KS1 + i) * 64; };
    auto w2_addr = [&](int fs, int i) { return w2p + ((int64_t)fs * NF2 + i) * 64; };
    // NO BRANCH anywhere in the step sequence: with a per-step exit branch every step is its own basic block and LLVM's
    // sink pass moves the refill loads into the block of their consumer (the next step), i.e. right in front of their
    // use.  So every wavefront runs exactly NSTEPS straight-line steps; a step beyond FF is neutralised by zeroing its
    // hidden fragment, and all addresses are clamped to the last valid step.
    // Every workgroup streams the SAME 2 MB of weights.  Walking them in the same order puts all 250 CUs on the same few
    // L2 channels at any instant, so each workgroup starts at its own phase (blockIdx.x) of the step sequence.
    const int rot = (int)(blockIdx.x % NSTEPS);
    auto step_of = [&](int s) { int q = s + rot; q = q >= NSTEPS ? q - NSTEPS : q; return q * 4 + wave; };
    auto step = [&](int s) {
        const int fs_raw = step_of(s);
        const bool valid = fs_raw < nsteps_total;       // wave-uniform
        const int last = nsteps_total - 1;
        const int nx_raw = step_of(s + 1 < NSTEPS ? s + 1 : s);
        const int nx = nx_raw < nsteps_total ? nx_raw : last;
        // first product: acc1[mf][nf] = H[row mf*16+l15][ff = fs*32 + nf*16 + 4g + r]
        f32x4 acc1[MF][2];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) acc1[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        u32x4 xf[MF][KS1];                               // all LDS fragment reads of the step up front: their latency
#pragma unroll                                          // overlaps the tail of the previous step's second product
        for (int kk = 0; kk < KS1; ++kk)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) xf[mf][kk] = xfrag(mf, kk);
#pragma unroll
        for (int kk = 0; kk < KS1; ++kk) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) acc1[mf][nf] = HT::mfma(w1r[nf * KS1 + kk], xf[mf][kk], acc1[mf][nf]);
            w1r[kk] = *w1_addr(nx, kk);                  // unconditional: a branch here makes the compiler's vmcnt
            w1r[KS1 + kk] = *w1_addr(nx, KS1 + kk);      // bookkeeping assume the load was NOT issued and drain to 0
            __builtin_amdgcn_sched_barrier(0);           // ...and pinned (a VMEM-only fence, mask 0x38F, is not enough: the
                                                         // MFMAs then migrate and the waits collapse to vmcnt(2..5))
        }
        const f32x4 bb0 = b1r[0], bb1 = b1r[1];
        b1r[0] = *(const f32x4*)(a.b1 + nx * 32 + 4 * g);
        b1r[1] = *(const f32x4*)(a.b1 + nx * 32 + 16 + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
        // bias + activation, convert to the second product's B operand (k order = accumulator order, see packing)
        u32x4 hf[MF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            f32x4 h0 = acc1[mf][0] + bb0, h1 = acc1[mf][1] + bb1;
            const int64_t trow = row0 + mf * 16 + l15;
            const int tcol = fs_raw * 32 + 4 * g;         // this lane's 4 columns of fragment 0; fragment 1 sits 16 columns on
            if constexpr (TRAIN) {
                if (valid && trow < a.M) {
                    const u32x4 zp = pack8<HT>(h0, h1);
                    *(u32x2*)(a.z_out + trow * a.FF + tcol) = (u32x2){zp.x, zp.y};
                    *(u32x2*)(a.z_out + trow * a.FF + tcol + 16) = (u32x2){zp.z, zp.w};
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                h0[r] = ACT == CFM_ACT_SILU ? siluf_(h0[r]) : fmaxf(h0[r], 0.f);
                h1[r] = ACT == CFM_ACT_SILU ? siluf_(h1[r]) : fmaxf(h1[r], 0.f);
            }
            if constexpr (TRAIN) {
                if (a.drop_h.thresh) {
                    const unsigned e0 = (unsigned)trow * (unsigned)a.FF + (unsigned)tcol;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        h0[r] = cfm_drop(a.drop_h, e0 + r, h0[r]);
                        h1[r] = cfm_drop(a.drop_h, e0 + 16 + r, h1[r]);
                    }
                }
            }
            hf[mf] = pack8<HT>(h0, h1);
            if (!valid) hf[mf] = (u32x4){0u, 0u, 0u, 0u};
            if constexpr (TRAIN) {
                if (valid && trow < a.M) {
                    *(u32x2*)(a.h_out + trow * a.FF + tcol) = (u32x2){hf[mf].x, hf[mf].y};
                    *(u32x2*)(a.h_out + trow * a.FF + tcol + 16) = (u32x2){hf[mf].z, hf[mf].w};
                }
            }
        }
#pragma unroll
        for (int nf = 0; nf < NF2; ++nf) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) acc2[mf][nf] = HT::mfma(w2r[nf], hf[mf], acc2[mf][nf]);
            w2r[nf] = *w2_addr(nx, nf);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        const int f00 = step_of(0);
        const int f0 = f00 < nsteps_total ? f00 : nsteps_total - 1;
#pragma unroll
        for (int i = 0; i < 2 * KS1; ++i) w1r[i] = *w1_addr(f0, i);
        b1r[0] = *(const f32x4*)(a.b1 + f0 * 32 + 4 * g);
        b1r[1] = *(const f32x4*)(a.b1 + f0 * 32 + 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < NF2; ++i) w2r[i] = *w2_addr(f0, i);
    }
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) step(s);

    // ---- cross-wavefront reduction, fixed order ((w0 + w2) + (w1 + w3)) --------------------------------------------------
    // acc2[mf][nf][r] = partial y[row mf*16 + l15][col nf*16 + 4g + r]
    auto slab_ptr = [&](int which, int mf, int nf) { return slab + which * FBM * XS_STRIDE + (mf * 16 + l15) * XS_STRIDE + nf * 16 + 4 * g; };
    if (wave >= 2) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF2; ++nf) *(f32x4*)slab_ptr(wave - 2, mf, nf) = acc2[mf][nf];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF2; ++nf) acc2[mf][nf] += *(const f32x4*)slab_ptr(wave, mf, nf);
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF2; ++nf) *(f32x4*)slab_ptr(wave, mf, nf) = acc2[mf][nf];
    }
    __syncthreads();

    // ---- epilogue: each wavefront finishes 8 complete rows ------------------------------------------------------------------
#pragma unroll
    for (int rr = 0; rr < FBM / 4; ++rr) {
        const int r = wave * (FBM / 4) + rr;
        const int64_t grow = row0 + r;
        f32x4 v[VPL];
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            const int c = (lane + 64 * it) * 4;
            v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                const f32x4 y = *(const f32x4*)(slab + r * XS_STRIDE + c) + *(const f32x4*)(slab + FBM * XS_STRIDE + r * XS_STRIDE + c);
                f32x4 br = y + *(const f32x4*)(a.b2 + c);
                if constexpr (TRAIN) {
                    if (a.drop_o.thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) br[e] = cfm_drop(a.drop_o, (unsigned)grow * (unsigned)D + (unsigned)(c + e), br[e]);
                    }
                }
                v[it] = a.alpha * br;
                if (a.add_x) v[it] += *(const f32x4*)(xs + r * XS_STRIDE + c);
            }
        }
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
        if (a.ln1_g) norm(a.ln1_g, a.ln1_b);
        if (grow < a.M && a.out_f32) {
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < D) *(f32x4*)(a.out_f32 + grow * D + c) = v[it];
            }
        }
        if (a.out16) {
            if (a.ln2_g) norm(a.ln2_g, a.ln2_b);
            if (grow < a.M) {
#pragma unroll
                for (int it = 0; it < VPL; ++it) {
                    const int c = (lane + 64 * it) * 4;
                    if (c < D) {
                        if (a.out16_dtype == CFM_BF16)
                            *(u32x2*)((u16*)a.out16 + grow * D + c) = (u32x2){pack2<BF16>(v[it].x, v[it].y), pack2<BF16>(v[it].z, v[it].w)};
                        else
                            *(u32x2*)((u16*)a.out16 + grow * D + c) = (u32x2){pack2<F16>(v[it].x, v[it].y), pack2<F16>(v[it].z, v[it].w)};
                    }
                }
            }
        }
    }
}

template <typename HT, int D>
int launch_ffn(const FfnArgs& a, hipStream_t s, const char* name) {
    const int steps = (a.FF / 32 + 3) / 4;               // steps per wavefront
    const unsigned grid = (unsigned)((a.M + FBM - 1) / FBM);
    const double flops = 4.0 * (double)a.M * D * a.FF;
    const double bytes = (double)a.M * D * 8 + 4.0 * D * a.FF;
    CfmProfScope prof(name, s, flops, bytes);
    const bool silu = a.act == CFM_ACT_SILU;
#define CFM_FFN_LAUNCH(NS)                                                                                      \
    do {                                                                                                        \
        if (silu) CFM_LAUNCH((cfm_ffn_kernel<HT, D, NS, CFM_ACT_SILU>), dim3(grid), dim3(256), 0, s, a); \
        else CFM_LAUNCH((cfm_ffn_kernel<HT, D, NS, CFM_ACT_RELU>), dim3(grid), dim3(256), 0, s, a);      \
    } while (0)
    // instances exist for 4, 5, 8 and 16 steps per wavefront; other counts run the next larger one (extra steps masked)
    if (steps <= 4) CFM_FFN_LAUNCH(4);
    else if (steps == 5) CFM_FFN_LAUNCH(5);
    else if (steps <= 8) CFM_FFN_LAUNCH(8);
    else if (steps <= 16) CFM_FFN_LAUNCH(16);
    else return cfm_fail(CFM_ERR_UNSUPPORTED, "cfm_ffn_fused: FF=%d needs more than 16 steps per wavefront (max FF 2048)", a.FF);
#undef CFM_FFN_LAUNCH
    return cfm_launch_status(name);
}

template <typename HT>
int launch_ffn_train(const FfnArgs& a, hipStream_t s, const char* name) {
    constexpr int D = 256;
    const int steps = (a.FF / 32 + 3) / 4;
    const unsigned grid = (unsigned)((a.M + FBM - 1) / FBM);
    CfmProfScope prof(name, s, 4.0 * (double)a.M * D * a.FF, (double)a.M * (D * 10.0 + 4.0 * a.FF) + 4.0 * D * a.FF);
    if (steps <= 8) CFM_LAUNCH((cfm_ffn_kernel<HT, D, 8, CFM_ACT_SILU, true>), dim3(grid), dim3(256), 0, s, a);
    else if (steps <= 16) CFM_LAUNCH((cfm_ffn_kernel<HT, D, 16, CFM_ACT_SILU, true>), dim3(grid), dim3(256), 0, s, a);
    else return cfm_fail(CFM_ERR_UNSUPPORTED, "cfm_ffn_train_forward: FF=%d needs more than 16 steps per wavefront (max FF 2048)", a.FF);
    return cfm_launch_status(name);
}

}  // namespace

extern "C" int cfm_ffn_train_supported(int32_t D, int32_t FF) { return D == 256 && FF > 0 && FF % 128 == 0 && FF <= 2048; }

extern "C" int cfm_ffn_train_forward(const cfm_ffn_train_desc* d, cfm_stream_t stream) {
    CFM_CHECK_ARG(d && d->x && d->ln_g && d->ln_b && d->w1f && d->w2f && d->b1 && d->b2 && d->y && d->xn_out && d->z_out && d->h_out, "cfm_ffn_train_forward: null pointer");
    CFM_CHECK_ARG(cfm_ffn_train_supported(d->D, d->FF), "cfm_ffn_train_forward: D=%d FF=%d has no instance (D = 256, FF %% 128 == 0, FF <= 2048)", d->D, d->FF);
    CFM_CHECK_ARG(d->M > 0 && d->M * (int64_t)d->FF < ((int64_t)1 << 32), "cfm_ffn_train_forward: fewer than 2^32 hidden elements (dropout index)");
    CFM_CHECK_ARG(d->w_dtype == CFM_BF16 || d->w_dtype == CFM_F16, "cfm_ffn_train_forward: w_dtype must be bf16 or fp16");
    CFM_CHECK_ARG(d->p_hidden >= 0.f && d->p_hidden < 1.f && d->p_out >= 0.f && d->p_out < 1.f, "cfm_ffn_train_forward: dropout probabilities in [0, 1)");
    FfnArgs a = {};
    a.x = d->x; a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.w1f = (const u16*)d->w1f; a.w2f = (const u16*)d->w2f; a.b1 = d->b1; a.b2 = d->b2;
    a.out_f32 = d->y; a.M = d->M; a.FF = d->FF; a.act = CFM_ACT_SILU; a.add_x = 1; a.alpha = d->alpha; a.eps = d->eps;
    a.xn_out = (u16*)d->xn_out; a.z_out = (u16*)d->z_out; a.h_out = (u16*)d->h_out;
    a.drop_h = cfm_make_drop(d->p_hidden, d->seed_hidden); a.drop_o = cfm_make_drop(d->p_out, d->seed_out);
    hipStream_t s = (hipStream_t)stream;
    return d->w_dtype == CFM_BF16 ? launch_ffn_train<BF16>(a, s, "ffn_train_fwd_bf16_d256") : launch_ffn_train<F16>(a, s, "ffn_train_fwd_f16_d256");
}

// Fragment-major weight packs of the fused feed-forward for a whole stack in one launch (cfm/packing.py pack_ffn_fragments on the device):
//   w1f[((ffb*KS1 + kk)*64 + lane)*8 + j] = W1[ffb*16 + (lane&15)][kk*32 + 8*(lane>>4) + j]
//   w2f[((fs*NF2 + nf)*64 + lane)*8 + j]  = W2[nf*16 + (lane&15)][fs*32 + (j<4 ? 0 : 16) + 4*(lane>>4) + (j&3)]
// job = 4 x int64: W1 f32 [FF, D], W2 f32 [D, FF], w1f, w2f (16-bit); D % 32 == 0.
namespace {
template <typename HT>
__global__ void cfm_pack_ffn_frag_kernel(const int64_t* __restrict__ jobs, int D, int FF) {
    const int64_t* job = jobs + (int64_t)blockIdx.y * 4;
    const float* W1 = (const float*)job[0];
    const float* W2 = (const float*)job[1];
    u16* w1f = (u16*)job[2];
    u16* w2f = (u16*)job[3];
    const int KS1 = D / 32, NF2 = D / 16;
    const int64_t n1 = (int64_t)(FF / 16) * KS1 * 64;       // 16-byte pieces of w1f; w2f has (FF/32) * NF2 * 64 = the same count
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < 2 * n1; id += (int64_t)gridDim.x * blockDim.x) {
        const bool second = id >= n1;
        const int64_t q = second ? id - n1 : id;
        const int lane = (int)(q & 63);
        f32x4 a, b;
        if (!second) {
            const int kk = (int)((q >> 6) % KS1), ffb = (int)((q >> 6) / KS1);
            const float* src = W1 + (int64_t)(ffb * 16 + (lane & 15)) * D + kk * 32 + 8 * (lane >> 4);
            a = *(const f32x4*)src;
            b = *(const f32x4*)(src + 4);
            *(u32x4*)(w1f + q * 8) = pack8<HT>(a, b);
        } else {
            const int nf = (int)((q >> 6) % NF2), fs = (int)((q >> 6) / NF2);
            const float* src = W2 + (int64_t)(nf * 16 + (lane & 15)) * FF + fs * 32 + 4 * (lane >> 4);
            a = *(const f32x4*)src;
            b = *(const f32x4*)(src + 16);
            *(u32x4*)(w2f + q * 8) = pack8<HT>(a, b);
        }
    }
}
}  // namespace

extern "C" int cfm_pack_ffn_fragments(const int64_t* jobs_dev, int32_t n_jobs, int32_t D, int32_t FF, int32_t w_dtype, cfm_stream_t stream) {
    CFM_CHECK_ARG(jobs_dev && n_jobs > 0 && n_jobs <= 65535 && D > 0 && D % 32 == 0 && FF > 0 && FF % 32 == 0, "cfm_pack_ffn_fragments: bad arguments (D, FF multiples of 32)");
    CFM_CHECK_ARG(w_dtype == CFM_BF16 || w_dtype == CFM_F16, "cfm_pack_ffn_fragments: 16-bit destination type");
    hipStream_t s = (hipStream_t)stream;
    const int64_t pieces = 2 * (int64_t)(FF / 16) * (D / 32) * 64;
    CfmProfScope prof("pack_ffn_fragments", s, 0.0, (double)n_jobs * pieces * 48);
    const dim3 grid((unsigned)((pieces + 255) / 256), (unsigned)n_jobs);
    if (w_dtype == CFM_BF16) CFM_LAUNCH((cfm_pack_ffn_frag_kernel<BF16>), grid, dim3(256), 0, s, jobs_dev, D, FF);
    else CFM_LAUNCH((cfm_pack_ffn_frag_kernel<F16>), grid, dim3(256), 0, s, jobs_dev, D, FF);
    return cfm_launch_status("cfm_pack_ffn_fragments");
}

extern "C" int cfm_ffn_fused(const cfm_ffn_desc* d, cfm_stream_t stream) {
    CFM_CHECK_ARG(d && d->x && d->w1f && d->w2f && d->b1 && d->b2, "cfm_ffn_fused: null pointer");
    CFM_CHECK_ARG(d->out_f32 || d->out16, "cfm_ffn_fused: no output requested");
    CFM_CHECK_ARG(d->M > 0 && d->FF > 0 && d->FF % 32 == 0, "cfm_ffn_fused: need FF %% 32 == 0 (M=%lld FF=%d)", (long long)d->M, d->FF);
    CFM_CHECK_ARG(d->w_dtype == CFM_BF16 || d->w_dtype == CFM_F16, "cfm_ffn_fused: w_dtype must be bf16 or fp16");
    CFM_CHECK_ARG(d->act == CFM_ACT_SILU || d->act == CFM_ACT_RELU, "cfm_ffn_fused: activation must be SiLU or ReLU");
    CFM_CHECK_ARG((d->ln_g == nullptr) == (d->ln_b == nullptr) && (d->ln1_g == nullptr) == (d->ln1_b == nullptr) &&
                      (d->ln2_g == nullptr) == (d->ln2_b == nullptr), "cfm_ffn_fused: LayerNorm gain/bias must come in pairs");
    CFM_CHECK_ARG(!d->ln2_g || d->out16, "cfm_ffn_fused: the second LayerNorm needs out16");
    CFM_CHECK_ARG(!d->out16 || d->out16_dtype == CFM_BF16 || d->out16_dtype == CFM_F16, "cfm_ffn_fused: out16 dtype must be 16-bit");
    FfnArgs a = {};
    a.x = d->x; a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.w1f = (const u16*)d->w1f; a.w2f = (const u16*)d->w2f; a.b1 = d->b1; a.b2 = d->b2;
    a.ln1_g = d->ln1_g; a.ln1_b = d->ln1_b; a.ln2_g = d->ln2_g; a.ln2_b = d->ln2_b; a.out_f32 = d->out_f32; a.out16 = d->out16;
    a.M = d->M; a.FF = d->FF; a.act = d->act; a.out16_dtype = d->out16_dtype; a.add_x = d->add_x; a.alpha = d->alpha; a.eps = d->eps;
    hipStream_t s = (hipStream_t)stream;
    const bool bf = d->w_dtype == CFM_BF16;
    switch (d->D) {
        case 144: return bf ? launch_ffn<BF16, 144>(a, s, "ffn_fused_bf16_d144") : launch_ffn<F16, 144>(a, s, "ffn_fused_f16_d144");
        case 256: return bf ? launch_ffn<BF16, 256>(a, s, "ffn_fused_bf16_d256") : launch_ffn<F16, 256>(a, s, "ffn_fused_f16_d256");
        default: return cfm_fail(CFM_ERR_UNSUPPORTED, "cfm_ffn_fused: D=%d has no fused instance (144, 256)", d->D);
    }
}
