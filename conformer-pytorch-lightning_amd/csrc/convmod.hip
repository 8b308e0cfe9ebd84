// convmod.hip -- the HBM-bound middle of the convolution module and the front-end's first conv.
//
//  cfm_dwconv_bn_silu : depthwise k-tap FIR along time on a channels-last [B,T,D] activation, folded
//                       BatchNorm(eval) affine, SiLU  (convolution.py:43-45).  The GLU in front of it
//                       is the epilogue of the pointwise-conv-1 GEMM, so the chain GLU->dw->BN->SiLU
//                       costs one read and one write of [B,T,D].
//                       A lane owns 2 adjacent channels (coalesced 256 B per wavefront per frame) and an
//                       8-frame segment; the 8+k-1 input frames are loaded up front (all loads in flight
//                       together, the halo re-reads are L2 hits) and the FIR runs out of registers.
//  cfm_conv1_relu     : Conv2d(1,C,3,stride 2)+ReLU of the fbank image (convolution.py:60-61), written
//                       channels-last [B,T1,F1,C] so that the second conv is an implicit GEMM whose
//                       A fragments are contiguous 16-byte loads.
#include "cfm_common.h"

namespace {

constexpr int TSEG = 8;  // frames per lane: 32 gave only B*T/32*D/512 = 128 workgroups at config 2 (half the CUs idle)

template <int KTAPS, bool IN_F32>
__global__ __launch_bounds__(256) void cfm_dwconv_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ dwb, const float* __restrict__ sc,
                                                         const float* __restrict__ sh, void* __restrict__ y, int x_dt, int y_dt,
                                                         int T, int D) {
    constexpr int HALF = (KTAPS - 1) / 2;
    constexpr int NIN = TSEG + KTAPS - 1;
    const int pairs = D >> 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int pr = idx % pairs;
    const int seg = idx / pairs;
    const int t0 = seg * TSEG;
    if (t0 >= T) return;
    const int b = blockIdx.y;
    const int c = pr * 2;
    const int64_t base = (int64_t)b * T * D + c;

    float xin0[NIN], xin1[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int t = t0 - HALF + i;
        float a0 = 0.f, a1 = 0.f;
        if (t >= 0 && t < T) {
            if constexpr (IN_F32) {
                const float2 v = *(const float2*)((const float*)x + base + (int64_t)t * D);
                a0 = v.x;
                a1 = v.y;
            } else {
                const unsigned v = *(const unsigned*)((const u16*)x + base + (int64_t)t * D);
                const u16 lo = (u16)(v & 0xffffu), hi = (u16)(v >> 16);
                a0 = x_dt == CFM_BF16 ? BF16::to_f32(lo) : F16::to_f32(lo);
                a1 = x_dt == CFM_BF16 ? BF16::to_f32(hi) : F16::to_f32(hi);
            }
        }
        xin0[i] = a0;
        xin1[i] = a1;
    }
    float w0[KTAPS], w1[KTAPS];
#pragma unroll
    for (int k = 0; k < KTAPS; ++k) {
        w0[k] = w[(int64_t)c * KTAPS + k];
        w1[k] = w[(int64_t)(c + 1) * KTAPS + k];
    }
    const float b0 = dwb[c], b1 = dwb[c + 1], s0 = sc[c], s1 = sc[c + 1], h0 = sh[c], h1 = sh[c + 1];
#pragma unroll
    for (int i = 0; i < TSEG; ++i) {
        const int t = t0 + i;
        if (t >= T) break;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < KTAPS; ++k) {
            a0 = fmaf(w0[k], xin0[i + k], a0);
            a1 = fmaf(w1[k], xin1[i + k], a1);
        }
        a0 = siluf_((a0 + b0) * s0 + h0);
        a1 = siluf_((a1 + b1) * s1 + h1);
        const int64_t o = base + (int64_t)t * D;
        if (y_dt == CFM_F32)
            *(float2*)((float*)y + o) = make_float2(a0, a1);
        else if (y_dt == CFM_BF16)
            *(unsigned*)((u16*)y + o) = pack2<BF16>(a0, a1);
        else
            *(unsigned*)((u16*)y + o) = pack2<F16>(a0, a1);
    }
}

// LDS-tiled variant (the one used for k = 15): a block owns 16 frames x all D channels of one utterance.  The 16+14 input
// frames are fetched with 16-byte loads into an f32 LDS tile, each thread runs the FIR for 2 adjacent channels over its
// share of the frames out of a register window filled from LDS, results go back through LDS and leave as 16-byte stores.
// (The register-only kernel above moves 4 bytes per lane per access: 13.8 us for 8 MB at config 2.)
constexpr int DW_TS = 16;

template <int KTAPS>
__global__ __launch_bounds__(256) void cfm_dwconv_tiled_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ dwb, const float* __restrict__ sc,
                                                               const float* __restrict__ sh, void* __restrict__ y, int x_dt, int y_dt,
                                                               int T, int D) {
    constexpr int HALF = (KTAPS - 1) / 2;
    constexpr int NIN = DW_TS + KTAPS - 1;
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];   // [NIN][D] inputs, then [DW_TS][D] outputs
    float* tin = dw_lds;
    float* tout = dw_lds + NIN * D;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * DW_TS;
    const int64_t ubase = (int64_t)b * T * D;
    const int c8n = D >> 3;
    for (int id = tid; id < NIN * c8n; id += 256) {
        const int row = id / c8n, c = (id - row * c8n) * 8;
        const int t = t0 - HALF + row;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        if (t >= 0 && t < T) {
            const int64_t o = ubase + (int64_t)t * D + c;
            if (x_dt == CFM_F32) {
                a0 = *(const f32x4*)((const float*)x + o);
                a1 = *(const f32x4*)((const float*)x + o + 4);
            } else {
                const u32x4 r = *(const u32x4*)((const u16*)x + o);
                const unsigned wv[4] = {r.x, r.y, r.z, r.w};
                float f[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u16 lo = (u16)(wv[i] & 0xffffu), hi = (u16)(wv[i] >> 16);
                    f[2 * i] = x_dt == CFM_BF16 ? BF16::to_f32(lo) : F16::to_f32(lo);
                    f[2 * i + 1] = x_dt == CFM_BF16 ? BF16::to_f32(hi) : F16::to_f32(hi);
                }
                a0 = (f32x4){f[0], f[1], f[2], f[3]};
                a1 = (f32x4){f[4], f[5], f[6], f[7]};
            }
        }
        *(f32x4*)(tin + row * D + c) = a0;
        *(f32x4*)(tin + row * D + c + 4) = a1;
    }
    __syncthreads();
    const int pairs = D >> 1;
    const int groups = 256 / pairs;                     // frame groups per block (D = 256: 2, D = 144: 3)
    const int fpt = (DW_TS + groups - 1) / groups;      // frames per thread
    const int pr = tid % pairs, grp = tid / pairs;
    if (grp < groups) {
        const int c = pr * 2;
        float w0[KTAPS], w1[KTAPS];
#pragma unroll
        for (int k = 0; k < KTAPS; ++k) {
            w0[k] = w[(int64_t)c * KTAPS + k];
            w1[k] = w[(int64_t)(c + 1) * KTAPS + k];
        }
        const float b0 = dwb[c], b1 = dwb[c + 1], s0 = sc[c], s1 = sc[c + 1], h0 = sh[c], h1 = sh[c + 1];
        const int f0 = grp * fpt;
        constexpr int FMAX = 8;                         // fpt <= 8 for every supported D (pairs >= 32)
        float xin0[FMAX + KTAPS - 1], xin1[FMAX + KTAPS - 1];
#pragma unroll
        for (int i = 0; i < FMAX + KTAPS - 1; ++i) {
            const int row = f0 + i;
            float2 v = make_float2(0.f, 0.f);
            if (i < fpt + KTAPS - 1 && row < NIN) v = *(const float2*)(tin + row * D + c);
            xin0[i] = v.x;
            xin1[i] = v.y;
        }
#pragma unroll
        for (int i = 0; i < FMAX; ++i) {
            if (i < fpt && f0 + i < DW_TS) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int k = 0; k < KTAPS; ++k) {
                    a0 = fmaf(w0[k], xin0[i + k], a0);
                    a1 = fmaf(w1[k], xin1[i + k], a1);
                }
                *(float2*)(tout + (f0 + i) * D + c) = make_float2(siluf_((a0 + b0) * s0 + h0), siluf_((a1 + b1) * s1 + h1));
            }
        }
    }
    __syncthreads();
    for (int id = tid; id < DW_TS * c8n; id += 256) {
        const int row = id / c8n, c = (id - row * c8n) * 8;
        const int t = t0 + row;
        if (t >= T) continue;
        const f32x4 a0 = *(const f32x4*)(tout + row * D + c), a1 = *(const f32x4*)(tout + row * D + c + 4);
        const int64_t o = ubase + (int64_t)t * D + c;
        if (y_dt == CFM_F32) {
            *(f32x4*)((float*)y + o) = a0;
            *(f32x4*)((float*)y + o + 4) = a1;
        } else {
            *(u32x4*)((u16*)y + o) = y_dt == CFM_BF16 ? pack8<BF16>(a0, a1) : pack8<F16>(a0, a1);
        }
    }
}

// generic tap count (no register window): re-reads inputs through L1/L2
__global__ __launch_bounds__(256) void cfm_dwconv_generic_kernel(const void* x, const float* w, const float* dwb, const float* sc,
                                                                 const float* sh, void* y, int x_dt, int y_dt, int T, int D,
                                                                 int ktaps) {
    const int64_t n = (int64_t)T * D;
    const int b = blockIdx.y;
    const int half = (ktaps - 1) / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % D);
        const int t = (int)(i / D);
        float a = 0.f;
        for (int k = 0; k < ktaps; ++k) {
            const int tt = t + k - half;
            if (tt >= 0 && tt < T) a = fmaf(w[(int64_t)c * ktaps + k], load_as_f32(x, ((int64_t)b * T + tt) * D + c, x_dt), a);
        }
        a = siluf_((a + dwb[c]) * sc[c] + sh[c]);
        store_from_f32(y, (int64_t)b * n + i, y_dt, a);
    }
}

// Block = 256 threads = 32 channel-octets x 8 position slots; a thread keeps its 8 channels' 72 weights in registers and
// walks CONV1_PPT positions, so the weights are fetched once per thread instead of once per output position (the first
// version re-read them from L1 for every position and was L1-bound at 1.4 TB/s of output).  The 9 fbank taps of a
// position are the same address for the 32 lanes that share it (one broadcast load each); the 16-byte stores of those
// lanes are 512 contiguous bytes of the channels-last image.
constexpr int CONV1_PPT = 8;

template <bool OUT_F32>
__global__ __launch_bounds__(256) void cfm_conv1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, void* __restrict__ y, int y_dt, int B,
                                                        int T, int F, int T1, int F1, int C, const float* __restrict__ cm_mean,
                                                        const float* __restrict__ cm_istd) {
    const int c8n = C >> 3;                      // channel octets; blockDim.x = 256 covers 256/c8n position slots ... see host
    const int oct = threadIdx.x % c8n;
    const int slot = threadIdx.x / c8n;
    const int slots = 256 / c8n;
    const int c0 = oct * 8;
    const int64_t npos = (int64_t)B * T1 * F1;
    f32x4 w0[9], w1[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        w0[k] = *(const f32x4*)(w + k * C + c0);
        w1[k] = *(const f32x4*)(w + k * C + c0 + 4);
    }
    const f32x4 b0 = *(const f32x4*)(bias + c0), b1 = *(const f32x4*)(bias + c0 + 4);
    // (b, t1, f1) of the thread's first position by ONE 32-bit decomposition, then carried forward: `slots` positions per step.
    // (The first version divided 64-bit indices four times per position: ~400 VALU instructions next to 36 packed FMAs.)
    const unsigned first = (blockIdx.x * (unsigned)CONV1_PPT) * (unsigned)slots + (unsigned)slot;
    unsigned f1 = first % (unsigned)F1, bt0 = first / (unsigned)F1;
    unsigned t1 = bt0 % (unsigned)T1, b = bt0 / (unsigned)T1;
    // all taps of all CONV1_PPT positions are requested before the first is used (no early exit inside the loop: positions
    // past the end are clamped for the loads and predicated for the store)
    float xv[CONV1_PPT][9];
    int64_t opos[CONV1_PPT];
#pragma unroll
    for (int it = 0; it < CONV1_PPT; ++it) {
        const int64_t pos = (int64_t)first + (int64_t)it * slots;
        opos[it] = pos < npos ? pos : -1;
        if (it) {
            f1 += (unsigned)slots;
            while (f1 >= (unsigned)F1) {                  // slots <= 32 < F1 in every configuration of interest: one pass
                f1 -= (unsigned)F1;
                if (++t1 == (unsigned)T1) { t1 = 0; ++b; }
            }
        }
        const unsigned bc = b < (unsigned)B ? b : (unsigned)B - 1u;
        const float* xp = x + ((int64_t)bc * T + 2 * t1) * F + 2 * f1;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) xv[it][kt * 3 + kf] = xp[kt * F + kf];
        if (cm_mean) {                                     // global CMVN folded into the taps: (x - mean[f]) * istd[f]
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const float mu = cm_mean[2 * f1 + kf], is = cm_istd ? cm_istd[2 * f1 + kf] : 1.0f;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const float d = xv[it][kt * 3 + kf] - mu;
                    xv[it][kt * 3 + kf] = cm_istd ? d * is : d;
                }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < CONV1_PPT; ++it) {
        f32x4 a0 = b0, a1 = b1;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            a0 += xv[it][k] * w0[k];
            a1 += xv[it][k] * w1[k];
        }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        a0 = __builtin_elementwise_max(a0, z);
        a1 = __builtin_elementwise_max(a1, z);
        if (opos[it] < 0) continue;
        const int64_t o = opos[it] * C + c0;
        if constexpr (OUT_F32) {
            *(f32x4*)((float*)y + o) = a0;
            *(f32x4*)((float*)y + o + 4) = a1;
        } else {
            *(u32x4*)((u16*)y + o) = y_dt == CFM_BF16 ? pack8<BF16>(a0, a1) : pack8<F16>(a0, a1);
        }
    }
}

// Second form of the same convolution, used when F % 4 == 0: a thread takes EIGHT CONSECUTIVE output positions of one output row
// (b, t1, f1_0 .. f1_0+7) instead of eight positions scattered over the image, so its input is one 3 x 17-float window read with
// 15 vector loads (the form above issues 72 single-dword loads per thread and is bound by load issue, not by its 319 MB of
// stores: it stays at ~3 TB/s even when the output fits the Infinity Cache).  Same arithmetic in the same order: bit-identical.
template <bool OUT_F32>
__global__ __launch_bounds__(256) void cfm_conv1_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, void* __restrict__ y, int y_dt, int B,
                                                             int T, int F, int T1, int F1, int C, const float* __restrict__ cm_mean,
                                                             const float* __restrict__ cm_istd) {
    const int c8n = C >> 3;
    const int oct = threadIdx.x % c8n, slot = threadIdx.x / c8n, slots = 256 / c8n;
    const int c0 = oct * 8;
    const int NG = (F1 + 7) >> 3;                          // groups of 8 positions per output row
    const unsigned group = blockIdx.x * (unsigned)slots + (unsigned)slot;
    if (group >= (unsigned)B * (unsigned)T1 * (unsigned)NG) return;
    const unsigned fg = group % (unsigned)NG, bt = group / (unsigned)NG;
    const unsigned t1 = bt % (unsigned)T1, b = bt / (unsigned)T1;
    const int f1_0 = (int)fg * 8;
    const int nvalid = min(8, F1 - f1_0);
    f32x4 w0[9], w1[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        w0[k] = *(const f32x4*)(w + k * C + c0);
        w1[k] = *(const f32x4*)(w + k * C + c0 + 4);
    }
    const f32x4 b0 = *(const f32x4*)(bias + c0), b1 = *(const f32x4*)(bias + c0 + 4);
    float win[3][20];                                      // 17 used; 5 x 16-byte loads per row, columns past F read as zero
    const int fbase = 2 * f1_0;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
        const float* xr = x + ((int64_t)b * T + 2 * t1 + kt) * F + fbase;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (fbase + 4 * q + 3 < F) {
                v = *(const f32x4*)(xr + 4 * q);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (fbase + 4 * q + e < F) v[e] = xr[4 * q + e];
            }
            win[kt][4 * q] = v.x; win[kt][4 * q + 1] = v.y; win[kt][4 * q + 2] = v.z; win[kt][4 * q + 3] = v.w;
        }
    }
    if (cm_mean) {                                         // global CMVN folded into the taps: (x - mean[f]) * istd[f]
#pragma unroll
        for (int j = 0; j < 17; ++j) {
            const int f = fbase + j;
            const float mu = f < F ? cm_mean[f] : 0.f, is = (cm_istd && f < F) ? cm_istd[f] : 1.0f;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const float d = win[kt][j] - mu;
                win[kt][j] = cm_istd ? d * is : d;
            }
        }
    }
    const int64_t pos0 = (int64_t)bt * F1 + f1_0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= nvalid) break;
        // scalar FMAs on purpose (this file is built with -fno-slp-vectorize): v_pk_fma_f32 costs more than two v_fma_f32
        float acc[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const float xv = win[kt][2 * i + kf];
                const f32x4 wa = w0[kt * 3 + kf], wb = w1[kt * 3 + kf];
                acc[0] = fmaf(xv, wa.x, acc[0]); acc[1] = fmaf(xv, wa.y, acc[1]); acc[2] = fmaf(xv, wa.z, acc[2]); acc[3] = fmaf(xv, wa.w, acc[3]);
                acc[4] = fmaf(xv, wb.x, acc[4]); acc[5] = fmaf(xv, wb.y, acc[5]); acc[6] = fmaf(xv, wb.z, acc[6]); acc[7] = fmaf(xv, wb.w, acc[7]);
            }
        f32x4 a0 = {fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f)};
        f32x4 a1 = {fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f)};
        const int64_t o = (pos0 + i) * C + c0;
        if constexpr (OUT_F32) {
            *(f32x4*)((float*)y + o) = a0;
            *(f32x4*)((float*)y + o + 4) = a1;
        } else {
            *(u32x4*)((u16*)y + o) = y_dt == CFM_BF16 ? pack8<BF16>(a0, a1) : pack8<F16>(a0, a1);
        }
    }
}

// Third form, on the matrix pipe: the 9 taps are a K = 32 contraction (3 taps of one input row in each of three 8-wide K groups, the
// bias in the fourth, the rest zero), the 16 x 16 x 32 MFMA takes the weights as its A operand (16 channel fragments of 4 VGPRs, built once per wavefront and
// kept in registers) and 16 output positions as its B operand, so a lane ends up with 4 consecutive channels of one position per
// fragment.  The FMA forms above spend ~55 us of VALU time on 9 x 159 M multiply-adds at config 2; here the arithmetic is 4 us of MFMA
// and what remains is the 319 MB write.  Inputs and weights are ROUNDED TO THE 16-BIT OPERAND TYPE (the FMA forms multiply in f32):
// used by the bf16 / fp16 precision modes, whose conv1 output is 16-bit anyway; the f32-accurate mode keeps the FMA form.
template <typename HT>
__global__ __launch_bounds__(256) void cfm_conv1_mma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, u16* __restrict__ y, int B, int T, int F,
                                                            int T1, int F1, int C, const float* __restrict__ cm_mean,
                                                            const float* __restrict__ cm_istd, int NH) {
    constexpr int MAXF = 16;                               // channel fragments (C <= 256)
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, kg = lane >> 4;
    const int nfr = C >> 4, cq = C >> 2;
    const int nfh = nfr / NH, cqh = cq / NH, Ch = C / NH;  // per pass (NH = 2: the channels go through the LDS tile in two halves)
    const unsigned npos = (unsigned)B * (unsigned)T1 * (unsigned)F1;   // < 2^31 (host check)
    const unsigned nfrag = (npos + 15u) >> 4;
    const unsigned wave0 = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    // Weights: the MFMA's output row i = 4q + r of channel fragment j (pass h = j / nfh, jh = j % nfh) is made channel
    // h*(C/NH) + q*(C/4/NH) + 4 jh + r, so that within a pass the lane that ends up with rows 4q .. 4q+3 of every fragment holds CONSECUTIVE
    // channels of its position (16-byte LDS writes) and the pass as a whole covers C/NH consecutive channels (whole 128-byte lines in
    // memory).  A-operand lane (row l15, K group kg) holds taps 3kg .. 3kg+2 of its channel, then five zeros.
    // The BIAS rides in the contraction too: K group 3 multiplies the constants (1, 1) by (bias_hi, bias_lo), the 16-bit hi/lo split of the
    // f32 bias (exact to ~16 mantissa bits), so the accumulator starts at zero and no 64 bias registers are carried: 151 -> ~90 VGPRs,
    // four resident workgroups per CU instead of three.
    u32x4 wf[MAXF];
#pragma unroll
    for (int j = 0; j < MAXF; ++j) {
        wf[j] = (u32x4){0u, 0u, 0u, 0u};
        if (j < nfr) {
            const int ch = (j / nfh) * Ch + (l15 >> 2) * cqh + 4 * (j % nfh) + (l15 & 3);
            if (kg < 3) {
                const float a = w[(3 * kg) * C + ch], b = w[(3 * kg + 1) * C + ch], c = w[(3 * kg + 2) * C + ch];
                wf[j].x = pack2<HT>(a, b);
                wf[j].y = pack2<HT>(c, 0.f);
            } else {
                const float bv = bias[ch];
                const float hi = HT::to_f32(HT::from_f32(bv));
                wf[j].x = pack2<HT>(hi, bv - hi);
            }
        }
    }
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one fragment = 16 consecutive output positions; lanes of K group kg read input row 2 t1 + kg, columns 2 f1 .. 2 f1 + 2
    auto gather = [&](unsigned fr, unsigned& p_out, bool& live_out) __attribute__((always_inline)) {
        unsigned p = fr * 16u + (unsigned)l15;
        live_out = p < npos;
        p = live_out ? p : npos - 1u;
        p_out = p;
        const unsigned f1 = p % (unsigned)F1, bt = p / (unsigned)F1;
        const unsigned t1 = bt % (unsigned)T1, b = bt / (unsigned)T1;
        u32x4 xf = (u32x4){0u, 0u, 0u, 0u};
        if (kg < 3) {
            const float* xr = x + (((int64_t)b * T + 2 * t1 + kg) * F + 2 * f1);
            float v0 = xr[0], v1 = xr[1], v2 = xr[2];
            if (cm_mean) {
                const unsigned f = 2 * f1;
                v0 -= cm_mean[f]; v1 -= cm_mean[f + 1]; v2 -= cm_mean[f + 2];
                if (cm_istd) { v0 *= cm_istd[f]; v1 *= cm_istd[f + 1]; v2 *= cm_istd[f + 2]; }
            }
            xf.x = pack2<HT>(v0, v1);
            xf.y = pack2<HT>(v2, 0.f);
        } else {
            xf.x = pack2<HT>(1.0f, 1.0f);                   // times (bias_hi, bias_lo)
        }
        return xf;
    };
    extern __shared__ __attribute__((aligned(16))) u16 conv1_lds[];
    const int rs = Ch + 8;                                 // LDS row stride in 16-bit elements (one pass of channels + 16 bytes)
    u16* const tile = conv1_lds + (threadIdx.x >> 6) * (16 * rs);
    unsigned p = 0;
    bool live = false;
    auto emit = [&](unsigned fr, const u32x4& xf) __attribute__((always_inline)) {
        // results go through a per-wavefront LDS tile [16 positions][C/NH] (rows padded by 16 B: conflict-free 16-byte writes) so that the
        // global stores are whole runs of consecutive bytes: a position's C/NH channels (256 or 512 B) per 16 or 32 lanes.  Stored
        // straight from the MFMA layout, every instruction touched 16 (8-byte pieces) or 64 (16-byte pieces) different 128-byte lines:
        // 119 and 256 us against 105 us for the FMA form.  Two passes halve the tile: more wavefronts fit a CU.
        u16* const tp = tile + l15 * rs + kg * cqh;
        const unsigned pbase = fr * 16u;                     // the fragment's 16 positions are consecutive in memory
        const int cpp = Ch >> 3;                             // 16-byte chunks per position and pass
        for (int h = 0; h < NH; ++h) {
            const int j0 = h * nfh, j1 = j0 + nfh;           // this pass's fragments (j0 even: NH = 2 only when nfr % 4 == 0)
#pragma unroll
            for (int j = 0; j < MAXF; j += 2) {              // static register indices; the pass test is wave-uniform
                if (j >= j0 && j < j1) {
                    f32x4 a = HT::mfma(wf[j], xf, zero4);
                    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                    if (j + 1 < j1) {
                        f32x4 c = HT::mfma(wf[j + 1 < MAXF ? j + 1 : j], xf, zero4);
                        c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
                        *(u32x4*)(tp + 4 * (j - j0)) = pack8<HT>(a, c);
                    } else {
                        *(u32x2*)(tp + 4 * (j - j0)) = (u32x2){pack2<HT>(a.x, a.y), pack2<HT>(a.z, a.w)};
                    }
                }
            }
            for (int ch = lane; ch < 16 * cpp; ch += 64) {
                const int pi = ch / cpp, cc = ch - pi * cpp;
                if (pbase + (unsigned)pi < npos)
                    *(u32x4*)(y + ((int64_t)(pbase + pi) * C + h * Ch + cc * 8)) = *(const u32x4*)(tile + pi * rs + cc * 8);
            }
        }
    };
    // two fragments per trip: both gathers are in flight before the first fragment's MFMAs and stores
    for (unsigned fr = wave0; fr < nfrag; fr += 2 * nwaves) {
        const unsigned fr2 = fr + nwaves;
        const u32x4 xa = gather(fr, p, live);
        u32x4 xb = (u32x4){0u, 0u, 0u, 0u};
        if (fr2 < nfrag) xb = gather(fr2, p, live);
        emit(fr, xa);
        if (fr2 < nfrag) emit(fr2, xb);
    }
}

__global__ void cfm_cast_kernel(const void* src, int sdt, void* dst, int ddt, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        store_from_f32(dst, i, ddt, load_as_f32(src, i, sdt));
}

__global__ void cfm_add_rows_kernel(float* x, const float* add, int64_t n4, int D4, int group) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D4;
        const int c = (int)(i % D4);
        f32x4* px = (f32x4*)x + i;
        *px = *px + *((const f32x4*)add + (r / group) * D4 + c);
    }
}

}  // namespace

extern "C" int cfm_add_rows(float* x, const float* add, int64_t rows, int32_t D, int32_t group, cfm_stream_t stream) {
    CFM_CHECK_ARG(x && add && rows > 0 && D > 0 && D % 4 == 0 && group > 0, "cfm_add_rows: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = rows * (D / 4);
    int64_t nb = (n4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    CfmProfScope prof("add_rows", s, 0.0, (double)rows * D * 8);
    CFM_LAUNCH(cfm_add_rows_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, add, n4, D / 4, group);
    return cfm_launch_status("cfm_add_rows");
}

extern "C" int cfm_dwconv_bn_silu(const void* x, int x_dtype, const float* w, const float* dw_bias, const float* bn_scale,
                                  const float* bn_shift, void* y, int y_dtype, int32_t B, int32_t T, int32_t D, int32_t ktaps,
                                  cfm_stream_t stream) {
    CFM_CHECK_ARG(x && w && dw_bias && bn_scale && bn_shift && y, "cfm_dwconv_bn_silu: null pointer");
    CFM_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 2 == 0 && B <= 65535, "cfm_dwconv_bn_silu: bad shape B=%d T=%d D=%d", B, T, D);
    CFM_CHECK_ARG(ktaps > 0 && ktaps % 2 == 1, "cfm_dwconv_bn_silu: tap count must be odd (got %d)", ktaps);
    hipStream_t s = (hipStream_t)stream;
    const double bytes = (double)B * T * D * (cfm_elt_size(x_dtype) + cfm_elt_size(y_dtype));
    CfmProfScope prof("dwconv_bn_silu", s, 2.0 * B * T * (double)D * ktaps, bytes);
    // tiled kernel: 256 threads cover D/2 channel pairs x (256 / (D/2)) frame groups, at most 8 frames per thread => D <= 256
    if (ktaps == 15 && D % 8 == 0 && D >= 64 && D <= 256 && (DW_TS + 256 / (D / 2) - 1) / (256 / (D / 2)) <= 8) {
        const dim3 grid((unsigned)((T + DW_TS - 1) / DW_TS), B), block(256);
        const size_t lds = (size_t)(DW_TS + 14 + DW_TS) * D * sizeof(float);
        CFM_LAUNCH((cfm_dwconv_tiled_kernel<15>), grid, block, lds, s, x, w, dw_bias, bn_scale, bn_shift, y, x_dtype, y_dtype, T, D);
    } else if (ktaps == 15) {
        const int segs = (T + TSEG - 1) / TSEG;
        const dim3 grid((unsigned)(((int64_t)segs * (D / 2) + 255) / 256), B), block(256);
        if (x_dtype == CFM_F32)
            CFM_LAUNCH((cfm_dwconv_kernel<15, true>), grid, block, 0, s, x, w, dw_bias, bn_scale, bn_shift, y, x_dtype,
                               y_dtype, T, D);
        else
            CFM_LAUNCH((cfm_dwconv_kernel<15, false>), grid, block, 0, s, x, w, dw_bias, bn_scale, bn_shift, y, x_dtype,
                               y_dtype, T, D);
    } else {
        int64_t nb = ((int64_t)T * D + 255) / 256;
        if (nb > 1024) nb = 1024;
        CFM_LAUNCH(cfm_dwconv_generic_kernel, dim3((unsigned)nb, B), dim3(256), 0, s, x, w, dw_bias, bn_scale, bn_shift, y,
                           x_dtype, y_dtype, T, D, ktaps);
    }
    return cfm_launch_status("cfm_dwconv_bn_silu");
}

extern "C" int cfm_conv1_relu(const float* x, const float* w, const float* bias, void* y, int y_dtype, int32_t B, int32_t T,
                              int32_t F, int32_t C, const float* cmvn_mean, const float* cmvn_istd, cfm_stream_t stream) {
    CFM_CHECK_ARG(cmvn_mean || !cmvn_istd, "cfm_conv1_relu: cmvn_istd without cmvn_mean");
    CFM_CHECK_ARG(x && w && bias && y, "cfm_conv1_relu: null pointer");
    CFM_CHECK_ARG(B > 0 && T >= 3 && F >= 3 && C > 0 && C % 8 == 0, "cfm_conv1_relu: bad shape B=%d T=%d F=%d C=%d", B, T, F, C);
    CFM_CHECK_ARG(C / 8 <= 256, "cfm_conv1_relu: C=%d too wide (max 2048)", C);
    const int T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
    const int64_t total = (int64_t)B * T1 * F1 * (C / 8);
    const int slots = 256 / (C / 8);                         // position slots per block (threads beyond slots*C/8 idle)
    const int64_t npos = (int64_t)B * T1 * F1;
    CFM_CHECK_ARG(npos < (int64_t)1 << 31, "cfm_conv1_relu: %lld output positions do not fit the kernel's 32-bit position index", (long long)npos);
    const int64_t nblocks = (npos + (int64_t)slots * CONV1_PPT - 1) / ((int64_t)slots * CONV1_PPT);
    hipStream_t s = (hipStream_t)stream;
    const double bytes = (double)B * T * F * 4 + (double)total * 8 * cfm_elt_size(y_dtype);
    CfmProfScope prof("conv1_relu", s, 2.0 * 9 * (double)total * 8, bytes);
    const dim3 grid((unsigned)nblocks), block((unsigned)(slots * (C / 8)));
    if (F % 4 == 0 && ((uintptr_t)x & 15) == 0) {          // row-window form: 15 vector loads per thread instead of 72 scalar ones
        const int64_t ngroups = (int64_t)B * T1 * ((F1 + 7) / 8);
        const dim3 g2((unsigned)((ngroups + slots - 1) / slots));
        if (y_dtype == CFM_F32)
            CFM_LAUNCH((cfm_conv1_rows_kernel<true>), g2, block, 0, s, x, w, bias, y, y_dtype, B, T, F, T1, F1, C, cmvn_mean, cmvn_istd);
        else
            CFM_LAUNCH((cfm_conv1_rows_kernel<false>), g2, block, 0, s, x, w, bias, y, y_dtype, B, T, F, T1, F1, C, cmvn_mean, cmvn_istd);
        return cfm_launch_status("cfm_conv1_relu");
    }
    if (y_dtype == CFM_F32)
        CFM_LAUNCH((cfm_conv1_kernel<true>), grid, block, 0, s, x, w, bias, y, y_dtype, B, T, F, T1, F1, C, cmvn_mean, cmvn_istd);
    else
        CFM_LAUNCH((cfm_conv1_kernel<false>), grid, block, 0, s, x, w, bias, y, y_dtype, B, T, F, T1, F1, C, cmvn_mean, cmvn_istd);
    return cfm_launch_status("cfm_conv1_relu");
}

extern "C" int cfm_conv1_relu_mma(const float* x, const float* w, const float* bias, void* y, int y_dtype, int32_t B, int32_t T, int32_t F,
                                  int32_t C, const float* cmvn_mean, const float* cmvn_istd, cfm_stream_t stream) {
    CFM_CHECK_ARG(cmvn_mean || !cmvn_istd, "cfm_conv1_relu_mma: cmvn_istd without cmvn_mean");
    CFM_CHECK_ARG(x && w && bias && y, "cfm_conv1_relu_mma: null pointer");
    CFM_CHECK_ARG(B > 0 && T >= 3 && F >= 3 && C > 0 && C % 16 == 0 && C <= 256, "cfm_conv1_relu_mma: bad shape B=%d T=%d F=%d C=%d (C %% 16 == 0, C <= 256)", B, T, F, C);
    CFM_CHECK_ARG(y_dtype == CFM_BF16 || y_dtype == CFM_F16, "cfm_conv1_relu_mma: the output (= operand) type must be bf16 or fp16");
    const int T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
    const int64_t npos = (int64_t)B * T1 * F1, nfrag = (npos + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    // channel passes through the LDS tile: 1.  Two passes (half the tile, five resident workgroups per CU instead of four) measured
    // slower, 87 vs 75 us: 256-byte instead of 512-byte runs per position and twice the store loops
    const int NH = 1;
    int64_t nb = (nfrag + 3) / 4;
    if (nb > 256 * 4) nb = 256 * 4;                          // 4 workgroups of 4 wavefronts per CU (LDS), each wavefront strides over the fragments
    const size_t lds = (size_t)4 * 16 * (C / NH + 8) * 2;    // one [16][C/NH + 8] 16-bit tile per wavefront
    CfmProfScope prof("conv1_relu_mma", s, 2.0 * 9 * (double)npos * C, (double)B * T * F * 4 + (double)npos * C * 2);
    if (y_dtype == CFM_BF16)
        CFM_LAUNCH((cfm_conv1_mma_kernel<BF16>), dim3((unsigned)nb), dim3(256), lds, s, x, w, bias, (u16*)y, B, T, F, T1, F1, C, cmvn_mean, cmvn_istd, NH);
    else
        CFM_LAUNCH((cfm_conv1_mma_kernel<F16>), dim3((unsigned)nb), dim3(256), lds, s, x, w, bias, (u16*)y, B, T, F, T1, F1, C, cmvn_mean, cmvn_istd, NH);
    return cfm_launch_status("cfm_conv1_relu_mma");
}

extern "C" int cfm_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, cfm_stream_t stream) {
    CFM_CHECK_ARG(src && dst && n > 0, "cfm_cast: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    CfmProfScope prof("cast", s, 0.0, (double)n * (cfm_elt_size(src_dtype) + cfm_elt_size(dst_dtype)));
    CFM_LAUNCH(cfm_cast_kernel, dim3((unsigned)nb), dim3(256), 0, s, src, src_dtype, dst, dst_dtype, n);
    return cfm_launch_status("cfm_cast");
}
