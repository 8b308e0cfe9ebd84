// train.hip -- the HBM-bound half of the backward pass (config 3: encoder + CTC loss + backward).
//
//   cfm_layernorm_bwd        d LayerNorm (+ the residual branch's gradient) and its gain / bias gradients  (encoder_layer.py:56-70)
//   cfm_glu_bwd              d GLU over the interleaved pointwise-conv-1 output                            (convolution.py:42)
//   cfm_dwconv_bn_train      depthwise conv -> BatchNorm1d with BATCH statistics (all B*T rows, padded ones included: quirk Q6)
//                            -> SiLU, running statistics updated as torch does                              (convolution.py:43-45)
//   cfm_dwconv_bn_train_bwd  its backward: d SiLU, d BatchNorm (batch statistics), d depthwise conv (input, taps, bias)
//   cfm_col2im_relu_bwd      scatter-free transpose of the 3x3 stride-2 im2col (a <= 4-term gather per element) * ReLU'
//   cfm_conv1_wgrad          tap / bias gradients of the first convolution                                  (convolution.py:60-61)
//   cfm_adam_step            fused clip-scale + Adam update over a flat parameter buffer                    (module.py:140-143, train.sh:35)
//   cfm_sumsq                sum of squares of a flat buffer (the global gradient norm of gradient_clip_val, executor.py:150)
//
// Everything here is a streaming pass: one read of its inputs, one write of its outputs, reductions over the M = B*T' rows as
// per-workgroup partials in a caller-provided f32 workspace followed by a fixed-order second stage (bitwise reproducible -- no
// atomics).  The GEMM-shaped part of the backward is gemm.hip (input gradients) and gemm_tn.hip (weight gradients).
#include "cfm_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------------------
// loads / stores by dtype, 4 consecutive elements
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 load4(const void* base, int dt, int64_t off) {
    if (dt == CFM_F32) return *(const f32x4*)((const float*)base + off);
    const u32x2 r = *(const u32x2*)((const u16*)base + off);
    if (dt == CFM_BF16)
        return (f32x4){BF16::to_f32((u16)(r.x & 0xffffu)), BF16::to_f32((u16)(r.x >> 16)), BF16::to_f32((u16)(r.y & 0xffffu)), BF16::to_f32((u16)(r.y >> 16))};
    return (f32x4){F16::to_f32((u16)(r.x & 0xffffu)), F16::to_f32((u16)(r.x >> 16)), F16::to_f32((u16)(r.y & 0xffffu)), F16::to_f32((u16)(r.y >> 16))};
}
__device__ __forceinline__ void store4(void* base, int dt, int64_t off, const f32x4& v) {
    if (dt == CFM_F32) *(f32x4*)((float*)base + off) = v;
    else if (dt == CFM_BF16) *(u32x2*)((u16*)base + off) = (u32x2){pack2<BF16>(v.x, v.y), pack2<BF16>(v.z, v.w)};
    else *(u32x2*)((u16*)base + off) = (u32x2){pack2<F16>(v.x, v.y), pack2<F16>(v.z, v.w)};
}
__device__ __forceinline__ float dsilu_(float z) {
    const float s = sigmoidf_(z);
    return s * (1.f + z * (1.f - s));
}

// out_a[j] = alpha * sum_b part[b*J + j] for j < J1, out_b[j - J1] for the rest: the fixed-order second stage of every reduction here
// 64 columns per workgroup, 4 threads per column each summing every 4th partial (fixed order), combined through LDS: the serial chain is
// nblk/4 loads long instead of nblk (latency-bound: ~0.1 us per dependent load)
constexpr int RP_Q = 16;
__global__ __launch_bounds__(1024) void cfm_reduce_partials_kernel(const float* __restrict__ part, int nblk, int J, int J1, float alpha, float* out_a, float* out_b) {
    __shared__ float red[RP_Q][64];
    const int jc = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jc;
    float s0 = 0.f, s1 = 0.f;
    if (j < J) {
        int b = q;
        for (; b + RP_Q < nblk; b += 2 * RP_Q) {
            s0 += part[(int64_t)b * J + j];
            s1 += part[(int64_t)(b + RP_Q) * J + j];
        }
        if (b < nblk) s0 += part[(int64_t)b * J + j];
    }
    red[q][jc] = s0 + s1;
    __syncthreads();
    if (q == 0 && j < J) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RP_Q; ++i) s += red[i][jc];              // fixed order
        s *= alpha;
        if (j < J1) out_a[j] = s;
        else out_b[j - J1] = s;
    }
}

int reduce_partials(const float* part, int nblk, int J, int J1, float alpha, float* out_a, float* out_b, hipStream_t s, const char* what) {
    CfmProfScope prof("reduce_partials", s, 0.0, (double)nblk * J * 4);
    CFM_LAUNCH(cfm_reduce_partials_kernel, dim3((unsigned)((J + 63) / 64)), dim3(64 * RP_Q), 0, s, part, nblk, J, J1, alpha, out_a, out_b);
    return cfm_launch_status(what);
}

// ------------------------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  One wavefront per row (as norm.hip), 32 rows per workgroup; the row statistics are recomputed from x.
//   xhat = (x - mean) * rstd;  gy = gamma * dy;   dx = dres + rstd * (gy - mean_D(gy) - xhat * mean_D(gy * xhat))
//   dgamma = sum_rows dy * xhat;  dbeta = sum_rows dy          (per-workgroup partials -> ws[blk][2][D])
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int LNB_WAVES = 8;      // 2 rows per wavefront (at M = 2 k rows a 32-row, 4-wavefront workgroup left three quarters of the CUs idle: 12 us -> ~4 us);
constexpr int LNB_ROWS = 2 * LNB_WAVES;   // 8 wavefronts halve the workgroups that meet in the parameter-gradient sums (atomics of the one-launch form: 10.3 us with 4)

// The element types of g / dg are TEMPLATE arguments: with a run-time dtype every one of the 30 window loads sits behind a branch, the compiler
// waits at each join and the loads of a thread go out one memory latency after the other (28 us per launch at a config-3 micro-batch).
template <int GDT>
__device__ __forceinline__ float ld_t(const void* p, int64_t i) {
    if constexpr (GDT == CFM_F32) return ((const float*)p)[i];
    else if constexpr (GDT == CFM_BF16) return BF16::to_f32(((const u16*)p)[i]);
    else return F16::to_f32(((const u16*)p)[i]);
}
template <int GDT>
__device__ __forceinline__ void st_t(void* p, int64_t i, float v) {
    if constexpr (GDT == CFM_F32) ((float*)p)[i] = v;
    else if constexpr (GDT == CFM_BF16) ((u16*)p)[i] = BF16::from_f32(v);
    else ((u16*)p)[i] = F16::from_f32(v);
}

// Optional second output (LnBwd2): the NEXT consumer of dx in a conformer block's backward is a residual branch whose gradient enters its
// GEMMs as dropout-mask * alpha * dx in the activation dtype (cfm_dropout_rows); written here it saves that launch and its read of dx.
struct LnBwd2 {
    void* y;
    int dt;
    float alpha;
    CfmDrop d1, d2;
    const uint8_t* mask;   // rows with mask == 0 are written as zeros (the consumer's row mask applied here)
};

template <int DT>
__device__ __forceinline__ f32x4 ld4_t(const void* p, int64_t i) {
    if constexpr (DT == CFM_F32) return *(const f32x4*)((const float*)p + i);
    else {
        const u32x2 r = *(const u32x2*)((const u16*)p + i);
        if constexpr (DT == CFM_BF16) return (f32x4){BF16::to_f32((u16)(r.x & 0xffffu)), BF16::to_f32((u16)(r.x >> 16)), BF16::to_f32((u16)(r.y & 0xffffu)), BF16::to_f32((u16)(r.y >> 16))};
        else return (f32x4){F16::to_f32((u16)(r.x & 0xffffu)), F16::to_f32((u16)(r.x >> 16)), F16::to_f32((u16)(r.y & 0xffffu)), F16::to_f32((u16)(r.y >> 16))};
    }
}

// CHAIN: a SECOND LayerNorm backward on the same rows, in registers -- consecutive conformer blocks end / begin with one (block l+1's
// norm_ff_macaron reads block l's norm_final output, encoder_layer.py:57,70), so the backward runs dLN_ffm then dLN_final on every row:
//     d1 = dres + dLN(dy; x, gamma)            (stage 1, never stored)
//     dx = dLN(d1; ch.x, ch.gamma)             (stage 2; dx and the optional second output o2 come from THIS)
// one launch instead of two (each ~13 us at a training window, launch-latency-bound).  Parameter gradients of both norms by atomics.
struct LnChain {
    const float* x;
    const float* gamma;
    float *acc_g, *acc_b;
};

template <int ITERS, int DYDT, bool CHAIN>
__global__ __launch_bounds__(64 * LNB_WAVES) void cfm_layernorm_bwd_kernel(const float* __restrict__ x, const void* __restrict__ dy, int dy_dt,
                                                                const float* __restrict__ gamma, const uint8_t* __restrict__ mask,
                                                                const float* dres, float* dx, float* __restrict__ ws, float eps, int64_t M, int D,
                                                                float* acc_g, float* acc_b, LnBwd2 o2, LnChain ch) {
    __shared__ float red[LNB_WAVES][CHAIN ? 4 : 2][ITERS * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[ITERS], db[ITERS], gm[ITERS];
    f32x4 dg2[CHAIN ? ITERS : 1], db2[CHAIN ? ITERS : 1], gm2[CHAIN ? ITERS : 1];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (lane + 64 * it) * 4;
        dg[it] = db[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gm[it] = c < D ? *(const f32x4*)(gamma + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (CHAIN) {
            dg2[it] = db2[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            gm2[it] = c < D ? *(const f32x4*)(ch.gamma + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const float invD = 1.0f / (float)D;
    // every load of the wavefront's two rows goes out first, unconditionally (clamped addresses, values selected afterwards): x, dy, the residual
    // gradient (it was read after the row's two reductions, one more exposed latency per row) and the mask byte
    constexpr int NR = LNB_ROWS / LNB_WAVES;
    const f32x4 z4f = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 xa[NR][ITERS], da[NR][ITERS], ra[NR][ITERS];
    f32x4 xb[CHAIN ? NR : 1][ITERS];                       // CHAIN: the second norm's input rows
    bool keep_a[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int64_t row_u = (int64_t)blockIdx.x * LNB_ROWS + wave + rr * LNB_WAVES;
        const int64_t row_c = row_u < M ? row_u : M - 1;
        keep_a[rr] = mask ? mask[row_c] != 0 : true;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (lane + 64 * it) * 4;
            const int cc = c < D ? c : 0;
            xa[rr][it] = *(const f32x4*)(x + row_c * D + cc);
            da[rr][it] = ld4_t<DYDT>(dy, row_c * D + cc);
            ra[rr][it] = dres ? *(const f32x4*)(dres + row_c * D + cc) : z4f;
            if constexpr (CHAIN) xb[rr][it] = *(const f32x4*)(ch.x + row_c * D + cc);
        }
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int64_t row = (int64_t)blockIdx.x * LNB_ROWS + wave + rr * LNB_WAVES;
        if (row >= M) break;                               // wave-uniform
        const bool keep = keep_a[rr];
        f32x4 xv[ITERS], dv[ITERS];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (lane + 64 * it) * 4;
            const bool in = c < D;
            xv[it] = in ? xa[rr][it] : z4f;
            dv[it] = (in && keep) ? da[rr][it] : z4f;
            s += (xv[it].x + xv[it].y) + (xv[it].z + xv[it].w);
        }
        const float mean = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (lane + 64 * it) * 4;
            if (c < D) {
                const f32x4 d = xv[it] - mean;
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * invD + eps);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (lane + 64 * it) * 4;
            if (c < D) {
                xv[it] = (xv[it] - mean) * rstd;           // xhat
                const f32x4 gy = gm[it] * dv[it];
                a += (gy.x + gy.y) + (gy.z + gy.w);
                const f32x4 t = gy * xv[it];
                b += (t.x + t.y) + (t.z + t.w);
                dg[it] += dv[it] * xv[it];
                db[it] += dv[it];
            }
        }
        const float m1 = wave_sum(a) * invD, m2 = wave_sum(b) * invD;
        f32x4 och[CHAIN ? ITERS : 1];
        if constexpr (CHAIN) {
            // stage 1's result stays in registers and becomes stage 2's dy; then the same four reductions on the second norm's input row
            f32x4 d1[ITERS], x2[ITERS];
            float s2 = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = (lane + 64 * it) * 4;
                const bool in = c < D;
                d1[it] = in ? (gm[it] * dv[it] - m1 - xv[it] * m2) * rstd + ra[rr][it] : z4f;
                x2[it] = in ? xb[rr][it] : z4f;
                s2 += (x2[it].x + x2[it].y) + (x2[it].z + x2[it].w);
            }
            const float mean2 = wave_sum(s2) * invD;
            float q2 = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < D) {
                    const f32x4 d = x2[it] - mean2;
                    q2 += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
            }
            const float rstd2 = 1.0f / sqrtf(wave_sum(q2) * invD + eps);
            float a2 = 0.f, b2 = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = (lane + 64 * it) * 4;
                if (c < D) {
                    x2[it] = (x2[it] - mean2) * rstd2;
                    const f32x4 gy = gm2[it] * d1[it];
                    a2 += (gy.x + gy.y) + (gy.z + gy.w);
                    const f32x4 t = gy * x2[it];
                    b2 += (t.x + t.y) + (t.z + t.w);
                    dg2[it] += d1[it] * x2[it];
                    db2[it] += d1[it];
                }
            }
            const float n1 = wave_sum(a2) * invD, n2 = wave_sum(b2) * invD;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) och[it] = (gm2[it] * d1[it] - n1 - x2[it] * n2) * rstd2;
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (lane + 64 * it) * 4;
            if (c < D) {
                f32x4 o;
                if constexpr (CHAIN) {
                    o = och[it];
                } else {
                    o = (gm[it] * dv[it] - m1 - xv[it] * m2) * rstd;
                    o += ra[rr][it];
                }
                *(f32x4*)(dx + row * D + c) = o;
                if (o2.y) {
                    f32x4 t;
                    const bool live2 = o2.mask ? o2.mask[row] != 0 : true;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float u = live2 ? o[e] * o2.alpha : 0.f;
                        if (o2.d1.thresh) u = cfm_drop(o2.d1, (unsigned)(row * D + c + e), u);
                        if (o2.d2.thresh) u = cfm_drop(o2.d2, (unsigned)(row * D + c + e), u);
                        t[e] = u;
                    }
                    store4(o2.y, o2.dt, row * D + c, t);
                }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (lane + 64 * it) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave][0][c + e] = dg[it][e];
            red[wave][1][c + e] = db[it][e];
            if constexpr (CHAIN) {
                red[wave][2][c + e] = dg2[it][e];
                red[wave][3][c + e] = db2[it][e];
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < (CHAIN ? 4 : 2) * D; j += 64 * LNB_WAVES) {
        const int which = j / D, c = j - which * D;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < LNB_WAVES; wv += 4) v += (red[wv][which][c] + red[wv + 1][which][c]) + (red[wv + 2][which][c] + red[wv + 3][which][c]);
        if constexpr (CHAIN) {
            unsafeAtomicAdd((which == 0 ? acc_g : which == 1 ? acc_b : which == 2 ? ch.acc_g : ch.acc_b) + c, v);
        } else {
            if (acc_g) unsafeAtomicAdd((which ? acc_b : acc_g) + c, v);  // one pass: the workgroups meet in the (caller-zeroed or running) sums
            else ws[((int64_t)blockIdx.x * 2 + which) * D + c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// GLU backward on the interleaved layout of the pointwise-conv-1 GEMM (column blk*32 + half*16 + i <-> output column blk*16 + i)
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void cfm_glu_bwd_kernel(const void* __restrict__ u, int u_dt, const void* __restrict__ dg, int dg_dt, void* __restrict__ du, int du_dt,
                                   int64_t M, int D) {
    const int qpr = D / 4;                                  // 4-column groups per output row
    const int64_t n = M * qpr;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = id / qpr;
        const int oc = (int)(id - row * qpr) * 4;           // output column
        const int blk = oc >> 4, i = oc & 15;
        const int64_t ua = row * 2 * D + blk * 32 + i;
        const f32x4 a = load4(u, u_dt, ua), gt = load4(u, u_dt, ua + 16), d = load4(dg, dg_dt, row * D + oc);
        f32x4 da, dgt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sg = sigmoidf_(gt[e]);
            da[e] = d[e] * sg;
            dgt[e] = d[e] * a[e] * sg * (1.f - sg);
        }
        store4(du, du_dt, ua, da);
        store4(du, du_dt, ua + 16, dgt);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// depthwise conv + BatchNorm (batch statistics) + SiLU, forward and backward.  Workgroup = (16 frames, one utterance), thread =
// channel (two when D > 256); a 16 + k - 1 frame window per channel lives in registers.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int DWT = 16;
constexpr int DWK = 15;                                     // taps (the only kernel size on the path, encoder.py:38 kernel_size=15)
constexpr int DWW = DWT + DWK - 1;

// The micro-batches of a training window (cfm_train_group): every kernel of the family takes the whole table, so a window is ONE launch per
// stage whatever the number of micro-batches -- each has its own length T, its own BatchNorm statistics (stats + gi*4*D, coef + gi*2*D)
// and its own range of partial-sum slots; a single (B, T) problem is a table of one.
constexpr int DW_GROUPS_MAX = 8;
struct DwGroup {
    int B, T, nblk_t;          // utterances, frames, 16-frame time blocks per utterance
    int blk0;                  // first (utterance, time block) slot of the group:  sum over earlier groups of B * nblk_t   (= its first workgroup)
    int bnb0;                  // first BNB_ROWS-row block of the group (backward sums):  sum over earlier groups of ceil(B*T / BNB_ROWS)
    int64_t row0;              // first row of the group in the window's [M, D] row matrices
};
struct DwGroups {
    DwGroup g[DW_GROUPS_MAX];
    int n;
};

__device__ __forceinline__ int dw_pick_blk(const DwGroups& G, int wg) {
    int idx = 0;
#pragma unroll
    for (int i = 1; i < DW_GROUPS_MAX; ++i)
        if (i < G.n && wg >= G.g[i].blk0) idx = i;          // uniform
    return idx;
}

template <int GDT>
__global__ __launch_bounds__(256) void cfm_dwconv_stats_kernel(const void* __restrict__ g, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ c_out, float* __restrict__ ws, const DwGroups G, int D) {
    const int gi = dw_pick_blk(G, (int)blockIdx.x);
    const DwGroup& gr = G.g[gi];
    const int rel = (int)blockIdx.x - gr.blk0, T = gr.T;
    const int b = rel / gr.nblk_t, t0 = (rel % gr.nblk_t) * DWT;
    const int64_t ub = (gr.row0 + (int64_t)b * T) * D;
    for (int c = threadIdx.x; c < D; c += 256) {
        float win[DWW], wk[DWK];
#pragma unroll
        for (int i = 0; i < DWW; ++i) {
            const int t = t0 - (DWK - 1) / 2 + i;
            const bool in = t >= 0 && t < T;
            const float gv = ld_t<GDT>(g, ub + (int64_t)(in ? t : 0) * D + c);      // unconditional load, selected afterwards
            win[i] = in ? gv : 0.f;
        }
#pragma unroll
        for (int k = 0; k < DWK; ++k) wk[k] = w[c * DWK + k];
        const float bs = bias[c];
        float mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int j = 0; j < DWT; ++j) {
            const int t = t0 + j;
            if (t < T) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < DWK; ++k) a = fmaf(wk[k], win[j + k], a);
                a += bs;
                c_out[ub + (int64_t)t * D + c] = a;
                const float dlt = a - mean;                 // Welford
                mean += dlt / (float)(j + 1);
                m2 += dlt * (a - mean);
            }
        }
        const int64_t blk = blockIdx.x;                     // = blk0 + b * nblk_t + time block
        ws[(blk * 2 + 0) * D + c] = mean;
        ws[(blk * 2 + 1) * D + c] = m2;
    }
}

// stats[0..3][D] = mean, rstd, scale = gamma*rstd, shift = beta - mean*scale; running statistics updated (torch: unbiased variance)
// one wavefront per channel: lane l combines partials l, l+64, ... (Chan's pairwise rule, fp64), then the 64 lane results are merged by a
// butterfly -- the serial chain is (B * nblk_t)/64 + 6 combinations instead of B * nblk_t (36 us at config 3 as one thread per channel).
// Micro-batch after micro-batch INSIDE the wavefront: the running statistics take the momentum updates in the order of the forward passes
// they stand for (the reference: one forward per micro-batch).
__device__ __forceinline__ void chan_merge(double& n, double& mean, double& m2, double nb, double mb, double qb) {
    if (nb <= 0.0) return;
    const double d = mb - mean, nn = n + nb;
    mean += d * nb / nn;
    m2 += qb + d * d * n * nb / nn;
    n = nn;
}
__device__ __forceinline__ double shfl_xor_f64(double v, int o) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __shfl_xor(lo, o, 64);
    hi = __shfl_xor(hi, o, 64);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
__global__ __launch_bounds__(256) void cfm_bn_finalize_kernel(const float* __restrict__ ws, const DwGroups G, int D, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
                                                              float* __restrict__ stats_all) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= D) return;                                     // wave-uniform
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int gi = 0; gi < G.n; ++gi) {
        const DwGroup& gr = G.g[gi];
        const int nblk_t = gr.nblk_t, T = gr.T;
        double n = 0.0, mean = 0.0, m2 = 0.0;
        const int total = gr.B * nblk_t;
        for (int blk = lane; blk < total; blk += 64) {
            const int tb = blk % nblk_t;
            const int64_t slot = gr.blk0 + blk;
            chan_merge(n, mean, m2, (double)min(DWT, T - tb * DWT), ws[(slot * 2 + 0) * D + c], ws[(slot * 2 + 1) * D + c]);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double on = shfl_xor_f64(n, o), om = shfl_xor_f64(mean, o), oq = shfl_xor_f64(m2, o);
            // merge the partner's triple; both lanes of a pair compute the same symmetric result
            const double nn = n + on;
            if (nn > 0.0) {
                const double d = om - mean;
                m2 = m2 + oq + d * d * n * on / nn;
                mean = (n * mean + on * om) / nn;
            }
            n = nn;
        }
        if (lane == 0) {
            float* stats = stats_all + (int64_t)gi * 4 * D;
            const double var = m2 / n;
            const float rstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = gamma[c] * rstd;
            stats[c] = (float)mean;
            stats[D + c] = rstd;
            stats[2 * D + c] = sc;
            stats[3 * D + c] = beta[c] - (float)mean * sc;
            rm = (1.f - momentum) * rm + momentum * (float)mean;
            rv = (1.f - momentum) * rv + momentum * (float)(m2 / (n > 1.0 ? n - 1.0 : 1.0));
        }
    }
    if (lane != 0) return;
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
}

__global__ void cfm_bn_silu_apply_kernel(const float* __restrict__ c, const float* __restrict__ stats_all, void* __restrict__ out, int out_dt, int64_t M, int D,
                                         const DwGroups G) {
    const int qpr = D / 4;
    const int64_t n = M * qpr;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(id % qpr) * 4;
        const int64_t row = id / qpr;
        int gi = 0;
#pragma unroll
        for (int i = 1; i < DW_GROUPS_MAX; ++i)
            if (i < G.n && row >= G.g[i].row0) gi = i;
        const float* stats = stats_all + (int64_t)gi * 4 * D;
        const f32x4 v = *(const f32x4*)(c + id * 4), sc = *(const f32x4*)(stats + 2 * D + col), sh = *(const f32x4*)(stats + 3 * D + col);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = siluf_(v[e] * sc[e] + sh[e]);
        store4(out, out_dt, id * 4, o);
    }
}

// dy = ds * silu'(c*scale + shift) -> dy_out (f32); per-workgroup sums over 16 rows of dy and dy * chat -> ws[blk][2][D]
constexpr int BNB_ROWS = 16;
template <int SDT>
__global__ __launch_bounds__(256) void cfm_bn_silu_bwd_kernel(const void* __restrict__ ds, const float* __restrict__ c, const float* __restrict__ stats_all,
                                                              float* __restrict__ dy_out, float* __restrict__ ws, const DwGroups G, int D) {
    int gi = 0;
#pragma unroll
    for (int i = 1; i < DW_GROUPS_MAX; ++i)
        if (i < G.n && (int)blockIdx.x >= G.g[i].bnb0) gi = i;
    const DwGroup& gr = G.g[gi];
    const float* stats = stats_all + (int64_t)gi * 4 * D;
    const int64_t M = gr.row0 + (int64_t)gr.B * gr.T;         // one past the group's last row
    const int64_t r0 = gr.row0 + (int64_t)((int)blockIdx.x - gr.bnb0) * BNB_ROWS;
    for (int ch = threadIdx.x; ch < D; ch += 256) {
        const float mean = stats[ch], rstd = stats[D + ch], sc = stats[2 * D + ch], sh = stats[3 * D + ch];
        float s1 = 0.f, s2 = 0.f;
        float cv[BNB_ROWS], dv[BNB_ROWS];
#pragma unroll
        for (int r = 0; r < BNB_ROWS; ++r) {                 // all loads first (rows past the group: clamped address, zeroed value)
            const int64_t row = r0 + r < M ? r0 + r : M - 1;
            cv[r] = c[row * D + ch];
            dv[r] = ld_t<SDT>(ds, row * D + ch);
        }
#pragma unroll
        for (int r = 0; r < BNB_ROWS; ++r) {
            if (r0 + r < M) {
                const float dyv = dv[r] * dsilu_(cv[r] * sc + sh);
                dy_out[(r0 + r) * D + ch] = dyv;
                s1 += dyv;
                s2 += dyv * ((cv[r] - mean) * rstd);
            }
        }
        ws[((int64_t)blockIdx.x * 2 + 0) * D + ch] = s1;
        ws[((int64_t)blockIdx.x * 2 + 1) * D + ch] = s2;
    }
}

// dc = gamma*rstd * (dy - k1 - chat*k2), then the depthwise conv's backward: dg[t] = sum_k w[k] dc[t-k+7] and per-workgroup partials of
// dw[k] = sum dc[t] g[t+k-7], db = sum dc  -> ws[blk][16][D]
// GLU: the GLU backward (convolution.py:42) in the same launch -- the depthwise input gradient dg[t, ch] is rounded to the output type (what the
// separate cfm_glu_bwd would have read back) and turned into the two columns of du [M, 2D] it belongs to (value / gate blocks of 16 interleaved,
// the pointwise-conv-1 GEMM's C_pre layout); dg itself is then not stored.
template <int ODT>
__device__ __forceinline__ float round_as(float v) {
    if constexpr (ODT == CFM_F32) return v;
    else if constexpr (ODT == CFM_BF16) return BF16::to_f32((u16)(pack2<BF16>(v, 0.f) & 0xffffu));
    else return F16::to_f32((u16)(pack2<F16>(v, 0.f) & 0xffffu));
}

template <int GDT, int ODT, bool GLU>
__global__ __launch_bounds__(256) void cfm_dwconv_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ c, const float* __restrict__ stats_all,
                                                             const float* __restrict__ coef_all, const void* __restrict__ g,
                                                             const float* __restrict__ w, void* __restrict__ dg_out, float* __restrict__ ws,
                                                             const DwGroups G, int D, const void* __restrict__ u, void* __restrict__ du) {
    const int gi = dw_pick_blk(G, (int)blockIdx.x);
    const DwGroup& gr = G.g[gi];
    const float* stats = stats_all + (int64_t)gi * 4 * D;
    const float* coef = coef_all + (int64_t)gi * 2 * D;
    const int rel = (int)blockIdx.x - gr.blk0, T = gr.T;
    const int b = rel / gr.nblk_t, t0 = (rel % gr.nblk_t) * DWT;
    const int64_t ub = (gr.row0 + (int64_t)b * T) * D;
    const int64_t blk = blockIdx.x;
    for (int ch = threadIdx.x; ch < D; ch += 256) {
        const float mean = stats[ch], rstd = stats[D + ch], sc = stats[2 * D + ch];       // sc = gamma * rstd
        const float k1 = coef[ch], k2 = coef[D + ch];
        float dcw[DWW], gw[DWW], wk[DWK];
#pragma unroll
        for (int i = 0; i < DWW; ++i) {
            const int t = t0 - (DWK - 1) / 2 + i;
            const bool in = t >= 0 && t < T;
            const int64_t o = ub + (int64_t)(in ? t : 0) * D + ch;
            const float chat = (c[o] - mean) * rstd;
            dcw[i] = in ? sc * (dy[o] - k1 - chat * k2) : 0.f;
            const float gv = ld_t<GDT>(g, o);
            gw[i] = in ? gv : 0.f;
        }
#pragma unroll
        for (int k = 0; k < DWK; ++k) wk[k] = w[ch * DWK + k];
#pragma unroll
        for (int j = 0; j < DWT; ++j) {
            const int t = t0 + j;
            if (t < T) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < DWK; ++k) a = fmaf(wk[k], dcw[j + DWK - 1 - k], a);
                if constexpr (GLU) {
                    const int64_t ua = (ub / D + t) * 2 * D + (ch >> 4) * 32 + (ch & 15);
                    const float dv = round_as<ODT>(a), av = ld_t<ODT>(u, ua), gt = ld_t<ODT>(u, ua + 16);
                    const float sg = sigmoidf_(gt);
                    st_t<ODT>(du, ua, dv * sg);
                    st_t<ODT>(du, ua + 16, dv * av * sg * (1.f - sg));
                } else {
                    st_t<ODT>(dg_out, ub + (int64_t)t * D + ch, a);
                }
            }
        }
        float db = 0.f;
#pragma unroll
        for (int k = 0; k < DWK; ++k) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < DWT; ++j) a = fmaf(dcw[j + (DWK - 1) / 2], gw[j + k], a);   // frames past T have dc = 0
            ws[(blk * 16 + k) * D + ch] = a;
        }
#pragma unroll
        for (int j = 0; j < DWT; ++j) db += dcw[j + (DWK - 1) / 2];
        ws[(blk * 16 + 15) * D + ch] = db;
    }
}

// second stage of the BatchNorm backward sums, micro-batch after micro-batch: coef_g = (S1_g/N_g, S2_g/N_g); dbeta = sum_g S1_g, dgamma = sum_g S2_g
__global__ __launch_bounds__(256) void cfm_bn_bwd_finalize_kernel(const float* __restrict__ ws, const DwGroups G, int D, float* dgamma, float* dbeta, float* coef_all,
                                                                  int acc) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);                  // one wavefront per channel
    if (c >= D) return;
    float tb = acc ? dbeta[c] : 0.f, tg = acc ? dgamma[c] : 0.f;        // acc: on top of what is there (one writer per element: reproducible)
    for (int gi = 0; gi < G.n; ++gi) {
        const DwGroup& gr = G.g[gi];
        const int64_t Mg = (int64_t)gr.B * gr.T;
        const int nblk = (int)((Mg + BNB_ROWS - 1) / BNB_ROWS);
        double s1 = 0.0, s2 = 0.0;
        for (int b = lane; b < nblk; b += 64) {
            s1 += ws[((int64_t)(gr.bnb0 + b) * 2 + 0) * D + c];
            s2 += ws[((int64_t)(gr.bnb0 + b) * 2 + 1) * D + c];
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            s1 += shfl_xor_f64(s1, o);
            s2 += shfl_xor_f64(s2, o);
        }
        if (lane == 0) {
            tb += (float)s1;
            tg += (float)s2;
            coef_all[(int64_t)gi * 2 * D + c] = (float)(s1 / (double)Mg);
            coef_all[(int64_t)gi * 2 * D + D + c] = (float)(s2 / (double)Mg);
        }
    }
    if (lane != 0) return;
    dbeta[c] = tb;
    dgamma[c] = tg;
}

// second stage of the depthwise gradients: dw_w[c][k] (the layout of depthwise_conv.weight (D,1,K)) and dw_b[c]
__global__ __launch_bounds__(1024) void cfm_dwconv_bwd_finalize_kernel(const float* __restrict__ ws, int nblk, int D, float* dw_w, float* dw_b, int acc) {
    __shared__ float red[RP_Q][64];
    const int jc = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int id = blockIdx.x * 64 + jc;                                // output (slot, c), 16 threads each (fixed-order partial sums)
    const int slot = id / D, c = id - slot * D;
    float s0 = 0.f;
    if (id < 16 * D)
        for (int b = q; b < nblk; b += RP_Q) s0 += ws[((int64_t)b * 16 + slot) * D + c];
    red[q][jc] = s0;
    __syncthreads();
    if (q == 0 && id < 16 * D) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RP_Q; ++i) s += red[i][jc];
        float* o = slot < DWK ? dw_w + c * DWK + slot : dw_b + c;
        *o = (acc ? *o : 0.f) + s;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// front-end backward helpers
// ------------------------------------------------------------------------------------------------------------------------------
// dh1[b,t1,f1,c] = (h1 > 0) * sum_{kt,kf : (t1-kt), (f1-kf) even and in range} dcol[(b,(t1-kt)/2,(f1-kf)/2), (kt*3+kf)*C + c]
// -- the transpose of the 3x3 stride-2 im2col as a gather (<= 4 terms), 8 channels per thread
__global__ void cfm_col2im_relu_bwd_kernel(const void* __restrict__ dcol, int dc_dt, const void* __restrict__ h1, int h_dt, void* __restrict__ dh1, int o_dt,
                                           int B, int T1, int F1, int T2, int F2, int C) {
    const int c4n = C / 4;
    const int64_t n = (int64_t)B * T1 * F1 * c4n;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(id % c4n) * 4;
        int64_t r = id / c4n;
        const int f1 = (int)(r % F1);
        r /= F1;
        const int t1 = (int)(r % T1);
        const int b = (int)(r / T1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int tt = t1 - kt;
            if (tt < 0 || (tt & 1) || (tt >> 1) >= T2) continue;
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int ff = f1 - kf;
                if (ff < 0 || (ff & 1) || (ff >> 1) >= F2) continue;
                const int64_t m = ((int64_t)b * T2 + (tt >> 1)) * F2 + (ff >> 1);
                acc += load4(dcol, dc_dt, m * (9 * (int64_t)C) + (kt * 3 + kf) * C + c);
            }
        }
        const f32x4 hv = load4(h1, h_dt, id * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = hv[e] > 0.f ? acc[e] : 0.f;
        store4(dh1, o_dt, id * 4, acc);
    }
}

// first convolution: dw1[tap][c] = sum_{b,t1,f1} dh1[b,t1,f1,c] * xin[b,2t1+kt,2f1+kf], db1[c] = sum dh1 -- workgroup = (8 output rows t1 of
// one utterance), thread = channel; the 3 input rows of each t1 go through LDS (every thread reads the same taps)
constexpr int C1_TB = 8;
template <int DDT>
__global__ __launch_bounds__(256) void cfm_conv1_wgrad_kernel(const void* __restrict__ dh1, const float* __restrict__ x, const float* __restrict__ cm,
                                                              const float* __restrict__ ci, float* __restrict__ ws, int T, int F, int T1, int F1, int C) {
    extern __shared__ float xr[];                           // [3][F]
    const int b = blockIdx.y, tb = blockIdx.x * C1_TB;
    const int64_t blk = (int64_t)b * gridDim.x + blockIdx.x;
    float acc0[10], acc1[10];                               // channels tid and tid + 256 (C <= 512)
#pragma unroll
    for (int i = 0; i < 10; ++i) acc0[i] = acc1[i] = 0.f;
    for (int dt = 0; dt < C1_TB; ++dt) {
        const int t1 = tb + dt;
        if (t1 >= T1) break;                                // uniform
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * F; i += 256) {
            const int rr = i / F, f = i - rr * F;
            float v = x[((int64_t)b * T + 2 * t1 + rr) * F + f];
            if (cm) v -= cm[f];                             // global CMVN folded into the first convolution (cmvn.py:22-33)
            if (ci) v *= ci[f];
            xr[i] = v;
        }
        __syncthreads();
        // 8 positions per trip, their gradient values requested together and unconditionally (clamped index, selected afterwards): one value
        // per trip behind a run-time dtype branch made every one of the T1-row's F1 loads wait out its own latency (157 us per launch)
        const int c0 = (int)threadIdx.x < C ? (int)threadIdx.x : 0, c1 = (int)threadIdx.x + 256 < C ? (int)threadIdx.x + 256 : 0;
        const bool two = C > 256;
        for (int fb = 0; fb < F1; fb += 8) {
            float d0[8], d1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f1 = fb + j < F1 ? fb + j : F1 - 1;
                const int64_t o = (((int64_t)b * T1 + t1) * F1 + f1) * C;
                d0[j] = ld_t<DDT>(dh1, o + c0);
                d1[j] = two ? ld_t<DDT>(dh1, o + c1) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (fb + j < F1) {                           // uniform
                    const int f1 = fb + j;
                    float taps[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) taps[k] = xr[(k / 3) * F + 2 * f1 + (k % 3)];
                    const float e0 = (int)threadIdx.x < C ? d0[j] : 0.f, e1 = (int)threadIdx.x + 256 < C ? d1[j] : 0.f;
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        acc0[k] = fmaf(e0, taps[k], acc0[k]);
                        acc1[k] = fmaf(e1, taps[k], acc1[k]);
                    }
                    acc0[9] += e0;
                    acc1[9] += e1;
                }
            }
        }
    }
    if ((int)threadIdx.x < C)
#pragma unroll
        for (int k = 0; k < 10; ++k) ws[(blk * 10 + k) * C + threadIdx.x] = acc0[k];
    if ((int)threadIdx.x + 256 < C)
#pragma unroll
        for (int k = 0; k < 10; ++k) ws[(blk * 10 + k) * C + threadIdx.x + 256] = acc1[k];
}

// ------------------------------------------------------------------------------------------------------------------------------
// optimizer: Adam over flat buffers with the clip coefficient read from device memory (no host sync between backward and step)
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void cfm_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                float b1, float b2, float eps, float weight_decay, float bc1, float bc2_sqrt, const float* __restrict__ gscale) {
    const float gs = gscale ? *gscale : 1.f;
    const int64_t n4 = n / 4;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n4; id += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pv = *(const f32x4*)(p + id * 4), gv = *(const f32x4*)(g + id * 4) * gs, mv = *(const f32x4*)(m + id * 4), vv = *(const f32x4*)(v + id * 4);
        gv += pv * weight_decay;
        mv = mv * b1 + gv * (1.f - b1);
        vv = vv * b2 + gv * gv * (1.f - b2);
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] -= (lr / bc1) * mv[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
        *(f32x4*)(p + id * 4) = pv;
        *(f32x4*)(m + id * 4) = mv;
        *(f32x4*)(v + id * 4) = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {         // tail
        const int64_t i = n4 * 4 + threadIdx.x;
        const float gq = g[i] * gs + p[i] * weight_decay;
        const float mq = m[i] * b1 + gq * (1.f - b1), vq = v[i] * b2 + gq * gq * (1.f - b2);
        p[i] -= (lr / bc1) * mq / (sqrtf(vq) / bc2_sqrt + eps);
        m[i] = mq;
        v[i] = vq;
    }
}

// The same update with the step's scalar glue inside: the gradient scale is computed from the device-side sum of squares (clip by the global norm
// of the AVERAGED gradient, torch's clip_grad_norm_ rule: min(1, clip / (norm + 1e-6)), times 1/world), the norm is written out for logging, and
// the gradient buffer is zeroed as it is read -- seven scalar launches and a 139 MB fill less per optimizer step.
__global__ void cfm_adam_clip_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                     float b1, float b2, float eps, float weight_decay, float bc1, float bc2_sqrt, const float* __restrict__ sumsq, float clip,
                                     float inv_world, int zero_grad, float* __restrict__ norm_out) {
    float gs = inv_world;
    if (sumsq) {
        const float norm = sqrtf(*sumsq) * inv_world;
        if (clip > 0.f) gs = fminf(clip / (norm + 1e-6f), 1.0f) * inv_world;
        if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
    }
    const int64_t n4 = n / 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n4; id += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pv = *(const f32x4*)(p + id * 4), gv = *(const f32x4*)(g + id * 4) * gs, mv = *(const f32x4*)(m + id * 4), vv = *(const f32x4*)(v + id * 4);
        gv += pv * weight_decay;
        mv = mv * b1 + gv * (1.f - b1);
        vv = vv * b2 + gv * gv * (1.f - b2);
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] -= (lr / bc1) * mv[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
        *(f32x4*)(p + id * 4) = pv;
        *(f32x4*)(m + id * 4) = mv;
        *(f32x4*)(v + id * 4) = vv;
        if (zero_grad) *(f32x4*)(g + id * 4) = z4;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {         // tail
        const int64_t i = n4 * 4 + threadIdx.x;
        const float gq = g[i] * gs + p[i] * weight_decay;
        const float mq = m[i] * b1 + gq * (1.f - b1), vq = v[i] * b2 + gq * gq * (1.f - b2);
        p[i] -= (lr / bc1) * mq / (sqrtf(vq) / bc2_sqrt + eps);
        m[i] = mq;
        v[i] = vq;
        if (zero_grad) g[i] = 0.f;
    }
}

__global__ __launch_bounds__(256) void cfm_sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n4; id += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(x + id * 4);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = x[n4 * 4 + threadIdx.x];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void cfm_dropout_rows_kernel(const void* __restrict__ x, int x_dt, void* __restrict__ y, int y_dt, const uint8_t* __restrict__ mask, float alpha,
                                        CfmDrop d1, CfmDrop d2, int64_t M, int N) {
    const int qpr = N / 4;
    const int64_t n = M * qpr;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = id / qpr;
        f32x4 v = load4(x, x_dt, id * 4);
        const bool keep = mask ? mask[row] != 0 : true;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = keep ? v[e] * alpha : 0.f;
            if (d1.thresh) t = cfm_drop(d1, (unsigned)(id * 4 + e), t);
            if (d2.thresh) t = cfm_drop(d2, (unsigned)(id * 4 + e), t);
            v[e] = t;
        }
        store4(y, y_dt, id * 4, v);
    }
}

__global__ void cfm_dropout_mask_kernel(uint8_t* __restrict__ out, int64_t n, CfmDrop d) {
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (int64_t)gridDim.x * blockDim.x)
        out[id] = cfm_hash32(d.seed, (unsigned)id) >= d.thresh ? 1 : 0;
}

// Weight packs of the training path in one launch: a table of jobs, each "gather N source rows of K f32 values (a device array of row
// pointers: concatenations and row permutations are just pointer tables) into the 16-bit matrix dst [N,K] and its transpose dst_t [K,N]"
// (+ the bf16 lo planes of the f32-accurate mode).  One 64 x 64 tile per workgroup through an LDS tile, so both outputs are written in
// 16-byte pieces along their own fast axis.  Job = 8 x int64: rows, N, K, dst, dst_lo, dst_t, dst_t_lo, first tile.
__global__ __launch_bounds__(256) void cfm_pack_kernel(const int64_t* __restrict__ jobs, int n_jobs, int dt, int split) {
    __shared__ float tile[64][65];
    int j = 0, hi = n_jobs;                                                           // last job whose first tile is <= blockIdx.x (uniform)
    while (hi - j > 1) {
        const int mid = (j + hi) >> 1;
        if (jobs[mid * 8 + 7] <= (int64_t)blockIdx.x) j = mid; else hi = mid;
    }
    const int64_t* job = jobs + j * 8;
    const float* const* rows = (const float* const*)job[0];
    const int N = (int)job[1], K = (int)job[2];
    u16 *dst = (u16*)job[3], *dst_lo = (u16*)job[4], *dst_t = (u16*)job[5], *dst_t_lo = (u16*)job[6];
    const int t = (int)((int64_t)blockIdx.x - job[7]);
    const int tiles_k = (K + 63) / 64;
    const int n0 = (t / tiles_k) * 64, k0 = (t % tiles_k) * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = i * 256 + tid, r = id >> 4, c4 = (id & 15) * 4;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (n0 + r < N && k0 + c4 < K) v = *(const f32x4*)(rows[n0 + r] + k0 + c4);       // K % 4 == 0
        tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    }
    __syncthreads();
    auto emit = [&](u16* hi_p, u16* lo_p, const f32x4& a, const f32x4& b) {
        if (split) {
            u32x4 hi, lo;
            split8(a, b, hi, lo);
            *(u32x4*)hi_p = hi;
            *(u32x4*)lo_p = lo;
        } else {
            *(u32x4*)hi_p = dt == CFM_BF16 ? pack8<BF16>(a, b) : pack8<F16>(a, b);
        }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = i * 256 + tid, r = id >> 3, c8 = (id & 7) * 8;
        if (dst && n0 + r < N && k0 + c8 < K) {                                             // K % 8 == 0
            const f32x4 a = {tile[r][c8], tile[r][c8 + 1], tile[r][c8 + 2], tile[r][c8 + 3]};
            const f32x4 b = {tile[r][c8 + 4], tile[r][c8 + 5], tile[r][c8 + 6], tile[r][c8 + 7]};
            const int64_t o = (int64_t)(n0 + r) * K + k0 + c8;
            emit(dst + o, dst_lo + o, a, b);
        }
        if (dst_t && k0 + r < K && n0 + c8 < N) {                                           // N % 8 == 0; row r of the transpose = column k0 + r
            const f32x4 a = {tile[c8][r], tile[c8 + 1][r], tile[c8 + 2][r], tile[c8 + 3][r]};
            const f32x4 b = {tile[c8 + 4][r], tile[c8 + 5][r], tile[c8 + 6][r], tile[c8 + 7][r]};
            const int64_t o = (int64_t)(k0 + r) * N + n0 + c8;
            emit(dst_t + o, dst_t_lo + o, a, b);
        }
    }
}

// Small f32 vectors of the training packs in one launch for a whole stack: out[i] = *a[i] + (b[i] ? *b[i] : 0) through two tables of element
// pointers -- the fused q|k|v bias (linear_q.bias + pos_bias_u | linear_k.bias | linear_v.bias, attention.py:62-64,81) and the GLU-interleaved
// pointwise-conv-1 bias are gathers of parameter elements whose addresses never move (the optimizer writes in place).
__global__ void cfm_pack_vectors_kernel(const float* const* __restrict__ a, const float* const* __restrict__ b, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* pb = b[i];
    out[i] = *a[i] + (pb ? *pb : 0.f);
}

inline int grid_for(int64_t n, int per_block = 256, int cap = 4096) {
    int64_t b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

// ================================================================================================================================
extern "C" int64_t cfm_layernorm_bwd_ws(int64_t M, int32_t D) { return ((M + LNB_ROWS - 1) / LNB_ROWS) * 2 * (int64_t)D; }

static int layernorm_bwd_impl(const float* x, const void* dy, int32_t dy_dtype, const float* gamma, const uint8_t* row_mask, const float* dres, float* dx,
                              float* dgamma, float* dbeta, float* ws, bool accumulate, const LnBwd2& o2, float eps, int64_t M, int32_t D, hipStream_t s,
                              const LnChain* chain = nullptr) {
    const int nblk = (int)((M + LNB_ROWS - 1) / LNB_ROWS);
    float *ag = accumulate ? dgamma : nullptr, *ab = accumulate ? dbeta : nullptr;
    {
        CfmProfScope prof("layernorm_bwd", s, 0.0, (double)M * D * (8.0 + cfm_elt_size(dy_dtype) + (dres ? 4 : 0) + (o2.y ? cfm_elt_size(o2.dt) : 0)));
        const dim3 grid((unsigned)nblk), block(64 * LNB_WAVES);
        LnChain ch = {};
        if (chain) ch = *chain;
#define CFM_LNB(IT, DT)                                                                                                                                    \
    do {                                                                                                                                                   \
        if (chain) CFM_LAUNCH((cfm_layernorm_bwd_kernel<IT, DT, true>), grid, block, 0, s, x, dy, dy_dtype, gamma, row_mask, dres, dx, ws, eps, M, D, ag, ab, o2, ch); \
        else CFM_LAUNCH((cfm_layernorm_bwd_kernel<IT, DT, false>), grid, block, 0, s, x, dy, dy_dtype, gamma, row_mask, dres, dx, ws, eps, M, D, ag, ab, o2, ch);  \
    } while (0)
#define CFM_LNB_DT(IT)                                        \
    do {                                                      \
        if (dy_dtype == CFM_F32) CFM_LNB(IT, CFM_F32);        \
        else if (dy_dtype == CFM_BF16) CFM_LNB(IT, CFM_BF16); \
        else CFM_LNB(IT, CFM_F16);                            \
    } while (0)
        if (D <= 256) CFM_LNB_DT(1);
        else if (D <= 512) CFM_LNB_DT(2);
        else CFM_LNB_DT(4);
#undef CFM_LNB_DT
#undef CFM_LNB
        if (int rc = cfm_launch_status("cfm_layernorm_bwd")) return rc;
    }
    if (accumulate) return CFM_OK;
    return reduce_partials(ws, nblk, 2 * D, D, 1.0f, dgamma, dbeta, s, "cfm_layernorm_bwd (reduce)");
}

extern "C" int cfm_layernorm_bwd(const float* x, const void* dy, int32_t dy_dtype, const float* gamma, const uint8_t* row_mask, const float* dres,
                                 float* dx, float* dgamma, float* dbeta, float* ws, float eps, int64_t M, int32_t D, cfm_stream_t stream) {
    CFM_CHECK_ARG(x && dy && gamma && dx && dgamma && dbeta && ws, "cfm_layernorm_bwd: null pointer");
    CFM_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "cfm_layernorm_bwd: need D %% 4 == 0 and D <= 1024 (M=%lld D=%d)", (long long)M, D);
    CFM_CHECK_ARG(dy_dtype >= CFM_F32 && dy_dtype <= CFM_F16, "cfm_layernorm_bwd: bad dy dtype");
    LnBwd2 none = {};
    return layernorm_bwd_impl(x, dy, dy_dtype, gamma, row_mask, dres, dx, dgamma, dbeta, ws, false, none, eps, M, D, (hipStream_t)stream);
}

extern "C" int cfm_layernorm_bwd_fused(const cfm_ln_bwd_desc* d, cfm_stream_t stream) {
    CFM_CHECK_ARG(d && d->x && d->dy && d->gamma && d->dx && d->dgamma && d->dbeta && (d->accumulate || d->ws), "cfm_layernorm_bwd_fused: null pointer");
    CFM_CHECK_ARG(d->M > 0 && d->D > 0 && d->D % 4 == 0 && d->D <= 1024, "cfm_layernorm_bwd_fused: need D %% 4 == 0 and D <= 1024");
    CFM_CHECK_ARG(d->dy_dtype >= CFM_F32 && d->dy_dtype <= CFM_F16 && (!d->dx2 || (d->dx2_dtype >= CFM_F32 && d->dx2_dtype <= CFM_F16)), "cfm_layernorm_bwd_fused: bad dtype");
    CFM_CHECK_ARG(d->p1 >= 0.f && d->p1 < 1.f && d->p2 >= 0.f && d->p2 < 1.f && d->M * d->D < ((int64_t)1 << 32), "cfm_layernorm_bwd_fused: p in [0,1), fewer than 2^32 elements");
    LnBwd2 o2 = {};
    if (d->dx2) { o2.y = d->dx2; o2.dt = d->dx2_dtype; o2.alpha = d->alpha2; o2.d1 = cfm_make_drop(d->p1, d->seed1); o2.d2 = cfm_make_drop(d->p2, d->seed2); o2.mask = d->dx2_row_mask; }
    LnChain ch = {};
    if (d->chain_x) {
        CFM_CHECK_ARG(d->chain_gamma && d->chain_dgamma && d->chain_dbeta && d->accumulate, "cfm_layernorm_bwd_fused: a chained second norm needs its gamma / gradient pointers and accumulate = 1");
        ch.x = d->chain_x; ch.gamma = d->chain_gamma; ch.acc_g = d->chain_dgamma; ch.acc_b = d->chain_dbeta;
    }
    return layernorm_bwd_impl(d->x, d->dy, d->dy_dtype, d->gamma, d->row_mask, d->dres, d->dx, d->dgamma, d->dbeta, d->ws, d->accumulate != 0, o2, d->eps, d->M,
                              d->D, (hipStream_t)stream, d->chain_x ? &ch : nullptr);
}

extern "C" int cfm_glu_bwd(const void* u, int32_t u_dtype, const void* dg, int32_t dg_dtype, void* du, int32_t du_dtype, int64_t M, int32_t D,
                           cfm_stream_t stream) {
    CFM_CHECK_ARG(u && dg && du, "cfm_glu_bwd: null pointer");
    CFM_CHECK_ARG(M > 0 && D > 0 && D % 16 == 0, "cfm_glu_bwd: need D %% 16 == 0 (M=%lld D=%d)", (long long)M, D);
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("glu_bwd", s, 0.0, (double)M * D * (2.0 * cfm_elt_size(u_dtype) + cfm_elt_size(dg_dtype) + 2.0 * cfm_elt_size(du_dtype)));
    CFM_LAUNCH(cfm_glu_bwd_kernel, dim3((unsigned)grid_for(M * (D / 4))), dim3(256), 0, s, u, u_dtype, dg, dg_dtype, du, du_dtype, M, D);
    return cfm_launch_status("cfm_glu_bwd");
}

extern "C" int64_t cfm_dwconv_bn_ws(int32_t B, int32_t T, int32_t D) {     // floats: enough for the forward and for the backward
    const int64_t nblk_t = (T + DWT - 1) / DWT;
    // backward: BatchNorm sums [nb][2][D] | coef [2][D] | depthwise partials [B*nblk_t][16][D]   (forward: [B*nblk_t][2][D]).  A window of
    // several micro-batches needs the SUM of its groups' sizes (the three regions of all groups laid out region by region).
    return (((int64_t)B * T + BNB_ROWS - 1) / BNB_ROWS) * 2 * D + 2 * (int64_t)D + (int64_t)B * nblk_t * 16 * D;
}

namespace {
// the kernels' group table from the C one; totals: (utterance, time block) slots, BNB row blocks, rows
int dw_groups(const cfm_train_group* groups, int n, int D, DwGroups& G, int& blks, int& bnbs, int64_t& rows, const char* who) {
    CFM_CHECK_ARG(groups && n > 0 && n <= DW_GROUPS_MAX, "%s: %d row groups (1 .. %d)", who, n, DW_GROUPS_MAX);
    CFM_CHECK_ARG(D > 0 && D % 4 == 0 && D <= 512, "%s: need D %% 4 == 0, D <= 512 (D=%d)", who, D);
    blks = 0; bnbs = 0; rows = 0;
    G.n = n;
    for (int i = 0; i < n; ++i) {
        const cfm_train_group& g = groups[i];
        CFM_CHECK_ARG(g.B > 0 && g.T > 0 && g.B <= 65535 && g.row0 == rows, "%s: group %d (B=%d T=%d row0=%lld) must start at row %lld", who, i, g.B, g.T,
                      (long long)g.row0, (long long)rows);
        DwGroup& o = G.g[i];
        o.B = g.B; o.T = g.T; o.nblk_t = (g.T + DWT - 1) / DWT; o.blk0 = blks; o.bnb0 = bnbs; o.row0 = g.row0;
        blks += g.B * o.nblk_t;
        bnbs += (int)(((int64_t)g.B * g.T + BNB_ROWS - 1) / BNB_ROWS);
        rows += (int64_t)g.B * g.T;
    }
    for (int i = n; i < DW_GROUPS_MAX; ++i) G.g[i] = G.g[0];
    return CFM_OK;
}
}  // namespace

extern "C" int cfm_dwconv_bn_train_groups(const void* g, int32_t g_dtype, const float* w, const float* dw_bias, const float* gamma, const float* beta,
                                          float* running_mean, float* running_var, float momentum, float eps, float* c_out, float* stats, void* s_out,
                                          int32_t s_dtype, float* ws, const cfm_train_group* groups, int32_t n_groups, int32_t D, int32_t ktaps,
                                          cfm_stream_t stream) {
    CFM_CHECK_ARG(g && w && dw_bias && gamma && beta && c_out && stats && s_out && ws, "cfm_dwconv_bn_train: null pointer");
    CFM_CHECK_ARG(ktaps == DWK, "cfm_dwconv_bn_train: %d taps (only %d is built: the conformer's kernel_size)", ktaps, DWK);
    DwGroups G;
    int blks, bnbs;
    int64_t M;
    if (int rc = dw_groups(groups, n_groups, D, G, blks, bnbs, M, "cfm_dwconv_bn_train")) return rc;
    hipStream_t s = (hipStream_t)stream;
    {
        CfmProfScope prof("dwconv_stats", s, 2.0 * M * D * DWK, (double)M * D * (cfm_elt_size(g_dtype) + 4.0));
        const dim3 grid((unsigned)blks);
        if (g_dtype == CFM_BF16) CFM_LAUNCH((cfm_dwconv_stats_kernel<CFM_BF16>), grid, dim3(256), 0, s, g, w, dw_bias, c_out, ws, G, D);
        else if (g_dtype == CFM_F16) CFM_LAUNCH((cfm_dwconv_stats_kernel<CFM_F16>), grid, dim3(256), 0, s, g, w, dw_bias, c_out, ws, G, D);
        else CFM_LAUNCH((cfm_dwconv_stats_kernel<CFM_F32>), grid, dim3(256), 0, s, g, w, dw_bias, c_out, ws, G, D);
        if (int rc = cfm_launch_status("cfm_dwconv_bn_train (conv)")) return rc;
    }
    {
        CfmProfScope prof("bn_finalize", s, 0.0, (double)blks * 2 * D * 4);
        CFM_LAUNCH(cfm_bn_finalize_kernel, dim3((unsigned)((D + 3) / 4)), dim3(256), 0, s, (const float*)ws, G, D, gamma, beta, running_mean, running_var, momentum,
                   eps, stats);
        if (int rc = cfm_launch_status("cfm_dwconv_bn_train (finalize)")) return rc;
    }
    CfmProfScope prof("bn_silu_apply", s, 0.0, (double)M * D * (4.0 + cfm_elt_size(s_dtype)));
    CFM_LAUNCH(cfm_bn_silu_apply_kernel, dim3((unsigned)grid_for(M * (D / 4))), dim3(256), 0, s, (const float*)c_out, (const float*)stats, s_out, s_dtype, M, D, G);
    return cfm_launch_status("cfm_dwconv_bn_train (apply)");
}

extern "C" int cfm_dwconv_bn_train(const void* g, int32_t g_dtype, const float* w, const float* dw_bias, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, float* c_out, float* stats, void* s_out,
                                   int32_t s_dtype, float* ws, int32_t B, int32_t T, int32_t D, int32_t ktaps, cfm_stream_t stream) {
    cfm_train_group one = {};
    one.B = B; one.T = T; one.row0 = 0;
    return cfm_dwconv_bn_train_groups(g, g_dtype, w, dw_bias, gamma, beta, running_mean, running_var, momentum, eps, c_out, stats, s_out, s_dtype, ws, &one, 1, D,
                                      ktaps, stream);
}

extern "C" int cfm_dwconv_bn_train_bwd(const void* ds, int32_t ds_dtype, const float* c, const float* stats, const void* g, int32_t g_dtype, const float* w,
                                       void* dg_out, int32_t dg_dtype, float* dw_w, float* dw_b, float* dgamma, float* dbeta, float* dy_ws, float* ws,
                                       int32_t B, int32_t T, int32_t D, int32_t ktaps, cfm_stream_t stream) {
    return cfm_dwconv_bn_train_bwd_acc(ds, ds_dtype, c, stats, g, g_dtype, w, dg_out, dg_dtype, dw_w, dw_b, dgamma, dbeta, dy_ws, ws, B, T, D, ktaps, 0, stream);
}

extern "C" int cfm_dwconv_bn_train_bwd_groups(const void* ds, int32_t ds_dtype, const float* c, const float* stats, const void* g, int32_t g_dtype, const float* w,
                                              void* dg_out, int32_t dg_dtype, float* dw_w, float* dw_b, float* dgamma, float* dbeta, float* dy_ws, float* ws,
                                              const cfm_train_group* groups, int32_t n_groups, int32_t D, int32_t ktaps, int32_t accumulate, const void* glu_u,
                                              void* glu_du, cfm_stream_t stream) {
    CFM_CHECK_ARG(ds && c && stats && g && w && (dg_out || (glu_u && glu_du)) && dw_w && dw_b && dgamma && dbeta && dy_ws && ws, "cfm_dwconv_bn_train_bwd: null pointer");
    CFM_CHECK_ARG(!glu_u == !glu_du && (!glu_u || (g_dtype == dg_dtype && D % 16 == 0)), "cfm_dwconv_bn_train_bwd: the fused GLU backward needs u and du, g / dg of one dtype and D %% 16 == 0");
    CFM_CHECK_ARG(ktaps == DWK, "cfm_dwconv_bn_train_bwd: %d taps (only %d is built)", ktaps, DWK);
    DwGroups G;
    int blks, bnbs;
    int64_t M;
    if (int rc = dw_groups(groups, n_groups, D, G, blks, bnbs, M, "cfm_dwconv_bn_train_bwd")) return rc;
    hipStream_t s = (hipStream_t)stream;
    float* coef = ws + (int64_t)bnbs * 2 * D;                               // [n_groups][2][D] behind the BatchNorm partials
    {
        CfmProfScope prof("bn_silu_bwd", s, 0.0, (double)M * D * (8.0 + cfm_elt_size(ds_dtype)));
        if (ds_dtype == CFM_BF16) CFM_LAUNCH((cfm_bn_silu_bwd_kernel<CFM_BF16>), dim3((unsigned)bnbs), dim3(256), 0, s, ds, c, stats, dy_ws, ws, G, D);
        else if (ds_dtype == CFM_F16) CFM_LAUNCH((cfm_bn_silu_bwd_kernel<CFM_F16>), dim3((unsigned)bnbs), dim3(256), 0, s, ds, c, stats, dy_ws, ws, G, D);
        else CFM_LAUNCH((cfm_bn_silu_bwd_kernel<CFM_F32>), dim3((unsigned)bnbs), dim3(256), 0, s, ds, c, stats, dy_ws, ws, G, D);
        if (int rc = cfm_launch_status("cfm_dwconv_bn_train_bwd (silu/bn sums)")) return rc;
    }
    {
        CfmProfScope prof("bn_bwd_finalize", s, 0.0, (double)bnbs * 2 * D * 4);
        CFM_LAUNCH(cfm_bn_bwd_finalize_kernel, dim3((unsigned)((D + 3) / 4)), dim3(256), 0, s, (const float*)ws, G, D, dgamma, dbeta, coef, accumulate);
        if (int rc = cfm_launch_status("cfm_dwconv_bn_train_bwd (finalize)")) return rc;
    }
    float* part = coef + 2 * (int64_t)D * n_groups;                         // depthwise partials [blk][16][D] behind coef
    {
        CfmProfScope prof("dwconv_bwd", s, 4.0 * M * D * DWK, (double)M * D * (8.0 + cfm_elt_size(g_dtype) + cfm_elt_size(dg_dtype)));
        const dim3 grid((unsigned)blks);
#define CFM_DWB(GD, OD) CFM_LAUNCH((cfm_dwconv_bwd_kernel<GD, OD, false>), grid, dim3(256), 0, s, (const float*)dy_ws, c, stats, (const float*)coef, g, w, dg_out, part, G, D, nullptr, nullptr)
#define CFM_DWG(GD) CFM_LAUNCH((cfm_dwconv_bwd_kernel<GD, GD, true>), grid, dim3(256), 0, s, (const float*)dy_ws, c, stats, (const float*)coef, g, w, dg_out, part, G, D, glu_u, glu_du)
        if (glu_u && g_dtype == CFM_BF16) CFM_DWG(CFM_BF16);
        else if (glu_u && g_dtype == CFM_F16) CFM_DWG(CFM_F16);
        else if (glu_u) CFM_DWG(CFM_F32);
        else if (g_dtype == CFM_BF16 && dg_dtype == CFM_BF16) CFM_DWB(CFM_BF16, CFM_BF16);
        else if (g_dtype == CFM_F16 && dg_dtype == CFM_F16) CFM_DWB(CFM_F16, CFM_F16);
        else if (g_dtype == CFM_F32 && dg_dtype == CFM_F32) CFM_DWB(CFM_F32, CFM_F32);
        else if (g_dtype == CFM_BF16 && dg_dtype == CFM_F32) CFM_DWB(CFM_BF16, CFM_F32);
        else if (g_dtype == CFM_F16 && dg_dtype == CFM_F32) CFM_DWB(CFM_F16, CFM_F32);
        else if (g_dtype == CFM_F32 && dg_dtype == CFM_BF16) CFM_DWB(CFM_F32, CFM_BF16);
        else if (g_dtype == CFM_F32 && dg_dtype == CFM_F16) CFM_DWB(CFM_F32, CFM_F16);
        else return cfm_fail(CFM_ERR_UNSUPPORTED, "cfm_dwconv_bn_train_bwd: g / dg dtype pair %d / %d", g_dtype, dg_dtype);
#undef CFM_DWB
#undef CFM_DWG
        if (int rc = cfm_launch_status("cfm_dwconv_bn_train_bwd (conv)")) return rc;
    }
    CfmProfScope prof("dwconv_bwd_finalize", s, 0.0, (double)blks * 16 * D * 4);
    CFM_LAUNCH(cfm_dwconv_bwd_finalize_kernel, dim3((unsigned)((16 * D + 63) / 64)), dim3(64 * RP_Q), 0, s, (const float*)part, blks, D, dw_w, dw_b, accumulate);
    return cfm_launch_status("cfm_dwconv_bn_train_bwd (reduce)");
}

extern "C" int cfm_dwconv_bn_train_bwd_acc(const void* ds, int32_t ds_dtype, const float* c, const float* stats, const void* g, int32_t g_dtype, const float* w,
                                           void* dg_out, int32_t dg_dtype, float* dw_w, float* dw_b, float* dgamma, float* dbeta, float* dy_ws, float* ws,
                                           int32_t B, int32_t T, int32_t D, int32_t ktaps, int32_t accumulate, cfm_stream_t stream) {
    cfm_train_group one = {};
    one.B = B; one.T = T; one.row0 = 0;
    return cfm_dwconv_bn_train_bwd_groups(ds, ds_dtype, c, stats, g, g_dtype, w, dg_out, dg_dtype, dw_w, dw_b, dgamma, dbeta, dy_ws, ws, &one, 1, D, ktaps, accumulate,
                                          nullptr, nullptr, stream);
}

extern "C" int cfm_col2im_relu_bwd(const void* dcol, int32_t dcol_dtype, const void* h1, int32_t h1_dtype, void* dh1, int32_t dh1_dtype, int32_t B, int32_t T1,
                                   int32_t F1, int32_t C, cfm_stream_t stream) {
    CFM_CHECK_ARG(dcol && h1 && dh1, "cfm_col2im_relu_bwd: null pointer");
    CFM_CHECK_ARG(B > 0 && T1 >= 3 && F1 >= 3 && C > 0 && C % 4 == 0, "cfm_col2im_relu_bwd: bad shape B=%d T1=%d F1=%d C=%d", B, T1, F1, C);
    hipStream_t s = (hipStream_t)stream;
    const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
    const int64_t n = (int64_t)B * T1 * F1 * (C / 4);
    CfmProfScope prof("col2im_relu_bwd", s, 0.0, (double)B * T2 * F2 * 9 * C * cfm_elt_size(dcol_dtype) + (double)n * 4 * (cfm_elt_size(h1_dtype) + cfm_elt_size(dh1_dtype)));
    CFM_LAUNCH(cfm_col2im_relu_bwd_kernel, dim3((unsigned)grid_for(n, 256, 16384)), dim3(256), 0, s, dcol, dcol_dtype, h1, h1_dtype, dh1, dh1_dtype, B, T1, F1, T2, F2, C);
    return cfm_launch_status("cfm_col2im_relu_bwd");
}

extern "C" int64_t cfm_conv1_wgrad_ws(int32_t B, int32_t T, int32_t C) {
    const int T1 = (T - 3) / 2 + 1;
    return (int64_t)B * ((T1 + C1_TB - 1) / C1_TB) * 10 * C;
}

extern "C" int cfm_conv1_wgrad(const void* dh1, int32_t dh1_dtype, const float* x, const float* cmvn_mean, const float* cmvn_istd, float* dw, float* db, float* ws,
                               int32_t B, int32_t T, int32_t F, int32_t C, cfm_stream_t stream) {
    CFM_CHECK_ARG(dh1 && x && dw && db && ws, "cfm_conv1_wgrad: null pointer");
    CFM_CHECK_ARG(B > 0 && T >= 3 && F >= 3 && C > 0 && C <= 512 && B <= 65535 && F <= 4096, "cfm_conv1_wgrad: bad shape B=%d T=%d F=%d C=%d", B, T, F, C);
    hipStream_t s = (hipStream_t)stream;
    const int T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
    const int nbt = (T1 + C1_TB - 1) / C1_TB;
    {
        CfmProfScope prof("conv1_wgrad", s, 20.0 * B * T1 * F1 * C, (double)B * T1 * F1 * C * cfm_elt_size(dh1_dtype));
        const dim3 grid((unsigned)nbt, (unsigned)B);
        const size_t lds = (size_t)3 * F * 4;
        if (dh1_dtype == CFM_BF16) CFM_LAUNCH((cfm_conv1_wgrad_kernel<CFM_BF16>), grid, dim3(256), lds, s, dh1, x, cmvn_mean, cmvn_istd, ws, T, F, T1, F1, C);
        else if (dh1_dtype == CFM_F16) CFM_LAUNCH((cfm_conv1_wgrad_kernel<CFM_F16>), grid, dim3(256), lds, s, dh1, x, cmvn_mean, cmvn_istd, ws, T, F, T1, F1, C);
        else CFM_LAUNCH((cfm_conv1_wgrad_kernel<CFM_F32>), grid, dim3(256), lds, s, dh1, x, cmvn_mean, cmvn_istd, ws, T, F, T1, F1, C);
        if (int rc = cfm_launch_status("cfm_conv1_wgrad")) return rc;
    }
    return reduce_partials(ws, B * nbt, 10 * C, 9 * C, 1.0f, dw, db, s, "cfm_conv1_wgrad (reduce)");     // dw [9][C] tap-major (the packed layout), db [C]
}

extern "C" int cfm_dropout_rows(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, const uint8_t* row_mask, float alpha, float p, uint32_t seed,
                                float p2, uint32_t seed2, int64_t M, int32_t N, cfm_stream_t stream) {
    CFM_CHECK_ARG(x && y && M > 0 && N > 0 && N % 4 == 0, "cfm_dropout_rows: bad arguments (N %% 4 == 0)");
    CFM_CHECK_ARG(p >= 0.f && p < 1.f && p2 >= 0.f && p2 < 1.f && M * N < ((int64_t)1 << 32), "cfm_dropout_rows: p in [0,1), fewer than 2^32 elements");
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("dropout_rows", s, 0.0, (double)M * N * (cfm_elt_size(x_dtype) + cfm_elt_size(y_dtype)));
    CFM_LAUNCH(cfm_dropout_rows_kernel, dim3((unsigned)grid_for(M * (N / 4))), dim3(256), 0, s, x, x_dtype, y, y_dtype, row_mask, alpha, cfm_make_drop(p, seed),
               cfm_make_drop(p2, seed2), M, N);
    return cfm_launch_status("cfm_dropout_rows");
}

extern "C" int cfm_pack_matrices(const int64_t* jobs_dev, int32_t n_jobs, int64_t total_tiles, int32_t w_dtype, int32_t split, cfm_stream_t stream) {
    CFM_CHECK_ARG(jobs_dev && n_jobs > 0 && n_jobs <= 512 && total_tiles > 0 && total_tiles < ((int64_t)1 << 31), "cfm_pack_matrices: bad arguments (at most 512 jobs)");
    CFM_CHECK_ARG(split ? w_dtype == CFM_BF16 : cfm_is16(w_dtype), "cfm_pack_matrices: 16-bit destination type (split: bf16 hi/lo planes)");
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("pack_matrices", s, 0.0, (double)total_tiles * 64 * 64 * 8);
    CFM_LAUNCH(cfm_pack_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, jobs_dev, n_jobs, w_dtype, split);
    return cfm_launch_status("cfm_pack_matrices");
}

extern "C" int cfm_pack_vectors(const float* const* a, const float* const* b, float* out, int64_t n, cfm_stream_t stream) {
    CFM_CHECK_ARG(a && b && out && n > 0 && n < ((int64_t)1 << 31), "cfm_pack_vectors: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("pack_vectors", s, 0.0, (double)n * 28);
    CFM_LAUNCH(cfm_pack_vectors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, out, n);
    return cfm_launch_status("cfm_pack_vectors");
}

extern "C" int cfm_dropout_mask(uint8_t* out, int64_t n, float p, uint32_t seed, cfm_stream_t stream) {
    CFM_CHECK_ARG(out && n > 0 && n < ((int64_t)1 << 32) && p >= 0.f && p < 1.f, "cfm_dropout_mask: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("dropout_mask", s, 0.0, (double)n);
    CFM_LAUNCH(cfm_dropout_mask_kernel, dim3((unsigned)grid_for(n)), dim3(256), 0, s, out, n, cfm_make_drop(p, seed));
    return cfm_launch_status("cfm_dropout_mask");
}

extern "C" int cfm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int64_t step, const float* grad_scale, cfm_stream_t stream) {
    CFM_CHECK_ARG(p && g && m && v && n > 0 && step > 0, "cfm_adam_step: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    CfmProfScope prof("adam_step", s, 0.0, (double)n * 28);
    CFM_LAUNCH(cfm_adam_kernel, dim3((unsigned)grid_for(n / 4 + 1, 256, 8192)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2),
               grad_scale);
    return cfm_launch_status("cfm_adam_step");
}

extern "C" int cfm_adam_clip_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  int64_t step, const float* sumsq, float clip, float inv_world, int32_t zero_grad, float* norm_out, cfm_stream_t stream) {
    CFM_CHECK_ARG(p && g && m && v && n > 0 && step > 0 && inv_world > 0.f && (clip <= 0.f || sumsq), "cfm_adam_clip_step: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    CfmProfScope prof("adam_clip_step", s, 0.0, (double)n * 32);
    CFM_LAUNCH(cfm_adam_clip_kernel, dim3((unsigned)grid_for(n / 4 + 1, 256, 8192)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2),
               sumsq, clip, inv_world, zero_grad, norm_out);
    return cfm_launch_status("cfm_adam_clip_step");
}

extern "C" int cfm_sumsq(const float* x, int64_t n, float* partials, int32_t n_partials, float* out, cfm_stream_t stream) {
    CFM_CHECK_ARG(x && partials && out && n > 0 && n_partials > 0 && n_partials <= 4096, "cfm_sumsq: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    {
        CfmProfScope prof("sumsq", s, 0.0, (double)n * 4);
        CFM_LAUNCH(cfm_sumsq_kernel, dim3((unsigned)n_partials), dim3(256), 0, s, x, n, partials);
        if (int rc = cfm_launch_status("cfm_sumsq")) return rc;
    }
    return reduce_partials(partials, n_partials, 1, 1, 1.0f, out, out, s, "cfm_sumsq (reduce)");
}
