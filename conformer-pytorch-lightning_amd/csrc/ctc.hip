// ctc.hip -- CTC negative log-likelihood on top of the vocabulary projection (gfx950).
//
// replaces the part of CTCDecoder.forward after the Linear (reference src/decoder.py:20-21):
//     probs = logits.transpose(0, 1).log_softmax(2);  loss = nn.CTCLoss(reduction='sum')(probs, labels, enc_lens, label_lens)
// blank = 0, no zero_infinity (nn.CTCLoss defaults).  The logits [B, T, V] (f32, row stride ld) come from cfm_gemm; nothing of
// size B x T x V is written again: the log-softmax is applied on the fly as  lp[t, c] = logits[t, c] - lse[t].
//
// Two launches:
//   rows   one wavefront per frame (b, t < len), the whole chip busy: lse = log sum_c exp(logits[t, c]) (16-byte loads, max-shifted,
//          fp32), then the <= 2U+1 log-probabilities the recursion will need, lp[b, t, s] = logits[t, z_s] - lse, shifted by the frame's
//          maximum o_t and written to a compact work buffer [B, T, 2 Umax + 2] (last slot: o_t) (z = (blank, y1, blank, y2, ..., blank) is the blank-extended label sequence);
//   alpha  one workgroup per utterance, thread s owns state s (two states per thread when S > 256), alpha double-buffered in
//          LDS, one barrier per frame, the lp rows of the next frames already in registers (they do not depend on alpha):
//              alpha_t[s] = lp[t, s] + logaddexp(alpha_{t-1}[s], alpha_{t-1}[s-1], alpha_{t-1}[s-2] if z_s != blank and z_s != z_{s-2})
//          nll = -logaddexp(alpha_{T-1}[S-1], alpha_{T-1}[S-2])     (an impossible alignment gives +inf, as nn.CTCLoss does)
// The first version did both in one workgroup per utterance with a dependent global gather in every step: 1.16 ms at B = 32,
// T' = 249, V = 5002 (32 of 256 CUs busy, ~1 us of memory latency per frame).  fp32 throughout, as the reference computes the loss
// on probs.to(float32).
#include "cfm_common.h"

namespace {

constexpr int CTC_NT = 256;
constexpr int CTC_MAXS = 2 * CTC_NT;                       // extended states (U <= 255)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// log(exp(a) + exp(b)) on the hardware exp2/log2 units (v_exp_f32 / v_log_f32, ~1 ulp): the recursion is a serial chain of these, the
// library expf / log1pf sequences made it 181 us at T' = 249.  1 + e is in (1, 2], where log needs no special care.
__device__ __forceinline__ float logaddexp_(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + __logf(1.0f + __expf(-fabsf(a - b)));
}

__device__ __forceinline__ int ext_label(const int* __restrict__ labels, int64_t base, int s, int V) {
    const int y = (s & 1) ? labels[base + (s >> 1)] : 0;   // labels outside [0, V) cannot index a row: clamped (the reference would raise)
    return y < 0 ? 0 : (y < V ? y : V - 1);
}

// rows: grid = ceil(B*T / 4) workgroups of 4 wavefronts, wavefront = one frame
__global__ __launch_bounds__(CTC_NT) void cfm_ctc_rows_kernel(const float* __restrict__ logits, int64_t ld, int B, int T, int V,
                                                              const int* __restrict__ enc_lens, const int* __restrict__ labels, int Umax,
                                                              const int* __restrict__ label_lens, float* __restrict__ work, float* __restrict__ lse_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row_id = (int64_t)blockIdx.x * (CTC_NT / 64) + (threadIdx.x >> 6);
    if (row_id >= (int64_t)B * T) return;
    const int b = (int)(row_id / T), t = (int)(row_id % T);
    if (t >= min(max(enc_lens[b], 0), T)) return;          // frames past the utterance are never read by the recursion
    const float* row = logits + row_id * ld;
    float m = -INFINITY;
    for (int c = lane * 4; c < V; c += 256) {
        if (c + 3 < V) {
            const f32x4 v = *(const f32x4*)(row + c);
            m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        } else {
            for (int k = c; k < V; ++k) m = fmaxf(m, row[k]);
        }
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane * 4; c < V; c += 256) {              // second pass over a row that is now in L1/L2
        if (c + 3 < V) {
            const f32x4 v = *(const f32x4*)(row + c);
            sum += (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m));
        } else {
            for (int k = c; k < V; ++k) sum += expf(row[k] - m);
        }
    }
    const float lse = m + logf(wave_sum(sum));
    if (lse_out && lane == 0) lse_out[row_id] = lse;
    // The recursions run on lp'[t,s] = lp[t,s] - o_t with o_t = max_s lp[t,s]: along the likely paths lp' is ~0, so alpha' / beta' stay small
    // numbers instead of growing like -8 per frame, and fp32 keeps ~1e-6 absolute in the log domain however long the utterance
    // (unshifted: |alpha| ~ 2000 at T' = 249, one ulp = 1.2e-4, gradients off by 1e-3).  sum_t o_t is added back to the loss in fp64;
    // the state posteriors do not depend on it.  o_t sits in the row's last slot (row stride 2 Umax + 2).
    const int S = 2 * min(max(label_lens[b], 0), Umax) + 1, SW = 2 * Umax + 2;
    float lpv[(CTC_MAXS + 63) / 64];
    float omax = -INFINITY;
#pragma unroll
    for (int i = 0; i < (CTC_MAXS + 63) / 64; ++i) {
        const int s = lane + 64 * i;
        lpv[i] = s < S ? row[ext_label(labels, (int64_t)b * Umax, s, V)] - lse : -INFINITY;
        omax = fmaxf(omax, lpv[i]);
    }
    omax = wave_max(omax);
#pragma unroll
    for (int i = 0; i < (CTC_MAXS + 63) / 64; ++i) {
        const int s = lane + 64 * i;
        if (s < S) work[row_id * SW + s] = lpv[i] - omax;
    }
    if (lane == 0) work[row_id * SW + SW - 1] = omax;
}

// alpha: one workgroup per utterance
__device__ __forceinline__ void ctc_alpha_body(const int b, const float* __restrict__ work, int T, int V, const int* __restrict__ enc_lens,
                                               const int* __restrict__ labels, int Umax, const int* __restrict__ label_lens,
                                               float* __restrict__ nll, float* __restrict__ alpha_out, float* __restrict__ nllp_out) {
    __shared__ float alpha[2][CTC_MAXS + 2];
    __shared__ double osum[CTC_NT];
    const int tid = threadIdx.x;
    const int len = min(max(enc_lens[b], 0), T);
    const int U = min(max(label_lens[b], 0), Umax);
    const int S = 2 * U + 1, SM = 2 * Umax + 2;             // row stride of work / alpha (the last slot of a work row is the frame's offset)
    if (len == 0) {                                        // no frames: only the empty label sequence is possible
        if (tid == 0) {
            nll[b] = U == 0 ? 0.f : INFINITY;
            if (nllp_out) nllp_out[b] = nll[b];
        }
        return;
    }
    const float* lp = work + (int64_t)b * T * SM;
    bool skip[2], live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = tid + i * CTC_NT;
        live[i] = s < S;
        skip[i] = false;
        if (live[i] && s >= 2 && (s & 1)) {
            const int z = ext_label(labels, (int64_t)b * Umax, s, V);
            skip[i] = z != 0 && z != ext_label(labels, (int64_t)b * Umax, s - 2, V);
        }
    }
    // alpha[.][0..1] are two -inf guard cells so that state s reads s-1 and s-2 without branches: state s lives at index s + 2
    if (tid < 2) alpha[0][tid] = alpha[1][tid] = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = tid + i * CTC_NT;
        if (live[i]) {
            const float a0 = s < 2 ? lp[s] : -INFINITY;
            alpha[0][s + 2] = a0;
            if (alpha_out) alpha_out[(int64_t)b * T * SM + s] = a0;
        }
    }
    constexpr int AHEAD = 4;                               // lp rows requested this many frames before they are used
    float nxt[AHEAD][2];
#pragma unroll
    for (int k = 0; k < AHEAD; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = 1 + k, s = tid + i * CTC_NT;
            nxt[k][i] = (live[i] && t < len) ? lp[(int64_t)t * SM + s] : 0.f;
        }
    __syncthreads();
    int cur = 0;
    for (int t0 = 1; t0 < len; t0 += AHEAD) {
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const int t = t0 + k;
            if (t < len) {                                 // uniform
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int s = tid + i * CTC_NT;
                    if (live[i]) {
                        float a = logaddexp_(alpha[cur][s + 2], alpha[cur][s + 1]);
                        if (skip[i]) a = logaddexp_(a, alpha[cur][s]);
                        alpha[cur ^ 1][s + 2] = a + nxt[k][i];
                        if (alpha_out) alpha_out[((int64_t)b * T + t) * SM + s] = a + nxt[k][i];
                        const int tn = t + AHEAD;
                        nxt[k][i] = tn < len ? lp[(int64_t)tn * SM + s] : 0.f;
                    }
                }
                __syncthreads();
                cur ^= 1;
            }
        }
    }
    double part = 0.0;                                      // sum_t o_t, fp64
    for (int t = tid; t < len; t += CTC_NT) part += (double)lp[(int64_t)t * SM + SM - 1];
    osum[tid] = part;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int i = 0; i < CTC_NT; ++i) tot += osum[i];
        const float a = S >= 2 ? logaddexp_(alpha[cur][S + 1], alpha[cur][S]) : alpha[cur][S + 1];
        nll[b] = (float)(-((double)a + tot));
        if (nllp_out) nllp_out[b] = -a;                     // the shifted recursion's own -log P': what the posteriors are normalised with
    }
}


// beta recursion, one workgroup per utterance, walking the frames backwards; ab[t][s] = log(alpha_t(s) beta_t(s) / y_t(z_s)) replaces
// alpha in place (both recursions include the emission at t, as torch's CTC does, hence the division)
// SEP: the recursion does not read alpha; it writes bml[t][s] = log(beta_t(s) / y_t(z_s)) to its own array, so that it can run BESIDE the alpha
// recursion (one launch of 2 B workgroups, cfm_ctc_nll_train with a beta buffer) and the gradient kernel forms alpha + bml -- the same f32
// sum the in-place form stores.
template <bool SEP>
__device__ __forceinline__ void ctc_beta_body(const int b, const float* __restrict__ work, float* __restrict__ ab, int T, int V, const int* __restrict__ enc_lens,
                                              const int* __restrict__ labels, int Umax, const int* __restrict__ label_lens) {
    __shared__ float beta[2][CTC_MAXS + 2];
    const int tid = threadIdx.x;
    const int len = min(max(enc_lens[b], 0), T);
    const int U = min(max(label_lens[b], 0), Umax);
    const int S = 2 * U + 1, SM = 2 * Umax + 2;
    if (len == 0) return;
    const float* lp = work + (int64_t)b * T * SM;
    float* abp = ab + (int64_t)b * T * SM;
    bool skip[2], live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = tid + i * CTC_NT;
        live[i] = s < S;
        skip[i] = false;
        if (live[i] && s + 2 < S && (s & 1)) {              // s -> s+2 skips a blank: allowed when the two labels differ
            const int z2 = ext_label(labels, (int64_t)b * Umax, s + 2, V);
            skip[i] = z2 != 0 && z2 != ext_label(labels, (int64_t)b * Umax, s, V);
        }
    }
    // state s lives at index s; indices S and S+1 are -inf guard cells so that s+1 / s+2 need no branch
    for (int i = tid; i < CTC_MAXS + 2; i += CTC_NT) beta[0][i] = beta[1][i] = -INFINITY;
    __syncthreads();
    int cur = 0;
    {
        const int t = len - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = tid + i * CTC_NT;
            if (live[i]) {
                const float l = lp[(int64_t)t * SM + s];
                const float bt = (s >= S - 2) ? l : -INFINITY;
                beta[0][s] = bt;
                if constexpr (SEP) abp[(int64_t)t * SM + s] = bt - l;
                else abp[(int64_t)t * SM + s] = abp[(int64_t)t * SM + s] + (bt - l);      // grouped as the SEP form's sum
            }
        }
        __syncthreads();
    }
    // the frame's log-probabilities and alpha values are requested AHEAD frames before they are used (as the alpha recursion does): read in
    // the step that needs them, every step waited out two L2 round trips (180 us per launch at config 3 against 114 us for alpha)
    constexpr int AHEAD = 4;
    float nl[AHEAD][2], na[AHEAD][2];
#pragma unroll
    for (int k = 0; k < AHEAD; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = len - 2 - k, s = tid + i * CTC_NT;
            const bool ok = live[i] && t >= 0;
            nl[k][i] = ok ? lp[(int64_t)t * SM + s] : 0.f;
            na[k][i] = (ok && !SEP) ? abp[(int64_t)t * SM + s] : 0.f;
        }
    for (int t0 = len - 2; t0 >= 0; t0 -= AHEAD) {
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const int t = t0 - k;
            if (t >= 0) {                                  // uniform
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int s = tid + i * CTC_NT;
                    if (live[i]) {
                        float a = logaddexp_(beta[cur][s], beta[cur][s + 1]);
                        if (skip[i]) a = logaddexp_(a, beta[cur][s + 2]);
                        beta[cur ^ 1][s] = a + nl[k][i];
                        abp[(int64_t)t * SM + s] = SEP ? a : na[k][i] + a;        // (alpha +) beta - lp
                        const int tn = t - AHEAD;
                        nl[k][i] = tn >= 0 ? lp[(int64_t)tn * SM + s] : 0.f;
                        if constexpr (!SEP) na[k][i] = tn >= 0 ? abp[(int64_t)tn * SM + s] : 0.f;
                    }
                }
                __syncthreads();
                cur ^= 1;
            }
        }
    }
}

__global__ __launch_bounds__(CTC_NT) void cfm_ctc_alpha_kernel(const float* __restrict__ work, int T, int V, const int* __restrict__ enc_lens,
                                                               const int* __restrict__ labels, int Umax, const int* __restrict__ label_lens,
                                                               float* __restrict__ nll, float* __restrict__ alpha_out, float* __restrict__ nllp_out) {
    ctc_alpha_body(blockIdx.x, work, T, V, enc_lens, labels, Umax, label_lens, nll, alpha_out, nllp_out);
}

__global__ __launch_bounds__(CTC_NT) void cfm_ctc_beta_kernel(const float* __restrict__ work, float* __restrict__ ab, int T, int V, const int* __restrict__ enc_lens,
                                                              const int* __restrict__ labels, int Umax, const int* __restrict__ label_lens) {
    ctc_beta_body<false>(blockIdx.x, work, ab, T, V, enc_lens, labels, Umax, label_lens);
}

// both recursions in ONE launch: workgroups [0, B) walk forwards, [B, 2B) backwards (each is a serial chain of T' steps on one CU; a training
// micro-batch has 5..40 utterances, so the two launches ran one after the other on a nearly empty chip: 113 + 113 us at config 3)
__global__ __launch_bounds__(CTC_NT) void cfm_ctc_alpha_beta_kernel(const float* __restrict__ work, int B, int T, int V, const int* __restrict__ enc_lens,
                                                                    const int* __restrict__ labels, int Umax, const int* __restrict__ label_lens,
                                                                    float* __restrict__ nll, float* __restrict__ alpha_out, float* __restrict__ nllp_out,
                                                                    float* __restrict__ bml) {
    if ((int)blockIdx.x < B) ctc_alpha_body(blockIdx.x, work, T, V, enc_lens, labels, Umax, label_lens, nll, alpha_out, nllp_out);
    else ctc_beta_body<true>((int)blockIdx.x - B, work, bml, T, V, enc_lens, labels, Umax, label_lens);
}

// the same for the micro-batches of a training window (cfm_ctc_nll_train_groups): each recursion is a serial chain of T' steps on one CU and a
// micro-batch has 5..40 utterances, so two micro-batches' launches ran one after the other on a nearly empty chip (2 x 113 us at config 3)
constexpr int CTC_GROUPS_MAX = 8;
struct CtcGroupArgs {
    const float* work[CTC_GROUPS_MAX];
    const int* enc_lens[CTC_GROUPS_MAX];
    const int* labels[CTC_GROUPS_MAX];
    const int* label_lens[CTC_GROUPS_MAX];
    float* nll[CTC_GROUPS_MAX];
    float* alpha[CTC_GROUPS_MAX];
    float* nllp[CTC_GROUPS_MAX];
    float* beta[CTC_GROUPS_MAX];
    int B[CTC_GROUPS_MAX], T[CTC_GROUPS_MAX], Umax[CTC_GROUPS_MAX];
    int first[CTC_GROUPS_MAX + 1];      // first workgroup of each micro-batch (2 * B workgroups each)
    int n, V;
};

__global__ __launch_bounds__(CTC_NT) void cfm_ctc_alpha_beta_group_kernel(const CtcGroupArgs G) {
    const int wg = (int)blockIdx.x;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < CTC_GROUPS_MAX; ++i)
        if (i < G.n && wg >= G.first[i]) gi = i;            // uniform
    const int rel = wg - G.first[gi], B = G.B[gi];
    if (rel < B) ctc_alpha_body(rel, G.work[gi], G.T[gi], G.V, G.enc_lens[gi], G.labels[gi], G.Umax[gi], G.label_lens[gi], G.nll[gi], G.alpha[gi], G.nllp[gi]);
    else ctc_beta_body<true>(rel - B, G.work[gi], G.beta[gi], G.T[gi], G.V, G.enc_lens[gi], G.labels[gi], G.Umax[gi], G.label_lens[gi]);
}

// d nll / d logits, one wavefront per frame, persistent over frames.  Each wavefront keeps an occupancy table over the vocabulary in LDS:
// the <= 2U+1 state posteriors exp(ab + nll) of a frame are scattered into it (several states share a class: blank, repeated labels),
// the row is written as gs * (softmax - occupancy), and the touched entries are cleared again.
__global__ __launch_bounds__(CTC_NT) void cfm_ctc_grad_kernel(const float* __restrict__ logits, int64_t ld, int B, int T, int V, const int* __restrict__ enc_lens,
                                                              const int* __restrict__ labels, int Umax, const int* __restrict__ label_lens,
                                                              const float* __restrict__ ab, const float* __restrict__ bml, const float* __restrict__ lse,
                                                              const float* __restrict__ nll, float gscale, const float* __restrict__ gscale_dev, float* __restrict__ dlogits) {
    extern __shared__ float occ_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Vp = (V + 3) & ~3;
    float* occ = occ_all + wave * Vp;
    for (int c = lane; c < Vp; c += 64) occ[c] = 0.f;
    __threadfence_block();
    const float gs = gscale * (gscale_dev ? *gscale_dev : 1.f);
    const int SM = 2 * Umax + 2;
    const int64_t rows = (int64_t)B * T;
    for (int64_t row_id = (int64_t)blockIdx.x * (CTC_NT / 64) + wave; row_id < rows; row_id += (int64_t)gridDim.x * (CTC_NT / 64)) {
        const int b = (int)(row_id / T), t = (int)(row_id % T);
        const float* row = logits + row_id * ld;
        float* drow = dlogits + row_id * ld;
        const float nl = nll[b];
        const bool live = t < min(max(enc_lens[b], 0), T) && nl < INFINITY && nl > -INFINITY;
        if (!live) {                                        // padded frame (or an impossible alignment: torch's gradient is undefined there)
            for (int c = lane * 4; c < (int)ld; c += 256) {
                if (c + 3 < (int)ld) *(f32x4*)(drow + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
                else for (int k = c; k < (int)ld; ++k) drow[k] = 0.f;
            }
            continue;
        }
        const int S = 2 * min(max(label_lens[b], 0), Umax) + 1;
        for (int s = lane; s < S; s += 64) {
            const float abv = bml ? ab[row_id * SM + s] + bml[row_id * SM + s] : ab[row_id * SM + s];      // alpha + (beta - lp): the in-place form's own sum
            atomicAdd(&occ[ext_label(labels, (int64_t)b * Umax, s, V)], __expf(abv + nl));
        }
        __threadfence_block();
        const float l = lse[row_id];
        for (int c = lane * 4; c < (int)ld; c += 256) {
            if (c + 3 < V) {
                const f32x4 v = *(const f32x4*)(row + c), o = *(const f32x4*)(occ + c);
                f32x4 gq;
#pragma unroll
                for (int e = 0; e < 4; ++e) gq[e] = gs * (__expf(v[e] - l) - o[e]);
                *(f32x4*)(drow + c) = gq;
            } else {
                for (int k = c; k < (int)ld && k < c + 4; ++k) drow[k] = k < V ? gs * (__expf(row[k] - l) - occ[k]) : 0.f;
            }
        }
        __threadfence_block();
        for (int s = lane; s < S; s += 64) occ[ext_label(labels, (int64_t)b * Umax, s, V)] = 0.f;
        __threadfence_block();
    }
}

}  // namespace

static int ctc_forward(const float* logits, int64_t ld, int32_t B, int32_t T, int32_t V, const int32_t* enc_lens, const int32_t* labels, int32_t Umax,
                       const int32_t* label_lens, float* work, float* alpha, float* lse, float* nll, float* nllp, float* beta, cfm_stream_t stream,
                       bool rows_only = false) {
    CFM_CHECK_ARG(logits && enc_lens && labels && label_lens && work && nll, "cfm_ctc_nll: null pointer");
    CFM_CHECK_ARG(B > 0 && T > 0 && V > 1 && Umax > 0, "cfm_ctc_nll: bad shape B=%d T=%d V=%d Umax=%d", B, T, V, Umax);
    CFM_CHECK_ARG(2 * Umax + 1 <= CTC_MAXS, "cfm_ctc_nll: Umax=%d labels exceeds %d", Umax, (CTC_MAXS - 1) / 2);
    CFM_CHECK_ARG(ld >= V && ld % 4 == 0, "cfm_ctc_nll: row stride %lld must be >= V and a multiple of 4", (long long)ld);
    hipStream_t s = (hipStream_t)stream;
    {
        const int64_t rows = (int64_t)B * T;
        CfmProfScope prof("ctc_rows", s, 0.0, (double)rows * V * 4);
        CFM_LAUNCH(cfm_ctc_rows_kernel, dim3((unsigned)((rows + CTC_NT / 64 - 1) / (CTC_NT / 64))), dim3(CTC_NT), 0, s, logits, ld, B, T, V, enc_lens,
                   labels, Umax, label_lens, work, lse);
        if (int rc = cfm_launch_status("cfm_ctc_nll (rows)")) return rc;
    }
    if (rows_only) return CFM_OK;
    if (beta) {
        CfmProfScope prof("ctc_alpha_beta", s, 0.0, (double)B * T * (2 * Umax + 1) * 16);
        CFM_LAUNCH(cfm_ctc_alpha_beta_kernel, dim3(2 * B), dim3(CTC_NT), 0, s, (const float*)work, B, T, V, enc_lens, labels, Umax, label_lens, nll, alpha, nllp, beta);
        return cfm_launch_status("cfm_ctc_nll (alpha | beta)");
    }
    CfmProfScope prof("ctc_alpha", s, 0.0, (double)B * T * (2 * Umax + 1) * 4);
    CFM_LAUNCH(cfm_ctc_alpha_kernel, dim3(B), dim3(CTC_NT), 0, s, (const float*)work, T, V, enc_lens, labels, Umax, label_lens, nll, alpha, nllp);
    return cfm_launch_status("cfm_ctc_nll (alpha)");
}

extern "C" int cfm_ctc_nll(const float* logits, int64_t ld, int32_t B, int32_t T, int32_t V, const int32_t* enc_lens,
                           const int32_t* labels, int32_t Umax, const int32_t* label_lens, float* work, float* nll, cfm_stream_t stream) {
    return ctc_forward(logits, ld, B, T, V, enc_lens, labels, Umax, label_lens, work, nullptr, nullptr, nll, nullptr, nullptr, stream);
}

extern "C" int cfm_ctc_nll_train(const float* logits, int64_t ld, int32_t B, int32_t T, int32_t V, const int32_t* enc_lens, const int32_t* labels, int32_t Umax,
                                 const int32_t* label_lens, float* work, float* alpha, float* lse, float* nll, float* nll_shifted, float* beta, cfm_stream_t stream) {
    CFM_CHECK_ARG(alpha && lse && nll_shifted, "cfm_ctc_nll_train: null pointer");
    return ctc_forward(logits, ld, B, T, V, enc_lens, labels, Umax, label_lens, work, alpha, lse, nll, nll_shifted, beta, stream);
}

extern "C" int cfm_ctc_nll_train_groups(const cfm_ctc_group* groups, int32_t n, int32_t V, cfm_stream_t stream) {
    CFM_CHECK_ARG(groups && n > 0 && n <= CTC_GROUPS_MAX, "cfm_ctc_nll_train_groups: 1 .. %d micro-batches", CTC_GROUPS_MAX);
    CtcGroupArgs G;
    G.n = n; G.V = V;
    int first = 0;
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const cfm_ctc_group& g = groups[i];
        CFM_CHECK_ARG(g.alpha && g.lse && g.nll_shifted && g.beta, "cfm_ctc_nll_train_groups: null pointer");
        if (int rc = ctc_forward(g.logits, g.ld, g.B, g.T, V, g.enc_lens, g.labels, g.Umax, g.label_lens, g.work, g.alpha, g.lse, g.nll, g.nll_shifted, g.beta, stream, true))
            return rc;
        G.work[i] = g.work; G.enc_lens[i] = g.enc_lens; G.labels[i] = g.labels; G.label_lens[i] = g.label_lens; G.nll[i] = g.nll; G.alpha[i] = g.alpha;
        G.nllp[i] = g.nll_shifted; G.beta[i] = g.beta; G.B[i] = g.B; G.T[i] = g.T; G.Umax[i] = g.Umax; G.first[i] = first;
        first += 2 * g.B;
        bytes += (double)g.B * g.T * (2 * g.Umax + 1) * 16;
    }
    for (int i = n; i < CTC_GROUPS_MAX; ++i) {
        G.work[i] = G.work[0]; G.enc_lens[i] = G.enc_lens[0]; G.labels[i] = G.labels[0]; G.label_lens[i] = G.label_lens[0]; G.nll[i] = G.nll[0]; G.alpha[i] = G.alpha[0];
        G.nllp[i] = G.nllp[0]; G.beta[i] = G.beta[0]; G.B[i] = G.B[0]; G.T[i] = G.T[0]; G.Umax[i] = G.Umax[0];
    }
    for (int i = n; i <= CTC_GROUPS_MAX; ++i) G.first[i] = first;
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("ctc_alpha_beta_group", s, 0.0, bytes);
    CFM_LAUNCH(cfm_ctc_alpha_beta_group_kernel, dim3((unsigned)first), dim3(CTC_NT), 0, s, G);
    return cfm_launch_status("cfm_ctc_nll_train_groups (alpha | beta)");
}

extern "C" int cfm_ctc_grad(const float* logits, int64_t ld, int32_t B, int32_t T, int32_t V, const int32_t* enc_lens, const int32_t* labels, int32_t Umax,
                            const int32_t* label_lens, const float* work, float* alpha_beta, const float* beta, const float* lse, const float* nll_shifted,
                            float gscale, const float* gscale_dev, float* dlogits, cfm_stream_t stream) {
    const float* nll = nll_shifted;
    CFM_CHECK_ARG(logits && enc_lens && labels && label_lens && work && alpha_beta && lse && nll && dlogits, "cfm_ctc_grad: null pointer");
    CFM_CHECK_ARG(B > 0 && T > 0 && V > 1 && Umax > 0 && 2 * Umax + 1 <= CTC_MAXS, "cfm_ctc_grad: bad shape B=%d T=%d V=%d Umax=%d", B, T, V, Umax);
    CFM_CHECK_ARG(ld >= V && ld % 4 == 0, "cfm_ctc_grad: row stride %lld must be >= V and a multiple of 4", (long long)ld);
    const int Vp = (V + 3) & ~3;
    const size_t lds = (size_t)(CTC_NT / 64) * Vp * 4;
    CFM_CHECK_ARG(lds <= 128 * 1024, "cfm_ctc_grad: vocabulary of %d classes exceeds the LDS occupancy tables (<= 8192)", V);
    hipStream_t s = (hipStream_t)stream;
    if (!beta) {                                            // the backward recursion has not run beside the forward one: here, in place over alpha
        CfmProfScope prof("ctc_beta", s, 0.0, (double)B * T * (2 * Umax + 1) * 12);
        CFM_LAUNCH(cfm_ctc_beta_kernel, dim3(B), dim3(CTC_NT), 0, s, work, alpha_beta, T, V, enc_lens, labels, Umax, label_lens);
        if (int rc = cfm_launch_status("cfm_ctc_grad (beta)")) return rc;
    }
    static bool attr_set = false;                           // > 64 KB of dynamic LDS needs the attribute once per process
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)cfm_ctc_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return cfm_fail(CFM_ERR_LAUNCH, "cfm_ctc_grad: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const int64_t rows = (int64_t)B * T;
    int64_t nb = (rows + CTC_NT / 64 - 1) / (CTC_NT / 64);
    nb = nb > 512 ? 512 : nb;
    CfmProfScope prof("ctc_grad", s, 0.0, (double)rows * ld * 8);
    CFM_LAUNCH(cfm_ctc_grad_kernel, dim3((unsigned)nb), dim3(CTC_NT), lds, s, logits, ld, B, T, V, enc_lens, labels, Umax, label_lens, (const float*)alpha_beta, beta, lse, nll,
               gscale, gscale_dev, dlogits);
    return cfm_launch_status("cfm_ctc_grad");
}
