// greedy.hip -- one step of the batched RNN-T greedy search (reference src/model.py:215-269, greedy.py) as SIX launches on f32 weights:
//
//   1, 2  LSTM layers   gates = [x | h] . [W_ih | W_hh]^T + (b_ih + b_hh);  c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c')
//                       (layer 0's x is the embedding row of the stream's current token: a gather, no launch of its own)
//   3     projection    pred = h' . Wp^T + bp                                     (predictor.py:83)
//   4     joint input   a = tanh(enc_proj[b, t_b] + pred . Wpf^T + bpf)           (joint.py:34-36; enc_ffn was applied to all frames once)
//   5     joint output  z = a . Wout^T + bout, per 16-class tile and stream the (max, first index) pair                (joint.py:37, model.py:254)
//   6     control       k = argmax over the tiles; the reference's branches as selects on the per-stream state (model.py:255-267)
//
// The torch-operation form of the same step (greedy.py) is ~45 small launches; the products here are "skinny": B <= 64 streams against weight
// matrices of 0.26-2.5 M elements, i.e. one read of 7.5 M f32 weights per step, spread over the chip.  Everything stays f32 so that the argmax is
// the reference's: the products run on the f32 MFMA (v_mfma_f32_16x16x4_f32: exact f32 multiplies, f32 accumulation) -- one wavefront
// per 16 output columns, the weight rows as the A operand and up to four 16-stream tiles as B operands, both read as 16-byte pieces straight
// from memory (the contraction index is walked in the order lane group g holds k = 16 q + 4 g + r, the same for both operands).
//   * the LSTM weight rows are packed [unit][gate] so that the lane that owns output rows 4 g .. 4 g + 3 of a tile holds the four gates
//     (i, f, g, o) of one hidden unit of one stream: the cell update is lane-local;
//   * streams that are finished, or whose step produced a blank, keep their state: the candidates (h', c') go to side buffers and the
//     control launch selects.
#include <math.h>

#include "cfm_common.h"

namespace {

struct SkinnyArgs {
    const float* W;          // [N, K] f32 row-major (N % 16 == 0, K % 16 == 0)
    const float* bias;       // [N]
    const float* x1;         // first K1 contraction columns: rows at x1 + row(b) * ld1 ...
    const int64_t* x1_rows;  // ... row(b) = x1_rows[b] when given (the embedding gather by token), else b
    int64_t ld1;
    int K1;
    const float* x2;         // remaining K - K1 columns: x2 + b * ld2 (LSTM: the layer's hidden state)
    int64_t ld2;
    int B, N, K;
    // epilogues
    float* out;              // EPI 0: out[b, n] (ld_out)
    int64_t ld_out;
    const float* c_in;       // EPI 1 (LSTM cell, N = 4 H): cell state [B, H]
    float *h_out, *c_out;    //        candidate h', c' [B, H]
    const float* enc;        // EPI 2: enc_proj [B, T, N];  a[b, n] = tanh(enc[b, min(t[b], T-1), n] + acc + bias)
    const int64_t* t_idx;
    int T;
    float* pmax;             // EPI 3: per (tile, b): best value / first index of the tile's 16 classes
    int* pidx;
};

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int EPI>
__global__ __launch_bounds__(256) void cfm_skinny_kernel(const SkinnyArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x * 4 + wave;                 // 16 output columns per wavefront
    if (tile * 16 >= a.N) return;
    const int n0 = tile * 16;
    const int nbt = (a.B + 15) / 16;                        // <= 4 stream tiles
    f32x4 acc[4];
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wrow = a.W + (int64_t)(n0 + l15) * a.K + 4 * g;
    const float* xr1[4];
    const float* xr2[4];
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
        int b = bt * 16 + l15;
        b = b < a.B ? b : a.B - 1;
        const int64_t r1 = a.x1_rows ? a.x1_rows[b] : (int64_t)b;
        xr1[bt] = a.x1 + r1 * a.ld1 + 4 * g;
        xr2[bt] = a.x2 ? a.x2 + (int64_t)b * a.ld2 + 4 * g : xr1[bt];
    }
    const int nq = a.K / 16, nq1 = a.K1 / 16;
    for (int q = 0; q < nq; ++q) {                          // (unrolling by 4 to batch the requests: no gain, 63 vs 58 us per step)
        const f32x4 wv = *(const f32x4*)(wrow + 16 * q);
        const bool first = q < nq1;                         // uniform
        f32x4 xv[4];
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
            if (bt < nbt) xv[bt] = first ? *(const f32x4*)(xr1[bt] + 16 * q) : *(const f32x4*)(xr2[bt] + 16 * (q - nq1));
        }
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
            if (bt < nbt) {                                  // uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r], xv[bt][r], acc[bt], 0, 0, 0);
            }
        }
    }
    // lane (l15, g) holds output rows n0 + 4 g + r (r = 0..3) of stream bt * 16 + l15
    const f32x4 bv = *(const f32x4*)(a.bias + n0 + 4 * g);
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
        const int b = bt * 16 + l15;
        if (bt >= nbt) break;
        const f32x4 v = acc[bt] + bv;
        const bool live = b < a.B;
        if constexpr (EPI == 0) {
            if (live) *(f32x4*)(a.out + (int64_t)b * a.ld_out + n0 + 4 * g) = v;
        } else if constexpr (EPI == 1) {                    // rows are [unit][i, f, g, o]: this lane owns unit (n0 >> 2) + g
            if (live) {
                const int H = a.N >> 2, u = (n0 >> 2) + g;
                const float c0 = a.c_in[(int64_t)b * H + u];
                const float c1 = sigmoid_acc(v.y) * c0 + sigmoid_acc(v.x) * tanhf(v.z);
                a.c_out[(int64_t)b * H + u] = c1;
                a.h_out[(int64_t)b * H + u] = sigmoid_acc(v.w) * tanhf(c1);
            }
        } else if constexpr (EPI == 2) {
            if (live) {
                int64_t t = a.t_idx[b];
                t = t < a.T ? t : a.T - 1;
                const f32x4 e = *(const f32x4*)(a.enc + ((int64_t)b * a.T + t) * a.N + n0 + 4 * g);
                *(f32x4*)(a.out + (int64_t)b * a.ld_out + n0 + 4 * g) = (f32x4){tanhf(e.x + v.x), tanhf(e.y + v.y), tanhf(e.z + v.z), tanhf(e.w + v.w)};
            }
        } else {                                            // EPI 3: the tile's best class per stream, lowest index on ties
            float best = v.x;
            int bi = n0 + 4 * g;
            if (v.y > best) { best = v.y; bi = n0 + 4 * g + 1; }
            if (v.z > best) { best = v.z; bi = n0 + 4 * g + 2; }
            if (v.w > best) { best = v.w; bi = n0 + 4 * g + 3; }
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (g == 0 && live) {
                a.pmax[(int64_t)tile * a.B + b] = best;
                a.pidx[(int64_t)tile * a.B + b] = bi;
            }
        }
    }
}

struct CtlArgs {
    const float* pmax;
    const int* pidx;
    int ntiles, B, L, H, blank, n_steps;
    int64_t *token, *t, *count, *frame_count, *hyps;
    const int64_t* lens;
    int64_t hyp_cap, hyp_ld;
    float *h, *c;                // [L, B, H] state
    const float *h_new, *c_new;  // [L, B, H] candidates of this step
    uint8_t* done;
    int* n_done;                 // number of finished streams after this step (one int)
};

// one workgroup per stream: argmax over the tiles, then model.py:255-267 as selects
__global__ __launch_bounds__(256) void cfm_greedy_control_kernel(const CtlArgs a) {
    __shared__ float smax[256];
    __shared__ int sidx[256];
    __shared__ int s_nb;
    const int b = blockIdx.x, tid = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int tl = tid; tl < a.ntiles; tl += 256) {
        const float v = a.pmax[(int64_t)tl * a.B + b];
        const int i = a.pidx[(int64_t)tl * a.B + b];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    smax[tid] = best;
    sidx[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float v = smax[tid + s];
            const int i = sidx[tid + s];
            if (v > smax[tid] || (v == smax[tid] && i < sidx[tid])) { smax[tid] = v; sidx[tid] = i; }
        }
        __syncthreads();
    }
    const bool was_done = a.done[b] != 0;
    if (tid == 0) {
        const int k = sidx[0];
        const bool nb = k != a.blank && !was_done;
        s_nb = nb ? 1 : 0;
        int64_t fc = a.frame_count[b], t = a.t[b];
        if (nb) {
            const int64_t pos = a.count[b] < a.hyp_cap ? a.count[b] : a.hyp_cap;
            a.hyps[(int64_t)b * a.hyp_ld + pos] = k;
            a.count[b] += 1;
            a.token[b] = k;
            fc += 1;
        }
        const bool adv = (k == a.blank || fc >= a.n_steps) && !was_done;
        if (adv) { t += 1; fc = 0; }
        a.t[b] = t;
        a.frame_count[b] = fc;
        const bool dn = t >= a.lens[b];
        a.done[b] = dn ? 1 : 0;
        if (dn && !was_done) atomicAdd(a.n_done, 1);
    }
    __syncthreads();
    if (s_nb) {                                             // a non-blank: the LSTM's new state becomes the stream's state
        for (int i = tid; i < a.L * a.H; i += 256) {
            const int l = i / a.H, u = i - l * a.H;
            const int64_t o = ((int64_t)l * a.B + b) * a.H + u;
            a.h[o] = a.h_new[o];
            a.c[o] = a.c_new[o];
        }
    }
}

template <int EPI>
int launch_skinny(const SkinnyArgs& a, hipStream_t s, const char* name) {
    const int tiles = a.N / 16;
    CfmProfScope prof(name, s, 2.0 * a.B * (double)a.N * a.K, (double)a.N * a.K * 4);
    CFM_LAUNCH((cfm_skinny_kernel<EPI>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, a);
    return cfm_launch_status(name);
}

}  // namespace

extern "C" int cfm_greedy_step(const cfm_greedy_desc* d, cfm_stream_t stream) {
    CFM_CHECK_ARG(d, "cfm_greedy_step: null descriptor");
    CFM_CHECK_ARG(d->B > 0 && d->B <= 64 && d->L >= 1 && d->L <= 4 && d->E % 16 == 0 && d->H % 16 == 0 && d->P % 16 == 0 && d->J % 16 == 0 && d->Vp % 16 == 0 &&
                      d->T > 0 && d->n_steps > 0,
                  "cfm_greedy_step: B <= 64 streams, <= 4 LSTM layers, sizes multiples of 16 (B=%d L=%d E=%d H=%d P=%d J=%d Vp=%d)", d->B, d->L, d->E, d->H, d->P,
                  d->J, d->Vp);
    CFM_CHECK_ARG(d->embed && d->proj_w && d->proj_b && d->pf_w && d->pf_b && d->out_w && d->out_b && d->enc_proj && d->token && d->t && d->lens && d->count &&
                      d->frame_count && d->hyps && d->h && d->c && d->h_new && d->c_new && d->pred && d->act && d->pmax && d->pidx && d->done && d->n_done,
                  "cfm_greedy_step: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int B = d->B, H = d->H;
    for (int l = 0; l < d->L; ++l) {
        CFM_CHECK_ARG(d->lstm_w[l] && d->lstm_b[l], "cfm_greedy_step: LSTM layer %d has no weights", l);
        SkinnyArgs a = {};
        a.W = d->lstm_w[l]; a.bias = d->lstm_b[l]; a.B = B; a.N = 4 * H; a.x2 = d->h + (int64_t)l * B * H; a.ld2 = H;
        if (l == 0) { a.x1 = d->embed; a.x1_rows = d->token; a.ld1 = d->E; a.K1 = d->E; }
        else { a.x1 = d->h_new + (int64_t)(l - 1) * B * H; a.ld1 = H; a.K1 = H; }
        a.K = a.K1 + H;
        a.c_in = d->c + (int64_t)l * B * H; a.h_out = d->h_new + (int64_t)l * B * H; a.c_out = d->c_new + (int64_t)l * B * H;
        if (int rc = launch_skinny<1>(a, s, "greedy_lstm")) return rc;
    }
    {
        SkinnyArgs a = {};
        a.W = d->proj_w; a.bias = d->proj_b; a.B = B; a.N = d->P; a.K = a.K1 = H; a.x1 = d->h_new + (int64_t)(d->L - 1) * B * H; a.ld1 = H; a.out = d->pred; a.ld_out = d->P;
        if (int rc = launch_skinny<0>(a, s, "greedy_proj")) return rc;
    }
    {
        SkinnyArgs a = {};
        a.W = d->pf_w; a.bias = d->pf_b; a.B = B; a.N = d->J; a.K = a.K1 = d->P; a.x1 = d->pred; a.ld1 = d->P; a.out = d->act; a.ld_out = d->J; a.enc = d->enc_proj;
        a.t_idx = d->t; a.T = d->T;
        if (int rc = launch_skinny<2>(a, s, "greedy_joint_in")) return rc;
    }
    {
        SkinnyArgs a = {};
        a.W = d->out_w; a.bias = d->out_b; a.B = B; a.N = d->Vp; a.K = a.K1 = d->J; a.x1 = d->act; a.ld1 = d->J; a.pmax = d->pmax; a.pidx = d->pidx;
        if (int rc = launch_skinny<3>(a, s, "greedy_joint_out")) return rc;
    }
    CtlArgs c;
    c.pmax = d->pmax; c.pidx = d->pidx; c.ntiles = d->Vp / 16; c.B = B; c.L = d->L; c.H = H; c.blank = d->blank; c.n_steps = d->n_steps;
    c.token = d->token; c.t = d->t; c.count = d->count; c.frame_count = d->frame_count; c.hyps = d->hyps; c.lens = d->lens; c.hyp_cap = d->hyp_cap;
    c.hyp_ld = d->hyp_ld; c.h = d->h; c.c = d->c; c.h_new = d->h_new; c.c_new = d->c_new; c.done = d->done; c.n_done = d->n_done;
    CfmProfScope prof("greedy_control", s, 0.0, (double)B * (d->Vp / 16) * 8);
    CFM_LAUNCH(cfm_greedy_control_kernel, dim3((unsigned)B), dim3(256), 0, s, c);
    return cfm_launch_status("cfm_greedy_step (control)");
}
