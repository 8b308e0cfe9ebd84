// train_layer.cpp -- one conformer block in TRAIN mode, forward and backward, enqueued from C++ (no host synchronisation, no allocation).
//
// reference: src/encoder_layer.py:49-71 under module.train() -- four residual sub-blocks (1/2 FFN, MHSA, convolution module with BatchNorm
// batch statistics, 1/2 FFN) + norm_final, dropout on every branch output / FFN hidden activation / attention probabilities -- and what
// autograd derives from it.  The same launches, in the same order, as the op-by-op composition in cfm/autograd.py (which remains for the
// bare modules and as the readable specification; tests run both and compare): ~17 launches forward, ~60 backward per block, issued from
// ONE host call each instead of ~60 Python -> ctypes round trips (~18 us apiece: the step was host-bound at 40 ms with 27 ms of kernels).
//
// Parameter gradients are written (accumulated: the caller zero-fills) straight into a caller-provided flat buffer through per-parameter
// pointers / row-offset maps -- in the data-parallel trainer that buffer IS the gradient bucket memory the RCCL all-reduce runs on.
//
// Round 3 -- what the training step's time was (13.4 ms, 1 460 launches of ~9 us, host and device both saturated) decided the shape of this file:
//   * ROW GROUPS.  Everything in a block except attention, the depthwise convolution and the BatchNorm statistics is row-local, so the
//     micro-batches of one accumulation window (train.sh:36 accum_grad 2; the weights do not change between them) are concatenated along
//     the row axis: io->groups describes G micro-batches of B_g x T_g frames at row offset row0_g of ONE [M, D] row matrix.  The dense
//     products, LayerNorms and their backward run ONCE over all M rows (half the launches, twice the rows per launch: these kernels are
//     latency-bound at 2 400 rows), the weight gradient of the window is one product over all rows (what accumulating two micro-batch
//     gradients computes), attention / depthwise / BatchNorm run per group -- BatchNorm statistics per micro-batch and the running
//     statistics updated group after group, exactly the reference's sequence of two forward passes.
//   * DEFERRED, GROUPED WEIGHT GRADIENTS.  The eight dW products of a block do not feed the chain of input gradients; with io->defer_wgrad
//     their operands are kept (one buffer per sub-block) and the block's backward ends with ONE cfm_gemm_tn_group launch.
//   * THE WHOLE STACK FROM ONE HOST CALL.  cfm_encoder_train_forward / _backward walk all blocks; the backward calls a host callback after
//     each block's launches are enqueued (the data-parallel trainer launches that block's gradient bucket all-reduce from it).
#include <math.h>

#include "cfm_common.h"

namespace {

struct Grp {
    int B, T;
    int64_t row0, bht0;          // first row of the group; first element of its [B,H,T] arrays (lse, delta)
    const uint8_t* mask;
    int64_t sb, sq;
};

struct TCtx {
    const cfm_layer_train_weights* w;
    const cfm_layer_train_io* io;
    int M, D, FF, H, dk, adt, wdt;
    bool split;
    cfm_stream_t st;
    cfm_stream_t side;      // weight-gradient products go here when set (backward)
    int ng;
    Grp grp[CFM_TRAIN_MAX_GROUPS];
    cfm_train_group cg[CFM_TRAIN_MAX_GROUPS];   // the same table in the C-ABI form (the depthwise / BatchNorm entry points take it)
    bool defer;             // weight-gradient products are collected in `pend` and launched as one group at the end of the block's backward
    cfm_stream_t wg_stream; // ... on this stream when set (the caller owns the hand-off back: cfm_encoder_train_backward)
    cfm_gemm_tn_desc pend[12];
    int npend;
};

// io -> the context's dimensions and row groups (n_groups == 0: the single micro-batch B x T of the round-2 interface)
int init_ctx(TCtx& c, const cfm_layer_train_weights* w, const cfm_layer_train_io* io, cfm_stream_t stream) {
    c.w = w; c.io = io; c.D = io->D; c.FF = io->FF; c.H = io->H; c.dk = io->D / io->H; c.adt = io->act_dtype; c.wdt = io->w_dtype;
    c.split = io->act_dtype == CFM_F32; c.st = stream; c.side = nullptr; c.defer = false; c.npend = 0; c.wg_stream = nullptr;
    if (io->n_groups > 0) {
        CFM_CHECK_ARG(io->groups && io->n_groups <= CFM_TRAIN_MAX_GROUPS, "train layer: %d row groups (at most %d)", io->n_groups, CFM_TRAIN_MAX_GROUPS);
        c.ng = io->n_groups;
        int64_t rows = 0, bht = 0;
        for (int i = 0; i < c.ng; ++i) {
            const cfm_train_group& g = io->groups[i];
            CFM_CHECK_ARG(g.B > 0 && g.T > 0 && g.row0 == rows, "train layer: group %d (B=%d T=%d row0=%lld) must start where group %d ends (%lld)", i, g.B, g.T,
                          (long long)g.row0, i - 1, (long long)rows);
            c.grp[i] = {g.B, g.T, g.row0, bht, g.attn_mask, g.am_sb, g.am_sq};
            c.cg[i] = g;
            rows += (int64_t)g.B * g.T;
            bht += (int64_t)g.B * io->H * g.T;
        }
        CFM_CHECK_ARG(rows < (1ll << 31), "train layer: %lld rows", (long long)rows);
        c.M = (int)rows;
    } else {
        c.ng = 1;
        c.grp[0] = {io->B, io->T, 0, 0, io->attn_mask, io->am_sb, io->am_sq};
        c.cg[0] = {};
        c.cg[0].B = io->B; c.cg[0].T = io->T; c.cg[0].row0 = 0;
        c.M = io->B * io->T;
    }
    return CFM_OK;
}

// A small pool of timing-free events for the main -> side stream hand-offs (created once per process; events are recorded and waited on
// in stream order, so one event can be reused as soon as its wait has been enqueued).
struct EventPool {
    hipEvent_t ev[32];
    int n = 0, next = 0;
    hipEvent_t get() {
        if (n < 32) {
            if (hipEventCreateWithFlags(&ev[n], hipEventDisableTiming) != hipSuccess) return nullptr;
            return ev[n++];
        }
        hipEvent_t e = ev[next];
        next = (next + 1) % 32;
        return e;
    }
};
thread_local EventPool g_events;

// make `to` wait for everything enqueued on `from` so far
int stream_after(cfm_stream_t from, cfm_stream_t to) {
    hipEvent_t e = g_events.get();
    if (!e || hipEventRecord(e, (hipStream_t)from) != hipSuccess || hipStreamWaitEvent((hipStream_t)to, e, 0) != hipSuccess)
        return cfm_fail(CFM_ERR_LAUNCH, "train layer: stream hand-off failed");
    return CFM_OK;
}

#define CFM_TRY(expr)                    \
    do {                                 \
        int rc__ = (expr);               \
        if (rc__ != CFM_OK) return rc__; \
    } while (0)

inline const void* eoff(const void* p, int64_t elems, int dt) { return (const char*)p + elems * cfm_elt_size(dt); }
inline void* eoffw(void* p, int64_t elems, int dt) { return (char*)p + elems * cfm_elt_size(dt); }

inline uint32_t site_seed(uint32_t seed, int site) { return seed + 0x9E3779B1u * (uint32_t)site; }

// C = epilogue(A . W^T) with the training options
int gemm(const TCtx& c, const void* A, int a_dt, int64_t lda, const void* W, const void* Wlo, const float* bias, void* C, int c_dt, int64_t ldc, int M, int N,
         int K, int act, const float* res, float alpha, const uint8_t* row_mask, int mask_mode, void* pre, const void* aux, float drop_p, uint32_t drop_seed,
         float drop2_p = 0.f, uint32_t drop2_seed = 0) {
    cfm_gemm_desc d = {};
    d.A = A; d.W = W; d.W_lo = c.split ? Wlo : nullptr; d.bias = bias; d.residual = res; d.row_mask = row_mask; d.C = C;
    d.lda = lda; d.ldc = ldc; d.ldr = ldc; d.M = M; d.N = N; d.K = K;
    d.a_dtype = a_dt; d.w_dtype = c.wdt; d.c_dtype = c_dt; d.act = act; d.alpha = alpha; d.mask_mode = mask_mode;
    if (pre) { d.C_pre = pre; d.ld_pre = N; d.pre_dtype = c.adt; }
    if (aux) { d.aux = aux; d.ld_aux = N; d.aux_dtype = c.adt; }
    d.drop_p = drop_p; d.drop_seed = drop_seed; d.drop2_p = drop2_p; d.drop2_seed = drop2_seed;
    d.tile = CFM_TILE_AUTO_TRAIN;
    if (c.split && !Wlo) return cfm_fail(CFM_ERR_ARG, "train layer: split mode needs the *_lo weight planes");
    return cfm_gemm(&d, c.st);
}

// dW (+)= alpha * A^T . B, db (+)= alpha * colsum(A), accumulated into caller memory.  With a side stream: issued there, after everything the
// main stream has enqueued so far (its operands); the operands must then stay untouched until the streams are joined (end of the backward)
int wgrad(TCtx& c, const void* A, int a_dt, int64_t lda, const void* B, int b_dt, int64_t ldb, float* dW, float* db, int M, int N, int K, float alpha,
          const uint8_t* row_mask, const int64_t* row_off, const int64_t* colsum_off, const int64_t* colsum_off2 = nullptr) {
    cfm_stream_t st = c.st;
    if (c.side) {
        if (int rc = stream_after(c.st, c.side)) return rc;
        st = c.side;
    }
    cfm_gemm_tn_desc d = {};
    d.A = A; d.B = B; d.C = dW; d.colsum = db; d.row_mask = row_mask; d.lda = lda; d.ldb = ldb; d.ldc = K; d.M = M; d.N = N; d.K = K;
    d.a_dtype = a_dt; d.b_dtype = b_dt; d.mma_dtype = c.wdt; d.split = c.split ? 1 : 0; d.accumulate = 1; d.splits = c.io->deterministic ? 1 : 0;
    d.alpha = alpha; d.row_off = row_off; d.colsum_off = colsum_off; d.colsum_off2 = colsum_off2;
    if (c.defer) {
        if (c.npend >= 12) return cfm_fail(CFM_ERR_ARG, "train layer: too many deferred weight-gradient products");
        c.pend[c.npend++] = d;
        return CFM_OK;
    }
    return cfm_gemm_tn(&d, st);
}

int flush_wgrads(TCtx& c) {
    if (!c.npend) return CFM_OK;
    const int n = c.npend;
    c.npend = 0;
    if (c.wg_stream) {                                   // after everything the block enqueued on the main stream (the operands), beside what follows
        if (int rc = stream_after(c.st, c.wg_stream)) return rc;
        return cfm_gemm_tn_group(c.pend, n, c.wg_stream);
    }
    return cfm_gemm_tn_group(c.pend, n, c.st);
}

int ln_fwd(const TCtx& c, const float* x, const float* g, const float* b, void* out, int out_dt, const uint8_t* mask) {
    // masked variant goes through the second output (out2 = mask ? LN : 0)
    if (mask) return cfm_layernorm(x, g, b, nullptr, 0, nullptr, nullptr, out, out_dt, mask, 1e-5f, c.M, c.D, c.st);
    return cfm_layernorm(x, g, b, out, out_dt, nullptr, nullptr, nullptr, 0, nullptr, 1e-5f, c.M, c.D, c.st);
}

// ---- feed-forward sub-block: x_out = x + 1/2 drop_o(W2 drop_h(silu(W1 LN(x) + b1)) + b2) -------------------------------------------------
int ffn_fwd(const TCtx& c, const float* x, const float* lg, const float* lb, const void* w1, const void* w1l, const float* b1, const void* w2, const void* w2l,
            const float* b2, void* xn, void* z, void* h, float* x_out, float p_h, uint32_t s_h, float p_o, uint32_t s_o, bool xn_ready = false,
            const void* w1f = nullptr, const void* w2f = nullptr) {
    if (w1f && w2f && !c.split && cfm_ffn_train_supported(c.D, c.FF)) {
        // ONE launch (csrc/ffn.hip, TRAIN): LayerNorm, both products, both dropout sites, the residual -- the hidden activation goes from the
        // first product's accumulators into the second's operand and is written out only for the backward (25 us against 5 + 19 + 17 at a
        // window's 3 400 rows, scripts/bench_ffn_fused_rows.py)
        cfm_ffn_train_desc d = {};
        d.x = x; d.ln_g = lg; d.ln_b = lb; d.w1f = w1f; d.w2f = w2f; d.b1 = b1; d.b2 = b2; d.y = x_out; d.xn_out = xn; d.z_out = z; d.h_out = h;
        d.M = c.M; d.D = c.D; d.FF = c.FF; d.w_dtype = c.wdt; d.alpha = 0.5f; d.eps = 1e-5f; d.p_hidden = p_h; d.seed_hidden = s_h; d.p_out = p_o; d.seed_out = s_o;
        return cfm_ffn_train_forward(&d, c.st);
    }
    if (!xn_ready) CFM_TRY(ln_fwd(c, x, lg, lb, xn, c.adt, nullptr));
    CFM_TRY(gemm(c, xn, c.adt, c.D, w1, w1l, b1, h, c.adt, c.FF, c.M, c.FF, c.D, CFM_ACT_SILU, nullptr, 0.f, nullptr, 0, z, nullptr, p_h, s_h));
    return gemm(c, h, c.adt, c.FF, w2, w2l, b2, x_out, CFM_F32, c.D, c.M, c.D, c.FF, CFM_ACT_NONE, x, 0.5f, nullptr, 0, nullptr, nullptr, p_o, s_o);
}

// What the NEXT sub-block (in backward order) wants from a LayerNorm backward besides dx: its branch gradient as a GEMM operand,
// dropout(alpha * dx) in the activation dtype -- cfm_dropout_rows folded into the launch that produces dx (buf == nullptr: nothing).
struct Next {
    void* buf;
    float alpha, p1;
    uint32_t s1;
    float p2;
    uint32_t s2;
    const uint8_t* mask;   // the consumer's row mask, applied to the operand itself (its GEMMs then take the unmasked, LDS-DMA paths)
};

// A LayerNorm backward that has NOT been launched: block l+1's norm_ff_macaron, left for block l's first launch, which runs it chained with its
// own norm_final (cfm_ln_bwd_desc.chain_*: the two norms are applied one after the other to the same rows, encoder_layer.py:70,57)
struct PendingLn {
    const float* x;        // the norm's input rows (= block l's output)
    const float* dy;       // f32 gradient of its output (the macaron feed-forward's input gradient, scratch dxn)
    const float* gamma;
    const float* dres;     // the residual stream's gradient at that point (block l+1's d)
    float *gg, *gb;        // where its parameter gradients go (block l+1's slab)
};

// LayerNorm backward of a sub-block: dx = dres + dLN(dy) in place on the residual gradient, parameter gradients into the (zero-filled,
// accumulating) slab -- with atomics in one launch unless the caller asked for reproducible sums -- and the next sub-block's operand
int ln_bwd(const TCtx& c, const cfm_layer_train_scratch* t, const float* x, const void* dy, const float* gamma, const uint8_t* mask, const float* dres, float* dx,
           float* gg, float* gb, const Next& nx, const PendingLn* first = nullptr) {
    cfm_ln_bwd_desc d = {};
    d.x = x; d.dy = dy; d.dy_dtype = CFM_F32; d.gamma = gamma; d.row_mask = mask; d.dres = dres; d.dx = dx; d.dgamma = gg; d.dbeta = gb; d.ws = t->ln_ws;
    if (first) {        // chained: stage 1 = the pending norm, stage 2 = this one (x, gamma, gg, gb as passed; its dy is stage 1's result)
        d.x = first->x; d.dy = first->dy; d.gamma = first->gamma; d.dres = first->dres; d.dgamma = first->gg; d.dbeta = first->gb; d.row_mask = nullptr;
        d.chain_x = x; d.chain_gamma = gamma; d.chain_dgamma = gg; d.chain_dbeta = gb;
    }
    d.accumulate = c.io->deterministic ? 0 : 1;
    d.dx2 = nx.buf; d.dx2_dtype = c.adt; d.alpha2 = nx.alpha; d.p1 = nx.p1; d.seed1 = nx.s1; d.p2 = nx.p2; d.seed2 = nx.s2; d.dx2_row_mask = nx.mask;
    d.eps = 1e-5f; d.M = c.M; d.D = c.D;
    return cfm_layernorm_bwd_fused(&d, c.st);
}

// d (f32 [M,D], the gradient of the sub-block's output) is updated in place to the gradient of its input.  dyb_buf != nullptr: the branch
// gradient through the output dropout is already there (written by the previous LayerNorm backward, see Next)
int ffn_bwd(TCtx& c, const cfm_layer_train_scratch* t, const void* dyb_buf, void* dz_buf, float* d, const float* x, const float* lg, const void* xn, const void* z,
            const void* h, const void* w1t, const void* w1tl, const void* w2t, const void* w2tl, float* gW1, float* gb1, float* gW2, float* gb2, float* glg,
            float* glb, float p_h, uint32_t s_h, const Next& nx, PendingLn* leave = nullptr) {
    const void* dyb = d;
    int dyb_dt = CFM_F32;
    float alpha = 0.5f;
    if (dyb_buf) { dyb = dyb_buf; dyb_dt = c.adt; alpha = 1.0f; }
    CFM_TRY(wgrad(c, dyb, dyb_dt, c.D, h, c.adt, c.FF, gW2, gb2, c.M, c.D, c.FF, alpha, nullptr, nullptr, nullptr));
    CFM_TRY(gemm(c, dyb, dyb_dt, c.D, w2t, w2tl, nullptr, dz_buf, c.adt, c.FF, c.M, c.FF, c.D, CFM_ACT_DSILU, nullptr, alpha, nullptr, 0, nullptr, z, p_h, s_h));
    CFM_TRY(wgrad(c, dz_buf, c.adt, c.FF, xn, c.adt, c.D, gW1, gb1, c.M, c.FF, c.D, 1.0f, nullptr, nullptr, nullptr));
    CFM_TRY(gemm(c, dz_buf, c.adt, c.FF, w1t, w1tl, nullptr, t->dxn, CFM_F32, c.D, c.M, c.D, c.FF, CFM_ACT_NONE, nullptr, 0.f, nullptr, 0, nullptr, nullptr, 0.f, 0));
    if (leave) {                                          // the block below runs this norm's backward chained with its own norm_final
        *leave = {x, t->dxn, lg, d, glg, glb};
        return CFM_OK;
    }
    return ln_bwd(c, t, x, t->dxn, lg, nullptr, d, d, glg, glb, nx);
}

// xn1_ready: the previous block's last launch already wrote this block's norm_ff_macaron output (sv->xn1); next_w / next_sv: the NEXT block, whose
// norm_ff_macaron is chained onto this block's norm_final in one launch (the two LayerNorms of consecutive blocks, encoder_layer.py:70,57)
int layer_forward(TCtx& c, const cfm_layer_train_saved* sv, const cfm_layer_train_scratch* t, const float* x_in, float* y_out, bool xn1_ready = false,
                  const cfm_layer_train_weights* next_w = nullptr, const cfm_layer_train_saved* next_sv = nullptr) {
    const cfm_layer_train_weights* w = c.w;
    const cfm_layer_train_io* io = c.io;
    cfm_stream_t stream = c.st;
    const int M = c.M, D = c.D, adt = c.adt;
    const uint32_t sd = io->seed;
    // (1) macaron feed-forward
    CFM_TRY(ffn_fwd(c, x_in, w->ln_ffm_g, w->ln_ffm_b, w->ffm_w1, w->ffm_w1_lo, w->ffm_b1, w->ffm_w2, w->ffm_w2_lo, w->ffm_b2, sv->xn1, sv->z1, sv->h1, sv->x1,
                    io->p_hidden_m, site_seed(sd, 1), io->p_branch, site_seed(sd, 2), xn1_ready, w->ffm_w1f, w->ffm_w2f));
    // (2) self-attention: q + pos_bias_u rides in the projection's bias; the batch path's positional term is softmax-invariant
    CFM_TRY(ln_fwd(c, sv->x1, w->ln_mha_g, w->ln_mha_b, sv->xn2, adt, nullptr));
    CFM_TRY(gemm(c, sv->xn2, adt, D, w->qkv_w, w->qkv_w_lo, w->qkv_b, sv->qkv, adt, 3 * D, M, 3 * D, D, CFM_ACT_NONE, nullptr, 0.f, nullptr, 0, nullptr, nullptr, 0.f, 0));
    {                                                        // attention mixes the frames of ONE utterance: a problem per micro-batch (its own T and mask), one launch
        cfm_attn_desc ad[CFM_TRAIN_MAX_GROUPS];
        for (int gi = 0; gi < c.ng; ++gi) {
            const Grp& G = c.grp[gi];
            cfm_attn_desc a = {};
            const int64_t sb = (int64_t)G.T * 3 * D, stt = 3 * D;
            const void* qkv = eoff(sv->qkv, G.row0 * 3 * D, adt);
            a.q = qkv; a.k = eoff(qkv, D, adt); a.v = eoff(qkv, 2 * D, adt);
            a.q_sb = a.k_sb = a.v_sb = sb; a.q_st = a.k_st = a.v_st = stt; a.k_sh = a.v_sh = c.dk;
            a.q_dtype = a.kv_dtype = adt; a.out = eoffw(sv->ctx, G.row0 * D, adt); a.out_dtype = adt;
            a.mask = G.mask; a.m_sb = G.sb; a.m_sq = G.sq;
            a.B = G.B; a.H = c.H; a.Tq = a.Tk = G.T; a.dk = c.dk; a.mma_dtype = c.wdt; a.split = c.split ? 1 : 0; a.scale = 1.0f / sqrtf((float)c.dk);
            a.lse = sv->lse + G.bht0; a.drop_p = io->p_attn; a.drop_seed = site_seed(sd, 3) + 0x7F4A7C15u * (uint32_t)gi;
            ad[gi] = a;
        }
        CFM_TRY(c.ng == 1 ? cfm_attention(&ad[0], stream) : cfm_attention_group(ad, c.ng, stream));
    }
    {
        float p1 = io->p_branch, p2 = io->p_attn_out;
        uint32_t s1 = site_seed(sd, 4), s2 = site_seed(sd, 5);
        if (p1 <= 0.f && p2 > 0.f) { p1 = p2; s1 = s2; p2 = 0.f; }
        CFM_TRY(gemm(c, sv->ctx, adt, D, w->out_w, w->out_w_lo, w->out_b, sv->x2, CFM_F32, D, M, D, D, CFM_ACT_NONE, sv->x1, 1.0f, nullptr, 0, nullptr, nullptr, p1, s1, p2, s2));
    }
    // (3) convolution module, BatchNorm in training mode: statistics per micro-batch, running statistics updated group after group
    CFM_TRY(ln_fwd(c, sv->x2, w->ln_conv_g, w->ln_conv_b, sv->xn3, adt, io->pad_valid));
    CFM_TRY(gemm(c, sv->xn3, adt, D, w->pw1_w, w->pw1_w_lo, w->pw1_b, sv->glu, adt, D, M, 2 * D, D, CFM_ACT_GLU, nullptr, 0.f, nullptr, 0, sv->u, nullptr, 0.f, 0));
    CFM_TRY(cfm_dwconv_bn_train_groups(sv->glu, adt, w->dw_w, w->dw_b, w->bn_gamma, w->bn_beta, w->bn_running_mean, w->bn_running_var, w->bn_momentum, w->bn_eps, sv->c,
                                       sv->stats, sv->s, adt, t->dwbn_ws, c.cg, c.ng, D, io->ktaps, stream));
    CFM_TRY(gemm(c, sv->s, adt, D, w->pw2_w, w->pw2_w_lo, w->pw2_b, sv->x3, CFM_F32, D, M, D, D, CFM_ACT_NONE, sv->x2, 1.0f, io->pad_valid, 0, nullptr, nullptr,
                 io->p_branch, site_seed(sd, 6)));
    // (4) feed-forward, (5) norm_final
    CFM_TRY(ffn_fwd(c, sv->x3, w->ln_ff_g, w->ln_ff_b, w->ff_w1, w->ff_w1_lo, w->ff_b1, w->ff_w2, w->ff_w2_lo, w->ff_b2, sv->xn4, sv->z2, sv->h2, sv->x4, io->p_hidden,
                    site_seed(sd, 7), io->p_branch, site_seed(sd, 8), false, w->ff_w1f, w->ff_w2f));
    if (next_w && next_sv)
        return cfm_layernorm(sv->x4, w->ln_final_g, w->ln_final_b, y_out, CFM_F32, next_w->ln_ffm_g, next_w->ln_ffm_b, next_sv->xn1, adt, nullptr, 1e-5f, M, D, stream);
    return cfm_layernorm(sv->x4, w->ln_final_g, w->ln_final_b, y_out, CFM_F32, nullptr, nullptr, nullptr, 0, nullptr, 1e-5f, M, D, stream);
}

// first: the block above left its norm_ff_macaron backward pending (dy is then that block's residual gradient, first->dres); leave: leave THIS
// block's norm_ff_macaron backward to the block below (dx then holds the gradient BEFORE that norm: the caller passes it on as first->dres)
int layer_backward(TCtx& c, const cfm_layer_train_saved* sv, const cfm_layer_train_scratch* t, const cfm_layer_train_grads* g, const float* x_in, const float* dy,
                   float* dx, const PendingLn* first = nullptr, PendingLn* leave = nullptr) {
    const cfm_layer_train_weights* w = c.w;
    const cfm_layer_train_io* io = c.io;
    cfm_stream_t stream = c.st;
    c.side = io->side_stream && io->side_stream != stream ? io->side_stream : nullptr;
    // deferral keeps every product's operands until the block's last launch: same buffers as the side-stream variant; the grouped kernel
    // takes 16-bit operands (the f32-accurate mode keeps its immediate, single products).  Deferral AND a side stream: the one grouped
    // launch goes to the side stream (c.wg_stream), everything else stays on the main one
    c.defer = io->defer_wgrad && !c.split;
    if (c.defer) { c.wg_stream = c.side; c.side = nullptr; }
    const bool keep_ops = c.side || c.defer;
    CFM_CHECK_ARG(!keep_ops || (t->dz2 && t->dyb2 && t->dyb3 && t->dyb4), "cfm_encoder_layer_train_backward: a side stream / deferred weight gradients need the dz2 / dyb2..4 scratch buffers");
    CFM_CHECK_ARG(!io->grads_accumulate || !io->deterministic, "cfm_encoder_layer_train_backward: grads_accumulate needs deterministic == 0");
    const int M = c.M, D = c.D, adt = c.adt;
    const uint32_t sd = io->seed;
    float* d = dx;                                        // the residual stream's gradient, updated in place from the block's output to its input
    // Each sub-block's branch gradient (dropout mask * alpha * d, act dtype) is written by the LayerNorm backward that produces d -- when it is
    // needed at all: with dropout, or when the operands must outlive the sub-block (d is overwritten by the sub-block's own LayerNorm backward
    // while a weight-gradient product on the side stream / at the end of the block still reads its operand; one buffer per sub-block then)
    const bool br = io->p_branch > 0.f || keep_ops;
    float pa1 = io->p_branch, pa2 = io->p_attn_out;
    uint32_t sa1 = site_seed(sd, 4), sa2 = site_seed(sd, 5);
    if (pa1 <= 0.f && pa2 > 0.f) { pa1 = pa2; sa1 = sa2; pa2 = 0.f; }
    const bool bra = pa1 > 0.f || keep_ops;
    const Next n_ff = {br ? t->dyb : nullptr, 0.5f, io->p_branch, site_seed(sd, 8), 0.f, 0, nullptr};
    const Next n_conv = {br ? (keep_ops ? t->dyb2 : t->dyb) : nullptr, 1.0f, io->p_branch, site_seed(sd, 6), 0.f, 0, io->pad_valid};
    const Next n_att = {bra ? (keep_ops ? t->dyb3 : t->dyb) : nullptr, 1.0f, pa1, sa1, pa2, sa2, nullptr};
    const Next n_ffm = {br ? (keep_ops ? t->dyb4 : t->dyb) : nullptr, 0.5f, io->p_branch, site_seed(sd, 2), 0.f, 0, nullptr};
    const Next n_none = {nullptr, 0.f, 0.f, 0, 0.f, 0, nullptr};
    // (5) norm_final
    CFM_TRY(ln_bwd(c, t, sv->x4, dy, w->ln_final_g, nullptr, nullptr, d, g->ln_final_g, g->ln_final_b, n_ff, first));
    // (4) feed-forward
    CFM_TRY(ffn_bwd(c, t, n_ff.buf, t->dz, d, sv->x3, w->ln_ff_g, sv->xn4, sv->z2, sv->h2, w->ff_w1t, w->ff_w1t_lo, w->ff_w2t, w->ff_w2t_lo, g->ff_w1, g->ff_b1, g->ff_w2, g->ff_b2,
                    g->ln_ff_g, g->ln_ff_b, io->p_hidden, site_seed(sd, 7), n_conv));
    // (3) convolution module: x3 = x2 + mask * drop(s . Wpw2^T + b)
    {
        const void* dyb = d;
        int dyb_dt = CFM_F32;
        const uint8_t* pm = io->pad_valid;
        if (n_conv.buf) { dyb = n_conv.buf; dyb_dt = adt; pm = nullptr; }   // the padded rows of the operand are already zero (Next.mask)
        CFM_TRY(wgrad(c, dyb, dyb_dt, D, sv->s, adt, D, g->pw2_w, g->pw2_b, M, D, D, 1.0f, pm, nullptr, nullptr));
        CFM_TRY(gemm(c, dyb, dyb_dt, D, w->pw2_t, w->pw2_t_lo, nullptr, t->ds, adt, D, M, D, D, CFM_ACT_NONE, nullptr, 0.f, pm, pm ? 1 : 0, nullptr, nullptr, 0.f, 0));
        // per micro-batch its own BatchNorm statistics, one launch per stage for all of them; the parameter gradients are summed over the micro-batches
        // ... and the GLU backward rides in the depthwise launch (du straight from the rounded dg; the separate elementwise pass is gone)
        CFM_TRY(cfm_dwconv_bn_train_bwd_groups(t->ds, adt, sv->c, sv->stats, sv->glu, adt, w->dw_w, t->dglu, adt, g->dw_w, g->dw_b, g->bn_g, g->bn_b, t->dy_ws, t->dwbn_ws,
                                               c.cg, c.ng, D, io->ktaps, io->grads_accumulate ? 1 : 0, sv->u, t->du, stream));
        CFM_TRY(wgrad(c, t->du, adt, 2 * D, sv->xn3, adt, D, g->slab, g->slab, M, 2 * D, D, 1.0f, nullptr, g->pw1_row_off, g->pw1_bias_off));
        CFM_TRY(gemm(c, t->du, adt, 2 * D, w->pw1_t, w->pw1_t_lo, nullptr, t->dxn, CFM_F32, D, M, D, 2 * D, CFM_ACT_NONE, nullptr, 0.f, nullptr, 0, nullptr, nullptr, 0.f, 0));
        CFM_TRY(ln_bwd(c, t, sv->x2, t->dxn, w->ln_conv_g, io->pad_valid, d, d, g->ln_conv_g, g->ln_conv_b, n_att));
    }
    // (2) self-attention: x2 = x1 + drop(ctx . Wo^T + bo)
    {
        const void* dyb = d;
        int dyb_dt = CFM_F32;
        if (n_att.buf) { dyb = n_att.buf; dyb_dt = adt; }
        CFM_TRY(wgrad(c, dyb, dyb_dt, D, sv->ctx, adt, D, g->out_w, g->out_b, M, D, D, 1.0f, nullptr, nullptr, nullptr));
        CFM_TRY(gemm(c, dyb, dyb_dt, D, w->out_t, w->out_t_lo, nullptr, t->dctx, adt, D, M, D, D, CFM_ACT_NONE, nullptr, 0.f, nullptr, 0, nullptr, nullptr, 0.f, 0));
        {
            cfm_attn_bwd_desc bd[CFM_TRAIN_MAX_GROUPS];
            for (int gi = 0; gi < c.ng; ++gi) {
                const Grp& G = c.grp[gi];
                cfm_attn_bwd_desc b = {};
                const int64_t sb = (int64_t)G.T * 3 * D, stt = 3 * D;
                const void* qkv = eoff(sv->qkv, G.row0 * 3 * D, adt);
                void* dqkv = eoffw(t->dqkv, G.row0 * 3 * D, adt);
                b.q = qkv; b.k = eoff(qkv, D, adt); b.v = eoff(qkv, 2 * D, adt);
                b.mask = G.mask; b.out = eoff(sv->ctx, G.row0 * D, adt); b.dout = eoff(t->dctx, G.row0 * D, adt); b.lse = sv->lse + G.bht0;
                b.grad_q = dqkv; b.grad_k = eoffw(dqkv, D, adt); b.grad_v = eoffw(dqkv, 2 * D, adt); b.delta = t->delta + G.bht0;
                b.q_sb = b.k_sb = b.v_sb = sb; b.q_st = b.k_st = b.v_st = stt; b.m_sb = G.sb; b.m_sq = G.sq;
                b.B = G.B; b.H = c.H; b.Tq = b.Tk = G.T; b.dk = c.dk; b.io_dtype = adt; b.dout_dtype = adt; b.mma_dtype = c.wdt; b.split = c.split ? 1 : 0;
                b.scale = 1.0f / sqrtf((float)c.dk); b.drop_p = io->p_attn; b.drop_seed = site_seed(sd, 3) + 0x7F4A7C15u * (uint32_t)gi;
                bd[gi] = b;
            }
            CFM_TRY(c.ng == 1 ? cfm_attention_bwd(&bd[0], stream) : cfm_attention_bwd_group(bd, c.ng, stream));
        }
        // d/du of (q + u) . k^T = d/d(linear_q.bias): the column sums of dq -- a second scatter table of the same product, or a copy after it
        const bool u_table = g->pos_bias_u && g->qkv_bias_off2;
        CFM_TRY(wgrad(c, t->dqkv, adt, 3 * D, sv->xn2, adt, D, g->slab, g->slab, M, 3 * D, D, 1.0f, nullptr, g->qkv_row_off, g->qkv_bias_off, u_table ? g->qkv_bias_off2 : nullptr));
        if (g->pos_bias_u && g->q_bias && !u_table) {
            CFM_CHECK_ARG(!c.defer, "cfm_encoder_layer_train_backward: deferred weight gradients need grads.qkv_bias_off2 for pos_bias_u");
            if (hipMemcpyAsync(g->pos_bias_u, g->q_bias, (size_t)D * 4, hipMemcpyDeviceToDevice, (hipStream_t)(c.side ? c.side : stream)) != hipSuccess)
                return cfm_fail(CFM_ERR_LAUNCH, "train layer: copy of the pos_bias_u gradient failed");
        }
        CFM_TRY(gemm(c, t->dqkv, adt, 3 * D, w->qkv_t, w->qkv_t_lo, nullptr, t->dxn, CFM_F32, D, M, D, 3 * D, CFM_ACT_NONE, nullptr, 0.f, nullptr, 0, nullptr, nullptr, 0.f, 0));
        CFM_TRY(ln_bwd(c, t, sv->x1, t->dxn, w->ln_mha_g, nullptr, d, d, g->ln_mha_g, g->ln_mha_b, n_ffm));
    }
    // (1) macaron feed-forward
    CFM_TRY(ffn_bwd(c, t, n_ffm.buf, keep_ops ? t->dz2 : t->dz, d, x_in, w->ln_ffm_g, sv->xn1, sv->z1, sv->h1, w->ffm_w1t, w->ffm_w1t_lo, w->ffm_w2t,
                    w->ffm_w2t_lo, g->ffm_w1, g->ffm_b1, g->ffm_w2, g->ffm_b2, g->ln_ffm_g, g->ln_ffm_b, io->p_hidden_m, site_seed(sd, 1), n_none, leave));
    CFM_TRY(flush_wgrads(c));                            // deferred: the block's eight weight-gradient products, one launch
    if (c.side) return stream_after(c.side, c.st);       // join: the block's gradients are complete when the main stream gets past this point
    return CFM_OK;                                       // (c.wg_stream: the caller joins -- cfm_encoder_layer_train_backward at once, the stack one block later)
}

int check_io(const cfm_layer_train_io* io, const char* who) {
    CFM_CHECK_ARG(io->D > 0 && io->H > 0 && io->D % io->H == 0 && io->FF > 0 && io->D % 16 == 0 && (io->n_groups > 0 || (io->B > 0 && io->T > 0)),
                  "%s: bad dims B=%d T=%d D=%d H=%d FF=%d", who, io->B, io->T, io->D, io->H, io->FF);
    return CFM_OK;
}

// per-layer dropout seeds of a stack: layer l draws from seed + l * odd constant (0 stays 0: no dropout anywhere)
inline uint32_t layer_seed(uint32_t seed, int l) { return seed ? seed + 0x632BE5ABu * (uint32_t)l : 0u; }

}  // namespace

extern "C" int cfm_encoder_layer_train_forward(const cfm_layer_train_weights* w, const cfm_layer_train_io* io, const cfm_layer_train_saved* sv,
                                               const cfm_layer_train_scratch* t, const float* x_in, float* y_out, cfm_stream_t stream) {
    CFM_CHECK_ARG(w && io && sv && t && x_in && y_out, "cfm_encoder_layer_train_forward: null pointer");
    CFM_TRY(check_io(io, "cfm_encoder_layer_train_forward"));
    TCtx c;
    CFM_TRY(init_ctx(c, w, io, stream));
    return layer_forward(c, sv, t, x_in, y_out);
}

extern "C" int cfm_encoder_layer_train_backward(const cfm_layer_train_weights* w, const cfm_layer_train_io* io, const cfm_layer_train_saved* sv,
                                                const cfm_layer_train_scratch* t, const cfm_layer_train_grads* g, const float* x_in, const float* dy,
                                                float* dx, cfm_stream_t stream) {
    CFM_CHECK_ARG(w && io && sv && t && g && x_in && dy && dx && dx != dy, "cfm_encoder_layer_train_backward: null pointer (dx must not alias dy)");
    CFM_TRY(check_io(io, "cfm_encoder_layer_train_backward"));
    TCtx c;
    CFM_TRY(init_ctx(c, w, io, stream));
    CFM_TRY(layer_backward(c, sv, t, g, x_in, dy, dx));
    if (c.wg_stream) return stream_after(c.wg_stream, c.st);
    return CFM_OK;
}

// ---- the whole stack (encoder.py:72-73 `for block in self.encoders` under module.train()) from one host call each way ------------------------
extern "C" int cfm_encoder_train_forward(int32_t n_layers, const cfm_layer_train_weights* w, const cfm_layer_train_io* io, const cfm_layer_train_saved* sv,
                                         const cfm_layer_train_scratch* t, float* const* xs, cfm_stream_t stream) {
    CFM_CHECK_ARG(n_layers > 0 && w && io && sv && t && xs, "cfm_encoder_train_forward: null pointer");
    CFM_TRY(check_io(io, "cfm_encoder_train_forward"));
    for (int l = 0; l < n_layers; ++l) {
        CFM_CHECK_ARG(xs[l] && xs[l + 1], "cfm_encoder_train_forward: xs[%d] is null", l);
        cfm_layer_train_io iol = *io;
        iol.seed = layer_seed(io->seed, l);
        TCtx c;
        CFM_TRY(init_ctx(c, &w[l], &iol, stream));
        // norm_final of block l and norm_ff_macaron of block l+1: one launch -- unless block l+1's macaron feed-forward is the fused launch, which
        // normalises its rows itself
        const bool fused_next = l + 1 < n_layers && w[l + 1].ffm_w1f && w[l + 1].ffm_w2f && io->act_dtype != CFM_F32 && cfm_ffn_train_supported(io->D, io->FF);
        const bool chain = l + 1 < n_layers && !fused_next;
        CFM_TRY(layer_forward(c, &sv[l], t, xs[l], xs[l + 1], l > 0, chain ? &w[l + 1] : nullptr, chain ? &sv[l + 1] : nullptr));
    }
    return CFM_OK;
}

extern "C" int cfm_encoder_train_backward(int32_t n_layers, const cfm_layer_train_weights* w, const cfm_layer_train_io* io, const cfm_layer_train_saved* sv,
                                          const cfm_layer_train_scratch* t, int32_t n_scratch, const cfm_layer_train_grads* g, float* const* xs, const float* dy,
                                          float* dbuf0, float* dbuf1, cfm_layer_done_fn done, void* user, float** dx_out, cfm_stream_t stream) {
    CFM_CHECK_ARG(n_layers > 0 && w && io && sv && t && g && xs && dy && dbuf0 && dbuf1 && dbuf0 != dbuf1 && dx_out && dy != dbuf0 && dy != dbuf1,
                  "cfm_encoder_train_backward: null or aliased pointer");
    CFM_CHECK_ARG(n_scratch == 1 || n_scratch == 2, "cfm_encoder_train_backward: n_scratch must be 1 or 2");
    CFM_TRY(check_io(io, "cfm_encoder_train_backward"));
    // Weight gradients beside the chain: with io->side_stream and io->defer_wgrad block l's grouped launch runs on the side stream while the main
    // stream goes on with block l-1.  Its operands live in scratch set l & 1 (two sets: n_scratch == 2), which block l-2 reuses -- the main
    // stream waits for block l's launch before block l-2 starts (two blocks later: never a stall in practice), and `done(l)` is reported one
    // block late, after that wait, so that whoever reduces block l's gradients from the MAIN stream sees them complete.
    const bool beside = io->side_stream && io->side_stream != stream && io->defer_wgrad && io->act_dtype != CFM_F32;
    CFM_CHECK_ARG(!beside || n_scratch == 2, "cfm_encoder_train_backward: weight gradients on a side stream need two scratch sets");
    static thread_local hipEvent_t wg_done[2] = {nullptr, nullptr};
    if (beside)
        for (int i = 0; i < 2; ++i)
            if (!wg_done[i] && hipEventCreateWithFlags(&wg_done[i], hipEventDisableTiming) != hipSuccess)
                return cfm_fail(CFM_ERR_LAUNCH, "cfm_encoder_train_backward: event creation failed");
    // consecutive blocks: block l+1's last LayerNorm backward (norm_ff_macaron) and block l's first (norm_final) act on the same rows one after the
    // other -- block l+1 leaves its own pending and block l's first launch runs both (cfm_ln_bwd_desc.chain_*); needs the atomic parameter sums
    const bool chain_ln = !io->deterministic;
    const float* cur = dy;
    float* bufs[2] = {dbuf0, dbuf1};
    int k = 0, report = -1;                               // report: a block whose gradients are complete only after the NEXT block's first launch(es)
    PendingLn pend = {}, left = {};
    bool have = false;
    for (int l = n_layers - 1; l >= 0; --l) {
        cfm_layer_train_io iol = *io;
        iol.seed = layer_seed(io->seed, l);
        TCtx c;
        CFM_TRY(init_ctx(c, &w[l], &iol, stream));
        const int set = n_scratch == 2 ? (l & 1) : 0;
        if (beside && l + 2 < n_layers && hipStreamWaitEvent((hipStream_t)stream, wg_done[set], 0) != hipSuccess)       // block l+2's products have read this set
            return cfm_fail(CFM_ERR_LAUNCH, "cfm_encoder_train_backward: stream wait failed");
        const bool leave = chain_ln && l > 0;
        CFM_TRY(layer_backward(c, &sv[l], &t[set], &g[l], xs[l], cur, bufs[k], have ? &pend : nullptr, leave ? &left : nullptr));
        cur = bufs[k];
        k ^= 1;
        have = leave;
        pend = left;
        if (beside && hipEventRecord(wg_done[set], (hipStream_t)io->side_stream) != hipSuccess)
            return cfm_fail(CFM_ERR_LAUNCH, "cfm_encoder_train_backward: event record failed");
        if (report >= 0) {                                // block l+1: its pending LayerNorm sums and (beside) its weight gradients are now enqueued / awaited
            if (beside && hipStreamWaitEvent((hipStream_t)stream, wg_done[set ^ 1], 0) != hipSuccess) return cfm_fail(CFM_ERR_LAUNCH, "cfm_encoder_train_backward: stream wait failed");
            if (done) done(report, user);
            report = -1;
        }
        if (beside || leave) report = l;
        else if (done) done(l, user);                     // every launch of block l's backward is enqueued: its gradients may be reduced
    }
    if (beside)
        if (int rc = stream_after(io->side_stream, stream)) return rc;     // join: block 0's (and, with it, every) weight gradient
    if (report >= 0 && done) done(report, user);
    *dx_out = (float*)cur;
    return CFM_OK;
}
