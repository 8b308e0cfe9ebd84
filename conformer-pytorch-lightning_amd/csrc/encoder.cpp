// encoder.cpp -- composite: one conformer block enqueued from C++ (2 launches when consecutive blocks are chained: attention, then the conv-in chain + depthwise + final chain
// + the next block's macaron chain in one; 3 launches with the row-local chains of rowchain.hip when the attention rides in the
// conv-in chain, 4 otherwise, 17 on the general path; no host sync).
//
// Mirrors reference src/encoder_layer.py:49-71:
//   x = x + 1/2 FFNm(LN(x)); x = x + MHSA(LN(x)); x = x + Conv(LN(x)); x = x + 1/2 FFN(LN(x)); out = LN(x)
// with every residual add, bias, activation, GLU and padding mask folded into a GEMM epilogue and the
// residual stream kept in f32.  Layer norms write the GEMM operand dtype directly.
#include <math.h>

#include <stdlib.h>

#include "cfm_common.h"

namespace {

inline const void* eoff(const void* p, int64_t elems, int dt) { return (const char*)p + elems * cfm_elt_size(dt); }

struct Ctx {
    const cfm_layer_io* io;
    int M, D, FF, act_dt, w_dt;
    bool split;
    cfm_stream_t st;
};

int gemm(const Ctx& c, const void* A, int a_dt, int64_t lda, const void* W, const void* Wlo, const float* bias, void* C, int c_dt,
         int64_t ldc, int M, int N, int K, int act, const float* res, float alpha, const uint8_t* row_mask) {
    cfm_gemm_desc d = {};
    d.A = A; d.W = W; d.W_lo = c.split ? Wlo : nullptr; d.bias = bias; d.residual = res; d.row_mask = row_mask; d.C = C;
    d.lda = lda; d.ldc = ldc; d.ldr = ldc; d.M = M; d.N = N; d.K = K;
    d.a_dtype = a_dt; d.w_dtype = c.w_dt; d.c_dtype = c_dt; d.act = act; d.alpha = alpha;
    if (c.split && !Wlo) return cfm_fail(CFM_ERR_ARG, "encoder layer: split mode needs the *_lo weight planes");
    return cfm_gemm(&d, c.st);
}

int ffn_fused(const Ctx& c, const float* x, const float* ln_g, const float* ln_b, const void* w1f, const void* w2f, const float* b1,
              const float* b2, const float* ln1_g, const float* ln1_b, const float* ln2_g, const float* ln2_b, float* out_f32,
              void* out16) {
    cfm_ffn_desc d = {};
    d.x = x; d.ln_g = ln_g; d.ln_b = ln_b; d.w1f = w1f; d.w2f = w2f; d.b1 = b1; d.b2 = b2;
    d.ln1_g = ln1_g; d.ln1_b = ln1_b; d.ln2_g = ln2_g; d.ln2_b = ln2_b; d.out_f32 = out_f32; d.out16 = out16;
    d.M = c.M; d.D = c.D; d.FF = c.FF; d.w_dtype = c.w_dt; d.out16_dtype = c.act_dt; d.act = CFM_ACT_SILU; d.add_x = 1;
    d.alpha = 0.5f; d.eps = 1e-5f;
    return cfm_ffn_fused(&d, c.st);
}

int& cin_merge_flag() {
    static int flag = getenv("CFM_CIN_MERGE") == nullptr || atoi(getenv("CFM_CIN_MERGE")) != 0;
    return flag;
}

#define CFM_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__ != CFM_OK) return rc__; \
    } while (0)

}  // namespace

extern "C" int32_t cfm_set_cin_merge(int32_t on) {
    const int prev = cin_merge_flag();
    cin_merge_flag() = on != 0;
    return prev;
}

extern "C" int cfm_encoder_layer_forward(const cfm_layer_weights* w, const cfm_layer_scratch* s, const cfm_layer_io* io,
                                         const float* x_in, float* x_out, int xn_ready, const float* next_g,
                                         const float* next_b, cfm_stream_t stream) {
    CFM_CHECK_ARG(w && s && io && x_in && x_out, "cfm_encoder_layer_forward: null pointer");
    CFM_CHECK_ARG(io->B > 0 && io->T > 0 && io->D > 0 && io->H > 0 && io->D % io->H == 0 && io->FF > 0,
                  "cfm_encoder_layer_forward: bad dims B=%d T=%d D=%d H=%d FF=%d", io->B, io->T, io->D, io->H, io->FF);
    CFM_CHECK_ARG(io->D % 16 == 0, "cfm_encoder_layer_forward: D must be a multiple of 16 (GLU interleave)");
    CFM_CHECK_ARG(x_in != x_out, "cfm_encoder_layer_forward: x_in and x_out must differ (inputs are not mutated)");
    Ctx c;
    c.io = io; c.M = io->B * io->T; c.D = io->D; c.FF = io->FF; c.act_dt = io->act_dtype; c.w_dt = io->w_dtype;
    c.split = io->act_dtype == CFM_F32;
    c.st = stream;
    const int M = c.M, D = c.D, FF = c.FF, H = io->H, dk = D / H, adt = c.act_dt;
    const float eps = 1e-5f;
    const bool has_pos = io->pos_rows > 0 && w->pos_w;
    const bool ring = io->kv_ring != nullptr;
    CFM_CHECK_ARG(!ring || (io->stream_offset && io->ring_T >= io->T && !io->attn_cache && !io->new_cache),
                  "encoder layer: the K/V ring needs stream_offset and ring_T >= T, and excludes attn_cache / new_cache");
    const int Tc = io->attn_cache ? io->cache_T : 0;
    const int Tk = ring ? io->ring_T : Tc + io->T;
    int P = 0;
    if (has_pos) {
        CFM_CHECK_ARG((io->pos_embed || io->pos_proj) && (io->pos_shared || io->pos_rows % io->B == 0),
                      "encoder layer: pos_embed rows (%d) must be a multiple of B (%d)", io->pos_rows, io->B);
        P = io->pos_shared ? io->pos_rows : io->pos_rows / io->B;
        CFM_CHECK_ARG(!io->pos_shared || P == Tk, "encoder layer: shared positions need one row per key (%d rows, Tk=%d)", io->pos_rows, Tk);
        CFM_CHECK_ARG(P == 1 || P == Tk, "encoder layer: pos_embed gives %d rows per item, need 1 or Tk=%d (attention.py:78-88)", P, Tk);
    }
    CFM_CHECK_ARG(Tc == 0 || io->new_cache, "encoder layer: a KV cache input needs new_cache storage");

    // ---- 6-launch path: the three row-local chains of rowchain.hip -----------------------------------------------------
    const bool chains = !c.split && w->ffm_w1f && w->ffm_w2n && w->ff_w1f && w->ff_w2n && w->qkv_wf && w->out_wf && w->pw1_wf &&
                        w->pw2_wf && cfm_rowchain_supported(D, FF);
    CFM_CHECK_ARG(!io->after_out || (io->after_g && io->after_b), "encoder layer: after_out needs after_g / after_b");
    // after_out outside the chain path: one more LayerNorm launch at the end (same result, nothing fused)
    auto after_tail = [&]() -> int {
        if (!io->after_out) return CFM_OK;
        return cfm_layernorm(x_out, io->after_g, io->after_b, io->after_out, CFM_F32, nullptr, nullptr, nullptr, 0, nullptr, eps, M, D, stream);
    };
    // 3-launch path: the attention runs as the input stage of the conv-in chain (batch path only: no cache / ring, key-validity mask, at most
    // one positional row per item, T <= 256, 4 heads x 64) on values the macaron chain's tail wrote transposed
    const bool merged = chains && s->vt && s->vt_ld >= 256 && s->vt_ld % 4 == 0 && D == 256 && H == 4 && !ring && Tc == 0 && !io->new_cache &&
                        io->T <= 256 && (!has_pos || P == 1) && (!io->attn_mask || io->am_sq == 0) && !io->macaron_done && !io->next_w;
    CFM_CHECK_ARG(!io->macaron_done || chains, "encoder layer: macaron_done needs the chain path");
    // few rows (a streaming step): both feed-forwards split over FF / 256 workgroups per 32-row tile (ffnsplit.hip) instead of inside the row
    // chains, where every tile's workgroup streams all 2 MB of a feed-forward's weights whatever the row count
    bool ring_written = false;                            // the split path's q|k|v launch also filled the K/V ring
    static const int ffsplit_rows = getenv("CFM_FFSPLIT_MAX_ROWS") ? atoi(getenv("CFM_FFSPLIT_MAX_ROWS")) : CFM_FFSPLIT_MAX_ROWS;   // experiments (scripts/bench_small_batch.py)
    const bool ffsplit = chains && s->psum && M <= ffsplit_rows && cfm_ffn_split_supported(D, FF) && s->psum_splits >= FF / 256 && !merged &&
                         !io->macaron_done && !io->next_w && w->pw2_w && io->ktaps == 15;
    auto split_desc = [&](int mode) {
        cfm_ffn_split_desc f = {};
        f.M = M; f.D = D; f.mode = mode; f.w_dtype = c.w_dt; f.eps = eps;
        return f;
    };
    // D = 512 with at most one row tile per CU pair (config 4: 125 tiles): both feed-forwards split over workgroup pairs, one half of FF each
    // (rowchain.hip FSPLIT); the halves meet in the next launch's row load
    static const int pair_rows = getenv("CFM_PAIR_MAX_ROWS") ? atoi(getenv("CFM_PAIR_MAX_ROWS")) : CFM_PAIR_MAX_ROWS;
    const bool pair = chains && s->psum && s->psum_splits >= 3 && M <= pair_rows && cfm_rowchain_pair_supported(D, FF) && !io->macaron_done && !io->next_w;
    if (pair) {
        cfm_rowchain_desc m = {};
        m.x = x_in; m.ln_g = w->ln_ffm_g; m.ln_b = w->ln_ffm_b; m.w1f = w->ffm_w1f; m.w2n = w->ffm_w2n; m.b1 = w->ffm_b1; m.b2 = w->ffm_b2;
        m.psum_out = s->psum; m.M = M; m.D = D; m.FF = FF; m.w_dtype = c.w_dt; m.alpha = 0.5f; m.eps = eps;
        CFM_TRY(cfm_rowchain(&m, stream));
        cfm_rowchain_desc q = {};                         // rows = x + 1/2 (half 0 + half 1 + b2) -> x_out, LN_mha, fused q|k|v projection
        q.x = x_in; q.psum_in = s->psum; q.psum_b2 = w->ffm_b2; q.psum_alpha = 0.5f; q.out_f32 = x_out; q.ln_g = w->ln_mha_g; q.ln_b = w->ln_mha_b;
        q.tail_w = w->qkv_wf; q.tail_b = w->qkv_b; q.tail_out = s->qkv; q.tail_N = 3 * D; q.M = M; q.D = D; q.FF = FF; q.w_dtype = c.w_dt; q.alpha = 1.0f; q.eps = eps;
        q.tail_pair = 1;
        CFM_TRY(cfm_rowchain(&q, stream));
    } else if (ffsplit) {
        // macaron feed-forward as partial slabs; then rows = x + 1/2 (sum + b2) -> x_out, LN_mha, fused q|k|v projection
        cfm_ffn_split_desc f = split_desc(2);
        f.x = x_in; f.ln_g = w->ln_ffm_g; f.ln_b = w->ln_ffm_b; f.w1 = w->ffm_w1f; f.b1 = w->ffm_b1; f.N1 = FF; f.act = CFM_ACT_SILU; f.w2 = w->ffm_w2n;
        f.psum_out = s->psum;
        CFM_TRY(cfm_ffn_split(&f, stream));
        cfm_ffn_split_desc q = split_desc(1);
        q.x = x_in; q.psum = s->psum; q.psum_b2 = w->ffm_b2; q.psum_splits = FF / 256; q.psum_alpha = 0.5f; q.rows_out = x_out;
        q.ln_g = w->ln_mha_g; q.ln_b = w->ln_mha_b; q.w1 = w->qkv_wf; q.b1 = w->qkv_b; q.N1 = 3 * D; q.act = CFM_ACT_NONE; q.out16 = s->qkv; q.ldo = 3 * D;
        ring_written = ring && dk % 4 == 0;
        if (ring_written) { q.kv_ring = io->kv_ring; q.ring_offsets = io->stream_offset; q.ring_T = io->ring_T; q.ring_H = H; q.ring_Tq = io->T; }
        CFM_TRY(cfm_ffn_split(&q, stream));
    } else if (chains && !io->macaron_done && !pair) {
        cfm_rowchain_desc m = {};
        if (merged) { m.tail_vt = s->vt; m.vt_T = io->T; m.vt_ld = s->vt_ld; }
        m.x = x_in; m.ln_g = w->ln_ffm_g; m.ln_b = w->ln_ffm_b; m.w1f = w->ffm_w1f; m.w2n = w->ffm_w2n; m.b1 = w->ffm_b1; m.b2 = w->ffm_b2;
        m.ln2_g = w->ln_mha_g; m.ln2_b = w->ln_mha_b; m.out_f32 = x_out; m.tail_w = w->qkv_wf; m.tail_b = w->qkv_b; m.tail_out = s->qkv;
        m.M = M; m.D = D; m.FF = FF; m.tail_N = 3 * D; m.tail_glu = 0; m.w_dtype = c.w_dt; m.alpha = 0.5f; m.eps = eps;
        CFM_TRY(cfm_rowchain(&m, stream));
    }
    // The fused feed-forward kernel (ffn.hip) covers LN + W1 + SiLU + W2 + residual (+ the following norms) in one launch.
    const bool fused_ffn = !chains && !c.split && w->ffm_w1f && w->ffm_w2f && w->ff_w1f && w->ff_w2f && (D == 144 || D == 256) &&
                           FF % 32 == 0 && FF <= 2048;

    // (1) macaron feed-forward: x1 = x + 1/2 W2 silu(W1 LN(x));  (2a) norm_mha
    if (chains) {
        // done above, together with the QKV projection
    } else if (fused_ffn) {
        CFM_TRY(ffn_fused(c, x_in, w->ln_ffm_g, w->ln_ffm_b, w->ffm_w1f, w->ffm_w2f, w->ffm_b1, w->ffm_b2, nullptr, nullptr,
                          w->ln_mha_g, w->ln_mha_b, x_out, s->xn));
    } else {
        if (!xn_ready) CFM_TRY(cfm_layernorm(x_in, w->ln_ffm_g, w->ln_ffm_b, nullptr, 0, nullptr, nullptr, s->xn, adt, nullptr, eps, M, D, stream));
        CFM_TRY(gemm(c, s->xn, adt, D, w->ffm_w1, w->ffm_w1_lo, w->ffm_b1, s->hid, adt, FF, M, FF, D, CFM_ACT_SILU, nullptr, 0.f, nullptr));
        CFM_TRY(gemm(c, s->hid, adt, FF, w->ffm_w2, w->ffm_w2_lo, w->ffm_b2, x_out, CFM_F32, D, M, D, FF, CFM_ACT_NONE, x_in, 0.5f, nullptr));
        CFM_TRY(cfm_layernorm(x_out, w->ln_mha_g, w->ln_mha_b, nullptr, 0, nullptr, nullptr, s->xn, adt, nullptr, eps, M, D, stream));
    }

    // (2) self-attention
    if (!chains)
        CFM_TRY(gemm(c, s->xn, adt, D, w->qkv_w, w->qkv_w_lo, w->qkv_b, s->qkv, adt, 3 * D, M, 3 * D, D, CFM_ACT_NONE, nullptr, 0.f, nullptr));
    if (has_pos && !io->pos_proj)
        CFM_TRY(gemm(c, io->pos_embed, CFM_F32, D, w->pos_w, w->pos_w_lo, nullptr, s->pos, adt, D, io->pos_rows, D, D, CFM_ACT_NONE,
                     nullptr, 0.f, nullptr));
    if (merged) {
        // attention + conv-in chain in one launch: context -> out-proj + residual -> LN_conv (pad mask) -> pointwise-conv-1 + GLU
        cfm_rowchain_desc ci = {};
        ci.att_qkv = s->qkv; ci.att_vt = s->vt; ci.att_vt_ld = s->vt_ld; ci.att_T = io->T; ci.att_H = H; ci.att_scale = 1.0f / sqrtf((float)dk);
        ci.att_mask = io->attn_mask; ci.att_m_sb = io->am_sb;
        if (has_pos) {
            const int64_t pld = io->pos_proj ? io->pos_proj_ld : D;
            ci.att_p = io->pos_proj ? io->pos_proj : s->pos; ci.att_p_sb = io->pos_shared ? 0 : pld;
            ci.att_bias_u = w->bias_u; ci.att_bias_v = w->bias_v;
        }
        ci.head_w = w->out_wf; ci.head_b = w->out_b; ci.head_res = x_out; ci.ln_g = w->ln_conv_g; ci.ln_b = w->ln_conv_b;
        ci.ln_mask = io->pad_valid; ci.out_f32 = x_out; ci.tail_w = w->pw1_wf; ci.tail_b = w->pw1_b; ci.tail_out = s->glu;
        ci.M = M; ci.D = D; ci.FF = FF; ci.tail_N = 2 * D; ci.tail_glu = 1; ci.w_dtype = c.w_dt; ci.alpha = 1.0f; ci.eps = eps;
        CFM_TRY(cfm_rowchain(&ci, stream));
    }
    const void* kq = eoff(s->qkv, D, adt);
    const void* vq = eoff(s->qkv, 2 * D, adt);
    const int64_t sb = (int64_t)io->T * 3 * D, stt = 3 * D;
    if (io->new_cache)
        CFM_TRY(cfm_kv_cache_pack(io->attn_cache, Tc, kq, vq, adt, sb, stt, sb, stt, io->new_cache, io->B, H, io->T, dk, stream));
    if (ring && !ring_written)
        CFM_TRY(cfm_kv_ring_write(kq, vq, adt, sb, stt, sb, stt, io->kv_ring, io->stream_offset, io->B, H, io->T, dk, io->ring_T, stream));
    cfm_attn_desc a = {};
    a.q = s->qkv; a.q_sb = sb; a.q_st = stt; a.q_dtype = adt;
    if (ring) {    // keys/values: every slot of the ring; the slot mask picks this step's context
        a.k = io->kv_ring; a.v = io->kv_ring + dk; a.kv_dtype = CFM_F32;
        a.k_sb = a.v_sb = (int64_t)H * Tk * 2 * dk; a.k_sh = a.v_sh = (int64_t)Tk * 2 * dk; a.k_st = a.v_st = 2 * dk;
    } else if (Tc > 0) {  // keys/values = [cache | new], already concatenated in new_cache (f32)
        a.k = io->new_cache; a.v = io->new_cache + dk; a.kv_dtype = CFM_F32;
        a.k_sb = a.v_sb = (int64_t)H * Tk * 2 * dk; a.k_sh = a.v_sh = (int64_t)Tk * 2 * dk; a.k_st = a.v_st = 2 * dk;
    } else {
        a.k = kq; a.v = vq; a.kv_dtype = adt;
        a.k_sb = a.v_sb = sb; a.k_sh = a.v_sh = dk; a.k_st = a.v_st = stt;
    }
    if (has_pos) {
        const int64_t pld = io->pos_proj ? io->pos_proj_ld : D;
        a.p = io->pos_proj ? io->pos_proj : s->pos; a.p_dtype = adt; a.p_sb = io->pos_shared ? 0 : (int64_t)P * pld; a.p_st = P == 1 ? 0 : pld;
        a.bias_u = w->bias_u; a.bias_v = w->bias_v;
    }
    a.mask = io->attn_mask; a.m_sb = io->am_sb; a.m_sq = io->am_sq;
    a.out = s->ctx; a.out_dtype = adt;
    a.B = io->B; a.H = H; a.Tq = io->T; a.Tk = Tk; a.dk = dk;
    a.mma_dtype = c.w_dt; a.split = c.split ? 1 : 0;
    a.scale = 1.0f / sqrtf((float)dk);
    if (!merged) CFM_TRY(cfm_attention(&a, stream));
    if (chains) {
        // conv-in chain: out-proj + residual -> LN_conv (pad mask) -> pointwise-conv-1 + GLU
        cfm_rowchain_desc ci = {};
        ci.head_a = s->ctx; ci.head_w = w->out_wf; ci.head_b = w->out_b; ci.head_res = x_out; ci.ln_g = w->ln_conv_g; ci.ln_b = w->ln_conv_b;
        ci.ln_mask = io->pad_valid; ci.out_f32 = x_out; ci.tail_w = w->pw1_wf; ci.tail_b = w->pw1_b; ci.tail_out = s->glu;
        ci.M = M; ci.D = D; ci.FF = FF; ci.tail_N = 2 * D; ci.tail_glu = 1; ci.w_dtype = c.w_dt; ci.alpha = 1.0f; ci.eps = eps;
        if (pair) { ci.tail_pair = 1; ci.out_f32 = s->psum + (int64_t)2 * M * D; }   // the pair's other workgroup still reads x_out: the rows go to the third slab
        // chained blocks at D = 256: the conv-in chain runs as the input stage of the next launch (depthwise + final chain + the next block's macaron chain) on
        // the tile's 32 + 14 halo rows -- no launch of its own (cfm.h cfm_rowchain_desc.cin_*)
        const bool cin = cin_merge_flag() != 0 && !merged && !pair && !ffsplit && io->next_w && io->next_x_out && D == 256 && FF == 2048 && io->ktaps == 15 && !io->causal_conv &&
                         !io->after_out;
        if (!merged && !cin) CFM_TRY(cfm_rowchain(&ci, stream));
        // the depthwise conv runs inside the final chain's input stage (15 taps); otherwise on its own
        const bool dw_fused = io->ktaps == 15 && !io->causal_conv && cfm_rowchain_dw_supported(D);
        const bool pair_dw = pair && io->ktaps == 15 && !io->causal_conv && getenv("CFM_PAIR_HEAD_GEMM") == nullptr;   // depthwise stage + pointwise-conv-2, columns over the pairs
        if (io->causal_conv) {
            CFM_TRY(cfm_dwconv_causal_bn_silu(s->glu, adt, io->conv_cache, w->dw_w, w->dw_b, w->bn_scale, w->bn_shift, s->dw, adt, io->B, io->T, D, io->ktaps, stream));
            if (io->conv_cache) CFM_TRY(cfm_conv_cache_update(s->glu, adt, io->conv_cache, io->B, io->T, D, io->ktaps, stream));
        } else if (!dw_fused && !pair_dw)
            CFM_TRY(cfm_dwconv_bn_silu(s->glu, adt, w->dw_w, w->dw_b, w->bn_scale, w->bn_shift, s->dw, adt, io->B, io->T, D, io->ktaps, stream));
        if (ffsplit) {
            // depthwise + BN + SiLU, pointwise-conv-2 + pad mask + residual (in place on x_out), then the feed-forward as partial slabs and the
            // reduce with norm_final (+ after_norm)
            if (dw_fused) {                                  // one launch: the depthwise stage as the input stage of the pointwise-conv-2 head
                cfm_rowchain_desc dh = {};
                dh.head_a = s->glu; dh.head_w = w->pw2_wf; dh.head_b = w->pw2_b; dh.head_res = x_out; dh.head_mask = io->pad_valid;
                dh.dw_w = w->dw_w; dh.dw_b = w->dw_b; dh.dw_scale = w->bn_scale; dh.dw_shift = w->bn_shift; dh.dw_T = io->T; dh.dw_K = 15;
                dh.out_f32 = x_out; dh.M = M; dh.D = D; dh.FF = FF; dh.w_dtype = c.w_dt; dh.alpha = 1.0f; dh.eps = eps;
                CFM_TRY(cfm_rowchain(&dh, stream));
            } else {
                CFM_TRY(gemm(c, s->dw, adt, D, w->pw2_w, w->pw2_w_lo, w->pw2_b, x_out, CFM_F32, D, M, D, D, CFM_ACT_NONE, x_out, 1.0f, io->pad_valid));
            }
            cfm_ffn_split_desc f = split_desc(2);
            f.x = x_out; f.ln_g = w->ln_ff_g; f.ln_b = w->ln_ff_b; f.w1 = w->ff_w1f; f.b1 = w->ff_b1; f.N1 = FF; f.act = CFM_ACT_SILU; f.w2 = w->ff_w2n;
            f.psum_out = s->psum;
            CFM_TRY(cfm_ffn_split(&f, stream));
            cfm_ffn_split_desc r = split_desc(0);
            r.x = x_out; r.psum = s->psum; r.psum_b2 = w->ff_b2; r.psum_splits = FF / 256; r.psum_alpha = 0.5f; r.ln1_g = w->ln_final_g; r.ln1_b = w->ln_final_b;
            r.rows_out = x_out;
            if (io->after_out) { r.ln2_g = io->after_g; r.ln2_b = io->after_b; r.rows2_out = io->after_out; }
            return cfm_ffn_split(&r, stream);
        }
        if (pair) {
            // pointwise-conv-2 + pad mask + residual (the conv-in chain's rows in the third slab) -> rows, parked in x_out -> LN_ff -> this workgroup's
            // half of the feed-forward; then rows + 1/2 (halves + b2) -> LN_final, in place on x_out (one workgroup per row tile)
            float* const park = s->psum + (int64_t)2 * M * D;  // the conv-in chain's rows
            cfm_rowchain_desc fa = {};
            if (pair_dw) {
                // depthwise conv + BatchNorm + SiLU as the input stage of pointwise-conv-2, each workgroup of a pair HALF of its output columns (no LayerNorm
                // behind the head in this launch, so half rows are complete results): + pad mask + residual -> x_out
                cfm_rowchain_desc dh = {};
                dh.head_a = s->glu; dh.head_w = w->pw2_wf; dh.head_b = w->pw2_b; dh.head_res = park; dh.head_mask = io->pad_valid;
                dh.dw_w = w->dw_w; dh.dw_b = w->dw_b; dh.dw_scale = w->bn_scale; dh.dw_shift = w->bn_shift; dh.dw_T = io->T; dh.dw_K = 15;
                dh.out_f32 = x_out; dh.tail_pair = 1; dh.M = M; dh.D = D; dh.FF = FF; dh.w_dtype = c.w_dt; dh.alpha = 1.0f; dh.eps = eps;
                CFM_TRY(cfm_rowchain(&dh, stream));
                fa.x = x_out;
            } else if (w->pw2_w && getenv("CFM_PAIR_HEAD_IN_CHAIN") == nullptr) {
                // the 0.5 MB head would be streamed by BOTH workgroups of every pair (+18 us per launch): it runs as a plain product over all CUs instead
                CFM_TRY(gemm(c, s->dw, adt, D, w->pw2_w, w->pw2_w_lo, w->pw2_b, x_out, CFM_F32, D, M, D, D, CFM_ACT_NONE, park, 1.0f, io->pad_valid));
                fa.x = x_out;
            } else {
                fa.head_a = s->dw; fa.head_w = w->pw2_wf; fa.head_b = w->pw2_b; fa.head_res = park; fa.head_mask = io->pad_valid; fa.out_f32 = x_out;
            }
            fa.ln_g = w->ln_ff_g; fa.ln_b = w->ln_ff_b; fa.w1f = w->ff_w1f; fa.w2n = w->ff_w2n; fa.b1 = w->ff_b1; fa.b2 = w->ff_b2;
            fa.psum_out = s->psum; fa.M = M; fa.D = D; fa.FF = FF; fa.w_dtype = c.w_dt; fa.alpha = 0.5f; fa.eps = eps;
            CFM_TRY(cfm_rowchain(&fa, stream));
            cfm_rowchain_desc fr = {};
            fr.x = x_out; fr.psum_in = s->psum; fr.psum_b2 = w->ff_b2; fr.psum_alpha = 0.5f; fr.ln_g = w->ln_final_g; fr.ln_b = w->ln_final_b;
            fr.out2_f32 = x_out; fr.M = M; fr.D = D; fr.FF = FF; fr.w_dtype = c.w_dt; fr.alpha = 1.0f; fr.eps = eps;
            CFM_TRY(cfm_rowchain(&fr, stream));
            return after_tail();
        }
        // final chain: pointwise-conv-2 + pad mask + residual -> LN_ff -> FFN -> +res -> LN_final, in place on x_out
        cfm_rowchain_desc fi = {};
        fi.head_a = dw_fused ? s->glu : s->dw; fi.head_w = w->pw2_wf;
        if (dw_fused) { fi.dw_w = w->dw_w; fi.dw_b = w->dw_b; fi.dw_scale = w->bn_scale; fi.dw_shift = w->bn_shift; fi.dw_T = io->T; fi.dw_K = 15; } fi.head_b = w->pw2_b; fi.head_res = x_out; fi.head_mask = io->pad_valid;
        fi.ln_g = w->ln_ff_g; fi.ln_b = w->ln_ff_b; fi.w1f = w->ff_w1f; fi.w2n = w->ff_w2n; fi.b1 = w->ff_b1; fi.b2 = w->ff_b2;
        fi.ln1_g = w->ln_final_g; fi.ln1_b = w->ln_final_b; fi.out_f32 = x_out;
        if (io->after_out) { fi.ln2_g = io->after_g; fi.ln2_b = io->after_b; fi.out2_f32 = io->after_out; }   // encoder.py:74 in the same launch
        fi.M = M; fi.D = D; fi.FF = FF; fi.w_dtype = c.w_dt; fi.alpha = 0.5f; fi.eps = eps;
        if (io->next_w) {
            // ... and the next block's macaron chain on the same rows, in the same launch (cfm.h cfm_layer_io.next_w)
            const cfm_layer_weights* nw = io->next_w;
            CFM_CHECK_ARG(dw_fused && !io->after_out && io->next_x_out && io->next_x_out != x_out && nw->ffm_w1f && nw->ffm_w2n && nw->qkv_wf && D == 256 && FF == 2048,
                          "encoder layer: chaining into the next block needs the fused depthwise stage, no after_out, a distinct next_x_out and the next "
                          "block's fragment-major packs (D = 256, FF = 2048)");
            fi.out_f32 = nullptr;
            fi.s2_ln_g = nw->ln_ffm_g; fi.s2_ln_b = nw->ln_ffm_b; fi.s2_w1f = nw->ffm_w1f; fi.s2_w2n = nw->ffm_w2n; fi.s2_b1 = nw->ffm_b1; fi.s2_b2 = nw->ffm_b2;
            fi.s2_out_f32 = io->next_x_out; fi.s2_alpha = 0.5f;
            fi.ln2_g = nw->ln_mha_g; fi.ln2_b = nw->ln_mha_b; fi.tail_w = nw->qkv_wf; fi.tail_b = nw->qkv_b; fi.tail_out = s->qkv; fi.tail_N = 3 * D; fi.tail_glu = 0;
            if (cin) {
                // the residual rows of the conv-in stage go to next_x_out (this tile's own rows: read back as the head's residual, overwritten at the end with
                // the next block's residual -- all by the same workgroup); halo rows are read from x_out, which this launch does not write
                fi.cin_a = s->ctx; fi.cin_w = w->out_wf; fi.cin_b = w->out_b; fi.cin_res = x_out; fi.cin_out = io->next_x_out; fi.cin_ln_g = w->ln_conv_g;
                fi.cin_ln_b = w->ln_conv_b; fi.cin_mask = io->pad_valid; fi.cin_tail_w = w->pw1_wf; fi.cin_tail_b = w->pw1_b; fi.head_res = io->next_x_out;
            }
        }
        return cfm_rowchain(&fi, stream);
    }
    CFM_TRY(gemm(c, s->ctx, adt, D, w->out_w, w->out_w_lo, w->out_b, x_out, CFM_F32, D, M, D, D, CFM_ACT_NONE, x_out, 1.0f, nullptr));

    // (3) convolution module: mask -> pw1+GLU -> depthwise+BN+SiLU -> pw2 -> mask
    CFM_TRY(cfm_layernorm(x_out, w->ln_conv_g, w->ln_conv_b, nullptr, 0, nullptr, nullptr, s->xn, adt, io->pad_valid, eps, M, D, stream));
    CFM_TRY(gemm(c, s->xn, adt, D, w->pw1_w, w->pw1_w_lo, w->pw1_b, s->glu, adt, D, M, 2 * D, D, CFM_ACT_GLU, nullptr, 0.f, nullptr));
    if (io->causal_conv) {
        CFM_TRY(cfm_dwconv_causal_bn_silu(s->glu, adt, io->conv_cache, w->dw_w, w->dw_b, w->bn_scale, w->bn_shift, s->dw, adt, io->B, io->T, D, io->ktaps, stream));
        if (io->conv_cache) CFM_TRY(cfm_conv_cache_update(s->glu, adt, io->conv_cache, io->B, io->T, D, io->ktaps, stream));
    } else
        CFM_TRY(cfm_dwconv_bn_silu(s->glu, adt, w->dw_w, w->dw_b, w->bn_scale, w->bn_shift, s->dw, adt, io->B, io->T, D, io->ktaps, stream));
    CFM_TRY(gemm(c, s->dw, adt, D, w->pw2_w, w->pw2_w_lo, w->pw2_b, x_out, CFM_F32, D, M, D, D, CFM_ACT_NONE, x_out, 1.0f, io->pad_valid));

    // (4) feed-forward + (5) norm_final: in place on x_out (a workgroup reads its 32 rows completely before writing them)
    if (fused_ffn) {
        CFM_TRY(ffn_fused(c, x_out, w->ln_ff_g, w->ln_ff_b, w->ff_w1f, w->ff_w2f, w->ff_b1, w->ff_b2, w->ln_final_g, w->ln_final_b,
                          nullptr, nullptr, x_out, nullptr));
        return after_tail();
    }
    CFM_TRY(cfm_layernorm(x_out, w->ln_ff_g, w->ln_ff_b, nullptr, 0, nullptr, nullptr, s->xn, adt, nullptr, eps, M, D, stream));
    CFM_TRY(gemm(c, s->xn, adt, D, w->ff_w1, w->ff_w1_lo, w->ff_b1, s->hid, adt, FF, M, FF, D, CFM_ACT_SILU, nullptr, 0.f, nullptr));
    CFM_TRY(gemm(c, s->hid, adt, FF, w->ff_w2, w->ff_w2_lo, w->ff_b2, x_out, CFM_F32, D, M, D, FF, CFM_ACT_NONE, x_out, 0.5f, nullptr));

    // (5) norm_final in place (+ the next block's first norm chained in registers)
    if (next_g)
        CFM_TRY(cfm_layernorm(x_out, w->ln_final_g, w->ln_final_b, x_out, CFM_F32, next_g, next_b, s->xn, adt, nullptr, eps, M, D, stream));
    else
        CFM_TRY(cfm_layernorm(x_out, w->ln_final_g, w->ln_final_b, x_out, CFM_F32, nullptr, nullptr, nullptr, 0, nullptr, eps, M, D, stream));
    return after_tail();
}
