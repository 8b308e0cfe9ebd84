"""Train-mode forward + backward of the hot path as ``torch.autograd.Function``s over the C ABI (BASELINE config 3).

Each Function's ``forward`` runs the train-mode kernels and keeps what the backward needs; its ``backward`` RETURNS the gradient of
every ``nn.Parameter`` it was given (so ``.grad`` accumulation, DDP reducer hooks and gradient accumulation work as with any torch
op -- SURVEY 8b "Threading").  Nothing here computes: the arithmetic is cfm.gemm (input gradients, on transposed weight packs, with
the activation-derivative epilogues), cfm.gemm_tn (weight / bias gradients), cfm.layernorm_bwd, cfm.attention_bwd,
cfm.dwconv_bn_train(_bwd), cfm.glu_bwd, cfm.col2im_relu_bwd, cfm.conv1_wgrad, cfm.ctc_nll_train / cfm.ctc_grad.

What train mode means here (reference: encoder_layer.py:49-71, convolution.py:34-49, decoder.py:18-23 under module.train()):
  * BatchNorm1d uses BATCH statistics over all B*T' positions, padded frames included (quirk Q6), and updates its running statistics;
  * every nn.Dropout is active.  Dropout is applied by the kernels from a counter-based generator (no bit-parity with torch's RNG, as
    SURVEY 2.2 notes): parity is defined and tested at p = 0;
  * the batch path's positional score term is constant along each softmax row, so linear_pos / pos_bias_v receive an exact
    zero gradient (the reference's is rounding noise ~1e-8) and the term is not evaluated; pos_bias_u rides in linear_q's bias.
"""
import ctypes
import os

import torch

import cfm
from cfm import packing


def _gemm(*args, **kw):
    """cfm.gemm with the training tile choice (cfm.h CFM_TILE_AUTO_TRAIN: K-group tiles allowed), as csrc/train_layer.cpp launches its products."""
    kw.setdefault("tile", cfm.TILE_AUTO_TRAIN)
    return cfm.gemm(*args, **kw)


def _f32c(t):
    return (t if t.dtype == torch.float32 else t.float()).contiguous()


def draw_seed():
    """A fresh 31-bit base seed for one Function call's dropout masks, from torch's CPU generator (torch.manual_seed makes a run
    reproducible; no device round trip).  Sites add small odd offsets."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def _drop(p, seed, site):
    return (float(p), (seed + 0x9E3779B1 * site) & 0xFFFFFFFF) if p and p > 0.0 else None


# ======================================================================================================================
# raw forward / backward pieces (no autograd inside): tensors in, tensors + saved state out
# ======================================================================================================================
def ffn_fwd(pk, x, ln, prec, alpha, act=cfm.ACT_SILU, drop_h=None, drop_o=None):
    """x f32 [M,D] -> (x + alpha * drop_o(FFN(LN(x))), saved).  ln = (gain, bias) or None (no norm, no residual: the bare module).
    drop_h: dropout on the hidden activation (feedforward.py:19); drop_o: on the branch output (encoder_layer.py:58,69); (p, seed) | None."""
    adt = prec.act_dtype
    M = x.shape[0]
    FF = pk.w1.shape[0]
    xn = cfm.layernorm(x, ln[0], ln[1], out1_dtype=adt)[0] if ln is not None else (x if x.dtype == adt else cfm.cast(x, adt))
    z = torch.empty((M, FF), dtype=adt, device=x.device)
    h = _gemm(xn, pk.w1, bias=pk.b1, w_lo=pk.w1_lo, act=act, out_dtype=adt, pre_out=z, drop=drop_h)
    if ln is not None:
        y = _gemm(h, pk.w2, bias=pk.b2, w_lo=pk.w2_lo, residual=x, alpha=alpha, drop=drop_o)
    else:
        y = _gemm(h, pk.w2, bias=pk.b2, w_lo=pk.w2_lo, out_dtype=torch.float32, drop=drop_o)
    return y, (x, xn, z, h)


def ffn_bwd(pk, saved, dy, ln, prec, alpha, act=cfm.ACT_SILU, drop_h=None, drop_o=None):
    """dy f32 [M,D] = d loss / d output -> (dx, grads).  With ln: dx = dy + dLN(...) computed in place over dy."""
    x, xn, z, h = saved
    mma, sp = prec.w_code, prec.split
    dact = cfm.ACT_DSILU if act == cfm.ACT_SILU else cfm.ACT_DRELU
    dyb = dy
    if drop_o is not None:                                          # the branch gradient through the output dropout, as a GEMM operand
        dyb, alpha = cfm.dropout_rows(dy, prec.act_dtype, alpha=alpha, drop=drop_o), 1.0
    if drop_h is not None and dact != cfm.ACT_DSILU:
        raise NotImplementedError("hidden dropout with a ReLU feed-forward in train mode")
    dW2, db2 = cfm.gemm_tn(dyb, h, want_colsum=True, alpha=alpha, mma_code=mma, split=sp)
    dz = _gemm(dyb, pk.w2t, w_lo=pk.w2t_lo, act=dact, aux=z, alpha=alpha, out_dtype=prec.act_dtype, drop=drop_h)
    dW1, db1 = cfm.gemm_tn(dz, xn, want_colsum=True, mma_code=mma, split=sp)
    dxn = _gemm(dz, pk.w1t, w_lo=pk.w1t_lo, out_dtype=torch.float32)
    grads = {"w_1.weight": dW1, "w_1.bias": db1, "w_2.weight": dW2, "w_2.bias": db2}
    if ln is None:
        return dxn, grads, None
    dx, dg, db = cfm.layernorm_bwd(x, dxn, ln[0], dres=dy, dx=dy)
    return dx, grads, (dg, db)


def mhsa_fwd(mod, pk, x, ln, B, T, mask8, m_str, prec, relative, drop_a=None, drop_o=None, drop_o2=None):
    """x f32 [B*T,D] -> (x + drop_o(MHSA(LN(x))) (ln given) or MHSA(x), saved).  Batch path: no cache, positional term not evaluated.
    drop_a: dropout on the probabilities (attention.py:93); drop_o / drop_o2: on the projected output (encoder_layer.py:61; the plain
    MHSA's own dropout after linear_out, attention.py:177)."""
    adt = prec.act_dtype
    M, D = x.shape
    H, dk = mod.num_heads, mod.d_k
    xn = cfm.layernorm(x, ln[0], ln[1], out1_dtype=adt)[0] if ln is not None else (x if x.dtype == adt else cfm.cast(x, adt))
    qkv = _gemm(xn, pk.qkv_w, bias=pk.qkv_b, w_lo=pk.qkv_w_lo, out_dtype=adt)
    ctx = torch.empty((M, D), dtype=adt, device=x.device)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=x.device)
    st = (T * 3 * D, 3 * D)
    cfm.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], B, H, T, T, dk, st, st + (dk,), st + (dk,), ctx, mask=mask8, mask_str=m_str, mma_code=prec.w_code,
                  split=prec.split, lse=lse, drop=drop_a)
    if drop_o is None and drop_o2 is not None:
        drop_o, drop_o2 = drop_o2, None
    if ln is not None:
        y = _gemm(ctx, pk.out_w, bias=pk.out_b, w_lo=pk.out_w_lo, residual=x, alpha=1.0, drop=drop_o, drop2=drop_o2)
    else:
        y = _gemm(ctx, pk.out_w, bias=pk.out_b, w_lo=pk.out_w_lo, out_dtype=torch.float32, drop=drop_o, drop2=drop_o2)
    return y, (x, xn, qkv, ctx, lse)


def mhsa_bwd(mod, pk, saved, dy, ln, B, T, mask8, m_str, prec, relative, drop_a=None, drop_o=None, drop_o2=None):
    x, xn, qkv, ctx, lse = saved
    M, D = x.shape
    H, dk = mod.num_heads, mod.d_k
    mma, sp = prec.w_code, prec.split
    if drop_o is None and drop_o2 is not None:
        drop_o, drop_o2 = drop_o2, None
    dyb = dy if drop_o is None else cfm.dropout_rows(dy, prec.act_dtype, drop=drop_o, drop2=drop_o2)
    dWo, dbo = cfm.gemm_tn(dyb, ctx, want_colsum=True, mma_code=mma, split=sp)
    dctx = _gemm(dyb, pk.out_t, w_lo=pk.out_t_lo, out_dtype=prec.act_dtype)
    dqkv = torch.empty_like(qkv)
    st = (T * 3 * D, 3 * D)
    cfm.attention_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], ctx, dctx, lse, B, H, T, T, dk, st, st, st, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], mask=mask8,
                      mask_str=m_str, mma_code=mma, split=sp, drop=drop_a)
    dWqkv, dbqkv = cfm.gemm_tn(dqkv, xn, want_colsum=True, mma_code=mma, split=sp)
    dxn = _gemm(dqkv, pk.qkv_t, w_lo=pk.qkv_t_lo, out_dtype=torch.float32)
    grads = {"linear_q.weight": dWqkv[:D], "linear_k.weight": dWqkv[D:2 * D], "linear_v.weight": dWqkv[2 * D:],
             "linear_q.bias": dbqkv[:D], "linear_k.bias": dbqkv[D:2 * D], "linear_v.bias": dbqkv[2 * D:],
             "linear_out.weight": dWo, "linear_out.bias": dbo}
    if relative:
        grads["pos_bias_u"] = dbqkv[:D].clone().view(H, dk)       # d/du of (q + u) . k^T = the column sums of dq (its own tensor: autograd
                                                                  # accumulates into what it is handed, and linear_q.bias holds the other copy)
        # softmax-invariant in the batch path (one positional row per utterance broadcast over keys, attention.py:20,84-88)
        grads["pos_bias_v"] = torch.zeros_like(mod.pos_bias_v)
        grads["linear_pos.weight"] = torch.zeros_like(mod.linear_pos.weight)
    if ln is None:
        return dxn, grads, None
    dx, dg, db = cfm.layernorm_bwd(x, dxn, ln[0], dres=dy, dx=dy)
    return dx, grads, (dg, db)


def conv_module_fwd(mod, pk, x, ln, B, T, keep, prec, drop_o=None):
    """x f32 [B*T,D] -> (x + ConvModule(mask(LN(x))) (ln given) or ConvModule(mask(x)), saved); BatchNorm in training mode."""
    adt = prec.act_dtype
    M, D = x.shape
    if ln is not None:
        if keep is not None:
            xn = cfm.layernorm(x, ln[0], ln[1], want1=False, out2_dtype=adt, row_mask=keep)[1]
        else:
            xn = cfm.layernorm(x, ln[0], ln[1], out1_dtype=adt)[0]
    else:
        xn = x if x.dtype == adt else cfm.cast(x, adt)
    u = torch.empty((M, 2 * D), dtype=adt, device=x.device)
    # without a LayerNorm in front the padded INPUT rows are zeroed in the GEMM (mask_mode 1: acc = 0, bias kept -- quirk Q5)
    glu = _gemm(xn, pk.pw1_w, bias=pk.pw1_b, w_lo=pk.pw1_w_lo, act=cfm.ACT_GLU, out_dtype=adt, pre_out=u,
                   row_mask=keep if ln is None else None, mask_mode=1)
    bn = mod.norm
    momentum = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    c, stats, s = cfm.dwconv_bn_train(glu.view(B, T, D), pk.dw_w, pk.dw_b, pk.gamma, pk.beta, rm, rv, momentum, bn.eps, adt)
    if bn.track_running_stats:
        bn.num_batches_tracked += 1
    if ln is not None:
        y = _gemm(s.view(M, D), pk.pw2_w, bias=pk.pw2_b, w_lo=pk.pw2_w_lo, row_mask=keep, mask_mode=0, residual=x, alpha=1.0, drop=drop_o)
    else:
        y = _gemm(s.view(M, D), pk.pw2_w, bias=pk.pw2_b, w_lo=pk.pw2_w_lo, row_mask=keep, mask_mode=0, out_dtype=torch.float32)
    return y, (x, xn, u, glu, c, stats, s)


def conv_module_bwd(mod, pk, saved, dy, ln, B, T, keep, prec, drop_o=None):
    x, xn, u, glu, c, stats, s = saved
    M, D = x.shape
    adt, mma, sp = prec.act_dtype, prec.w_code, prec.split
    # masked OUTPUT rows have no gradient: with the branch gradient materialised (dropout), its padded rows are zeroed there and the two products
    # run unmasked (the 16-bit weight-gradient kernel without a row mask) -- as csrc/train_layer.cpp does
    dyb = dy if drop_o is None else cfm.dropout_rows(dy, adt, drop=drop_o, row_mask=keep)
    km = keep if drop_o is None else None
    dW2, db2 = cfm.gemm_tn(dyb, s.view(M, D), want_colsum=True, row_mask=km, mma_code=mma, split=sp)
    ds = _gemm(dyb, pk.pw2_t, w_lo=pk.pw2_t_lo, row_mask=km, mask_mode=1 if km is not None else 0, out_dtype=adt)
    dglu, ddw_w, ddw_b, dgamma, dbeta = cfm.dwconv_bn_train_bwd(ds.view(B, T, D), c, stats, glu.view(B, T, D), pk.dw_w, adt)
    du = cfm.glu_bwd(u, dglu.view(M, D), adt)
    # pointwise-conv-1 saw zeroed padded rows: xn already is (LayerNorm path) or is masked here (bare module)
    dW1i, db1i = cfm.gemm_tn(du, xn, want_colsum=True, mma_code=mma, split=sp) if ln is not None or keep is None else _pw1_wgrad_masked(du, xn, keep, mma, sp)
    dxn = _gemm(du, pk.pw1_t, w_lo=pk.pw1_t_lo, out_dtype=torch.float32)
    dW1 = torch.empty_like(dW1i)
    dW1[pk.idx] = dW1i                                            # undo the value / gate row interleave of the pack
    db1 = torch.empty_like(db1i)
    db1[pk.idx] = db1i
    K = pk.dw_w.shape[1]
    grads = {"pointwise_conv1.weight": dW1.view(2 * D, D, 1), "pointwise_conv1.bias": db1, "depthwise_conv.weight": ddw_w.view(D, 1, K),
             "depthwise_conv.bias": ddw_b, "norm.weight": dgamma, "norm.bias": dbeta, "pointwise_conv2.weight": dW2.view(D, D, 1),
             "pointwise_conv2.bias": db2}
    if ln is None:
        if keep is not None:
            dxn = dxn * keep.view(M, 1).to(dxn.dtype)             # bare module: d masked_fill (convolution.py:36-37)
        return dxn, grads, None
    dx, dg, db = cfm.layernorm_bwd(x, dxn, ln[0], row_mask=keep, dres=dy, dx=dy)
    return dx, grads, (dg, db)


def _pw1_wgrad_masked(du, xn, keep, mma, sp):
    """bare ConvolutionModule with a pad mask: the weight gradient sees the masked input rows (bias gradient: all rows)."""
    M = du.shape[0]
    xm = xn * keep.view(M, 1).to(xn.dtype)
    return cfm.gemm_tn(du, xm, want_colsum=True, mma_code=mma, split=sp)


def subsampling_fwd(pk, x, cmvn, prec):
    """x f32 [B,T,F] -> (y f32 [B*T2, D], saved)."""
    adt = prec.act_dtype
    B, T, F = x.shape
    C = pk.C
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    h1 = cfm.conv1_relu(x, pk.w1, pk.b1, adt, cmvn=cmvn, mma=cfm.conv1_relu_mma_supported(C, adt))
    h2 = _gemm(h1, pk.w2, bias=pk.b2, w_lo=pk.w2_lo, act=cfm.ACT_RELU, conv=(C, T1, F1, T2, F2, B * T2 * F2), out_dtype=adt)
    y = _gemm(h2.view(B * T2, F2 * C), pk.wl, bias=pk.bl, w_lo=pk.wl_lo, out_dtype=torch.float32)
    return y, (x, h1, h2, (B, T, F, C, T1, F1, T2, F2), cmvn)


def subsampling_bwd(pk, saved, dy, prec):
    x, h1, h2, (B, T, F, C, T1, F1, T2, F2), cmvn = saved
    adt, mma, sp = prec.act_dtype, prec.w_code, prec.split
    Dout = pk.wl.shape[0]
    h2f = h2.view(B * T2, F2 * C)
    dWl, dbl = cfm.gemm_tn(dy, h2f, want_colsum=True, mma_code=mma, split=sp)
    dh2 = _gemm(dy, pk.wlt, w_lo=pk.wlt_lo, act=cfm.ACT_DRELU, aux=h2f, alpha=1.0, out_dtype=adt).view(B * T2 * F2, C)
    dW2, db2 = cfm.gemm_tn(dh2, h1, conv=(C, T1, F1, T2, F2), want_colsum=True, mma_code=mma, split=sp)
    dcol = _gemm(dh2, pk.w2t, w_lo=pk.w2t_lo, out_dtype=adt)
    dh1 = cfm.col2im_relu_bwd(dcol, h1, adt)
    dw1, db1 = cfm.conv1_wgrad(dh1, x, cmvn=cmvn)
    return {"conv.0.weight": dw1.t().reshape(C, 1, 3, 3), "conv.0.bias": db1,
            "conv.2.weight": dW2.view(C, 3, 3, C).permute(0, 3, 1, 2), "conv.2.bias": db2,
            "out.0.weight": dWl.view(Dout, F2, C).permute(0, 2, 1).reshape(Dout, C * F2), "out.0.bias": dbl}


# ======================================================================================================================
# autograd Functions
# ======================================================================================================================
def _params(mod):
    names, tensors = [], []
    for n, p in mod.named_parameters():
        names.append(n)
        tensors.append(p)
    return names, tensors


def _ordered(names, grads, tensors):
    out = []
    for n, p in zip(names, tensors):
        g = grads.get(n)
        if g is None:
            raise RuntimeError("no gradient was produced for parameter %s" % n)
        out.append(g.reshape(p.shape).to(p.dtype) if p.requires_grad else None)
    return out


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm on f32 rows (encoder.py:74 after_norm)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x2 = _f32c(x.reshape(-1, x.shape[-1]))
        y = cfm.layernorm(x2, weight.detach(), bias.detach(), eps=eps)[0]
        ctx.save_for_backward(x2, weight)
        ctx.eps, ctx.shape = eps, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dx, dg, db = cfm.layernorm_bwd(x2, _f32c(dy.reshape(x2.shape)), weight.detach(), eps=ctx.eps)
        return dx.view(ctx.shape), dg, db, None


class FeedForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, prec, act, *params):
        pk = packing.pack_ffn_train(mod, prec)
        x2 = _f32c(x.reshape(-1, x.shape[-1]))
        ctx.drop_h = _drop(mod.dropout.p, draw_seed(), 1) if mod.dropout.p > 0 else None
        y, saved = ffn_fwd(pk, x2, None, prec, 1.0, act, drop_h=ctx.drop_h)
        ctx.mod, ctx.prec, ctx.pk, ctx.saved, ctx.act, ctx.shape = mod, prec, pk, saved, act, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        dx, grads, _ = ffn_bwd(ctx.pk, ctx.saved, _f32c(dy.reshape(-1, dy.shape[-1])), None, ctx.prec, 1.0, ctx.act, drop_h=ctx.drop_h)
        names, tensors = _params(ctx.mod)
        return (dx.view(ctx.shape), None, None, None) + tuple(_ordered(names, grads, tensors))


class AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, prec, relative, mask8, m_str, *params):
        pk = packing.pack_mhsa_train(mod, prec, relative)
        B, T, D = x.shape
        p = mod.dropout.p
        seed = draw_seed() if p > 0 else 0
        drops = (_drop(p, seed, 1), None, None if relative else _drop(p, seed, 2))      # probabilities; the plain MHSA also drops its output
        y, saved = mhsa_fwd(mod, pk, _f32c(x.reshape(B * T, D)), None, B, T, mask8, m_str, prec, relative, *drops)
        ctx.args = (mod, prec, relative, mask8, m_str, pk, saved, B, T, D, drops)
        return y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        mod, prec, relative, mask8, m_str, pk, saved, B, T, D, drops = ctx.args
        dx, grads, _ = mhsa_bwd(mod, pk, saved, _f32c(dy.reshape(B * T, D)), None, B, T, mask8, m_str, prec, relative, *drops)
        names, tensors = _params(mod)
        return (dx.view(B, T, D), None, None, None, None, None) + tuple(_ordered(names, grads, tensors))


class ConvModuleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, prec, keep, *params):
        pk = packing.pack_conv_module_train(mod, prec)
        B, T, D = x.shape
        y, saved = conv_module_fwd(mod, pk, _f32c(x.reshape(B * T, D)), None, B, T, keep, prec)
        ctx.args = (mod, prec, keep, pk, saved, B, T, D)
        return y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        mod, prec, keep, pk, saved, B, T, D = ctx.args
        dx, grads, _ = conv_module_bwd(mod, pk, saved, _f32c(dy.reshape(B * T, D)), None, B, T, keep, prec)
        names, tensors = _params(mod)
        return (dx.view(B, T, D), None, None, None) + tuple(_ordered(names, grads, tensors))


class SubsamplingFn(torch.autograd.Function):
    """conv.0 + ReLU + conv.2 + ReLU + out.0 of the front-end (convolution.py:70-74); the fbank input gets no gradient."""

    @staticmethod
    def forward(ctx, x, mod, prec, cmvn, *params):
        pk = packing.pack_subsampling_train(mod, prec)
        y, saved = subsampling_fwd(pk, _f32c(x), cmvn, prec)
        ctx.args = (mod, prec, pk, saved)
        B, T2 = x.shape[0], saved[3][6]
        return y.view(B, T2, -1)

    @staticmethod
    def backward(ctx, dy):
        mod, prec, pk, saved = ctx.args
        grads = subsampling_bwd(pk, saved, _f32c(dy.reshape(-1, dy.shape[-1])), prec)
        names, tensors = [], []
        for n, p in mod.named_parameters():
            if n.startswith("conv.") or n.startswith("out."):
                names.append(n)
                tensors.append(p)
        return (None, None, None, None) + tuple(_ordered(names, grads, tensors))


def subsampling_params(mod):
    return [p for n, p in mod.named_parameters() if n.startswith("conv.") or n.startswith("out.")]


# ----------------------------------------------------------------------------------------------------------------------
# the block as TWO host calls (csrc/train_layer.cpp): the same launches as the op-by-op composition above, enqueued from C++
# ----------------------------------------------------------------------------------------------------------------------
USE_COMPOSITE = True          # False: EncoderLayerFn runs the op-by-op composition (tests compare the two)

_GRAD_FIELDS = {"norm_ff_macaron.weight": "ln_ffm_g", "norm_ff_macaron.bias": "ln_ffm_b", "norm_mha.weight": "ln_mha_g", "norm_mha.bias": "ln_mha_b",
                "norm_conv.weight": "ln_conv_g", "norm_conv.bias": "ln_conv_b", "norm_ff.weight": "ln_ff_g", "norm_ff.bias": "ln_ff_b",
                "norm_final.weight": "ln_final_g", "norm_final.bias": "ln_final_b",
                "feed_forward_macaron.w_1.weight": "ffm_w1", "feed_forward_macaron.w_1.bias": "ffm_b1", "feed_forward_macaron.w_2.weight": "ffm_w2",
                "feed_forward_macaron.w_2.bias": "ffm_b2", "feed_forward.w_1.weight": "ff_w1", "feed_forward.w_1.bias": "ff_b1",
                "feed_forward.w_2.weight": "ff_w2", "feed_forward.w_2.bias": "ff_b2", "self_attn.linear_out.weight": "out_w", "self_attn.linear_out.bias": "out_b",
                "conv_module.pointwise_conv2.weight": "pw2_w", "conv_module.pointwise_conv2.bias": "pw2_b", "conv_module.depthwise_conv.weight": "dw_w",
                "conv_module.depthwise_conv.bias": "dw_b", "conv_module.norm.weight": "bn_g", "conv_module.norm.bias": "bn_b", "self_attn.pos_bias_u": "pos_bias_u"}


def layer_grad_layout(layer, offsets=None):
    """Where each parameter's gradient lives in the block's flat gradient slab: {name: (offset, numel)} in floats (16-byte aligned), the slab
    length, and the device row-offset maps of the two fused products.  offsets: a ready-made {name: offset} (the data-parallel trainer's
    bucket layout) or None for a private slab in named_parameters order.  Cached on the layer per layout."""
    key = None if offsets is None else tuple(sorted(offsets.items()))
    cache = layer.__dict__.setdefault("_grad_layout", {})
    hit = cache.get(key)
    dev = next(layer.parameters()).device
    if hit is not None and hit["device"] == dev:
        return hit
    lay, n = {}, 0
    for name, p in layer.named_parameters():
        if offsets is None:
            lay[name] = (n, p.numel())
            n += (p.numel() + 3) // 4 * 4
        else:
            lay[name] = (offsets[name], p.numel())
            n = max(n, offsets[name] + p.numel())
    D = layer.encoder_dim
    ar = torch.arange(D, dtype=torch.int64)
    qo, ko, vo = (lay["self_attn.linear_%s.weight" % c][0] for c in "qkv")
    qb, kb, vb = (lay["self_attn.linear_%s.bias" % c][0] for c in "qkv")
    qkv_row = torch.cat([qo + ar * D, ko + ar * D, vo + ar * D])
    qkv_bias = torch.cat([qb + ar, kb + ar, vb + ar])
    idx = packing.glu_interleave_index(D, torch.device("cpu"))            # GEMM row j of the interleaved pack = reference row idx[j]
    pw1_row = lay["conv_module.pointwise_conv1.weight"][0] + idx * D
    pw1_bias = lay["conv_module.pointwise_conv1.bias"][0] + idx
    hit = dict(device=dev, layout=lay, numel=n, qkv_row=qkv_row.to(dev), qkv_bias=qkv_bias.to(dev), pw1_row=pw1_row.to(dev), pw1_bias=pw1_bias.to(dev))
    cache[key] = hit
    return hit


def _train_weights_struct(layer, pks):
    hit = layer.__dict__.get("_train_w_struct")
    if hit is not None and hit[0] is pks:                   # the same pack objects (packing.pack_layer_train returns them until a parameter changes)
        bn = layer.conv_module.norm
        if bn.momentum is None:                             # cumulative moving average: the factor follows the batch count
            hit[1].bn_momentum = 1.0 / float(int(bn.num_batches_tracked) + 1)
        return hit[1]
    w = _build_train_weights_struct(layer, pks)
    layer.__dict__["_train_w_struct"] = (pks, w)
    return w


def _build_train_weights_struct(layer, pks):
    ffm, att, cv, ff = pks
    w = cfm.LayerTrainWeights()
    for short, name in (("ffm", "norm_ff_macaron"), ("mha", "norm_mha"), ("conv", "norm_conv"), ("ff", "norm_ff"), ("final", "norm_final")):
        ln = getattr(layer, name)
        setattr(w, "ln_%s_g" % short, ln.weight.data_ptr())
        setattr(w, "ln_%s_b" % short, ln.bias.data_ptr())
    for pre, pk in (("ffm", ffm), ("ff", ff)):
        for f in ("w1", "w1_lo", "w2", "w2_lo", "w1t", "w1t_lo", "w2t", "w2t_lo", "b1", "b2"):
            setattr(w, pre + "_" + f, cfm.ptr(getattr(pk, f)))
        for f in ("w1f", "w2f"):                            # fragment-major packs (the stack pack builds them): the one-launch feed-forward forward
            setattr(w, pre + "_" + f, cfm.ptr(getattr(pk, f, None)))
    for f in ("qkv_w", "qkv_w_lo", "qkv_t", "qkv_t_lo", "out_w", "out_w_lo", "out_t", "out_t_lo", "qkv_b", "out_b"):
        setattr(w, f, cfm.ptr(getattr(att, f)))
    for f in ("pw1_w", "pw1_w_lo", "pw1_t", "pw1_t_lo", "pw2_w", "pw2_w_lo", "pw2_t", "pw2_t_lo", "pw1_b", "pw2_b", "dw_w", "dw_b"):
        setattr(w, f, cfm.ptr(getattr(cv, f)))
    w.bn_gamma, w.bn_beta = cv.gamma.data_ptr(), cv.beta.data_ptr()
    bn = layer.conv_module.norm
    if bn.track_running_stats:
        w.bn_running_mean, w.bn_running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    w.bn_momentum = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
    w.bn_eps = bn.eps
    return w


DIRECT_GRADS = os.environ.get("CFM_DIRECT_GRADS", "0") != "0"     # flat-leaf blocks add their gradients straight into the trainer's gradient buffer: measured
# SLOWER at config 3 (17.8 vs 17.1 ms per step: the atomics / read-add-stores then hit a cold 139 MB buffer instead of a just-zeroed slab in cache) -- opt-in
USE_PACK_KERNEL = os.environ.get("CFM_PACK_KERNEL", "1") != "0"   # one cfm_pack_matrices launch per block and step instead of ~25 torch ops
WGRAD_BESIDE = os.environ.get("CFM_WGRAD_BESIDE", "0") != "0"     # stack path: each block's grouped weight-gradient launch on a side stream
OVERLAP_WGRAD = os.environ.get("CFM_OVERLAP_WGRAD", "0") != "0"   # measured at config 3: 21.3 ms per step with, 19.7 ms without
_SIDE = {}


def _side_stream(dev):
    """The per-device stream the composite backward issues its weight-gradient products on (cfm.h: cfm_layer_train_io.side_stream)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=dev)
    return st


def _composite_ok(layer, x, flat=False):
    """flat: the block's parameters are views of the trainer's flat f32 leaf (trainer.py) -- contiguous float32 by construction, so the walk over
    the module tree (~0.1 ms of host time per call, and the training step is host-bound) is skipped."""
    bn = layer.conv_module.norm
    return (USE_COMPOSITE and layer.kernel_size == 15 and bn.weight is not None and bn.bias is not None and
            layer.conv_module.pointwise_conv1.bias is not None and layer.conv_module.depthwise_conv.bias is not None and
            (flat or all(p.dtype == torch.float32 and p.is_contiguous() for p in layer.parameters())))


def _layer_composite_forward(ctx, x0, layer, prec, mask8, m_str, keep, pks, B, T, D):
    dev, adt = x0.device, prec.act_dtype
    M, FF, H = B * T, layer.hidden_dim, layer.num_heads
    esz = 4 if adt == torch.float32 else 2
    # everything the backward needs, in two allocations: act-dtype rows and f32 rows
    widths = dict(xn1=D, z1=FF, h1=FF, xn2=D, qkv=3 * D, ctx=D, xn3=D, u=2 * D, glu=D, s=D, xn4=D, z2=FF, h2=FF)
    act = torch.empty((M * sum(widths.values()),), dtype=adt, device=dev)
    f32 = torch.empty((5 * M * D + B * H * T + 4 * D,), dtype=torch.float32, device=dev)
    sv = cfm.LayerTrainSaved()
    o = 0
    for name, wd in widths.items():
        setattr(sv, name, act.data_ptr() + o * esz)
        o += M * wd
    o = 0
    for name, n in (("x1", M * D), ("x2", M * D), ("x3", M * D), ("x4", M * D), ("c", M * D), ("lse", B * H * T), ("stats", 4 * D)):
        setattr(sv, name, f32.data_ptr() + o * 4)
        o += n
    sc = cfm.LayerTrainScratch()
    sc.dwbn_ws = cfm.scratch("dwbn", cfm.lib().cfm_dwconv_bn_ws(B, T, D), torch.float32, dev).data_ptr()
    io = cfm.LayerTrainIO()
    io.B, io.T, io.D, io.H, io.FF, io.ktaps, io.act_dtype, io.w_dtype = B, T, D, H, FF, layer.kernel_size, prec.act_code, prec.w_code
    io.attn_mask, io.am_sb, io.am_sq, io.pad_valid = cfm.ptr(mask8), m_str[0], m_str[1], cfm.ptr(keep)
    p_br, p_a = layer.dropout.p, layer.self_attn.dropout.p
    io.p_hidden_m, io.p_hidden, io.p_branch, io.p_attn = layer.feed_forward_macaron.dropout.p, layer.feed_forward.dropout.p, p_br, p_a
    io.p_attn_out = 0.0 if layer.use_relative else p_a
    io.seed = draw_seed() if max(io.p_hidden_m, io.p_hidden, p_br, p_a) > 0 else 0
    io.deterministic = 1 if cfm.ops._deterministic[0] else 0
    w = _train_weights_struct(layer, pks)
    y = torch.empty((M, D), dtype=torch.float32, device=dev)
    cfm.check(cfm.lib().cfm_encoder_layer_train_forward(ctypes.byref(w), ctypes.byref(io), ctypes.byref(sv), ctypes.byref(sc), x0.data_ptr(), y.data_ptr(),
                                                        cfm.stream()), "cfm_encoder_layer_train_forward")
    bn = layer.conv_module.norm
    if bn.track_running_stats:
        bn.num_batches_tracked += 1
    ctx.comp = (layer, prec, mask8, keep, pks, w, io, sv, (act, f32, x0), B, T, D)
    return y


def _layer_composite_backward(ctx, dy):
    layer, prec, mask8, keep, pks, w, io, sv, (act, f32, x0), B, T, D = ctx.comp
    dev, adt = x0.device, prec.act_dtype
    M, FF, H = B * T, layer.hidden_dim, layer.num_heads
    # the slab mirrors the trainer's flat layout of this block when the block takes part as one autograd leaf (trainer.py), else a private
    # named_parameters-order layout; zero-filled: weight / bias gradients are accumulated into it by the split-M products
    leaf = layer.__dict__.get("_flat_leaf") if ctx.flat else None
    lay = layer_grad_layout(layer, layer.__dict__.get("_flat_grad_offsets") if ctx.flat else None)
    # A block registered with a gradient SINK (trainer.py: its range of the step's flat gradient buffer + a "ready" callback) has its gradients
    # added straight into that buffer -- no zero-filled slab, no AccumulateGrad add; needs the accumulating (atomic) LayerNorm sums
    sink = layer.__dict__.get("_flat_grad_sink") if (ctx.flat and DIRECT_GRADS and not cfm.ops._deterministic[0]) else None
    if sink is not None and (sink[0].numel() != leaf.numel() or sink[0].device != dev):
        sink = None
    io.grads_accumulate = 1 if sink is not None else 0
    slab = sink[0] if sink is not None else torch.zeros((leaf.numel() if leaf is not None else lay["numel"],), dtype=torch.float32, device=dev)
    g = cfm.LayerTrainGrads()
    g.slab = slab.data_ptr()
    for name, field in _GRAD_FIELDS.items():
        if name in lay["layout"]:
            setattr(g, field, slab.data_ptr() + 4 * lay["layout"][name][0])
    g.q_bias = slab.data_ptr() + 4 * lay["layout"]["self_attn.linear_q.bias"][0]
    g.qkv_row_off, g.qkv_bias_off, g.pw1_row_off, g.pw1_bias_off = (lay[k].data_ptr() for k in ("qkv_row", "qkv_bias", "pw1_row", "pw1_bias"))
    esz = 4 if adt == torch.float32 else 2
    sc = cfm.LayerTrainScratch()
    sc.dxn = cfm.scratch("t_dxn", M * D, torch.float32, dev).data_ptr()
    for name, wd in (("dz", FF), ("dyb", D), ("ds", D), ("dglu", D), ("du", 2 * D), ("dctx", D), ("dqkv", 3 * D)):
        setattr(sc, name, cfm.scratch("t_" + name, M * wd, adt, dev).data_ptr())
    sc.delta = cfm.scratch("attn_delta", B * H * T, torch.float32, dev).data_ptr()
    sc.ln_ws = cfm.scratch("ln_bwd", cfm.lib().cfm_layernorm_bwd_ws(M, D), torch.float32, dev).data_ptr()
    sc.dwbn_ws = cfm.scratch("dwbn", cfm.lib().cfm_dwconv_bn_ws(B, T, D), torch.float32, dev).data_ptr()
    sc.dy_ws = cfm.scratch("dwbn_dy", M * D, torch.float32, dev).data_ptr()
    if OVERLAP_WGRAD:                                       # weight-gradient products on a second stream, overlapped with the input-gradient chain
        for name, wd in (("dz2", FF), ("dyb2", D), ("dyb3", D), ("dyb4", D)):
            setattr(sc, name, cfm.scratch("t_" + name, M * wd, adt, dev).data_ptr())
        io.side_stream = _side_stream(dev).cuda_stream
    else:
        io.side_stream = None
    dyc = _f32c(dy.reshape(M, D))
    dx = torch.empty((M, D), dtype=torch.float32, device=dev)
    cfm.check(cfm.lib().cfm_encoder_layer_train_backward(ctypes.byref(w), ctypes.byref(io), ctypes.byref(sv), ctypes.byref(sc), ctypes.byref(g), x0.data_ptr(),
                                                         dyc.data_ptr(), dx.data_ptr(), cfm.stream()), "cfm_encoder_layer_train_backward")
    if sink is not None:
        sink[1]()                                           # the trainer's ready hook (bucket all-reduce), after the kernels are enqueued
        return dx, None, lay
    return dx, slab, lay


class EncoderLayerFn(torch.autograd.Function):
    """One conformer block in train mode (encoder_layer.py:49-71): four residual sub-blocks + norm_final, one autograd node."""

    @staticmethod
    def forward(ctx, x, layer, prec, mask8, m_str, keep, *params):
        B, T, D = x.shape
        rel = layer.use_relative
        ctx.comp = None
        ctx.flat = len(params) == 1 and params[0] is getattr(layer, "_flat_leaf", None)
        comp = _composite_ok(layer, x, ctx.flat)
        if ctx.flat and not comp:
            raise RuntimeError("a block registered with a flat parameter leaf (trainer.py) needs the composite train path")
        if comp:
            pks = packing.pack_layer_train(layer, prec, rel) if USE_PACK_KERNEL else (
                packing.pack_ffn_train(layer.feed_forward_macaron, prec), packing.pack_mhsa_train(layer.self_attn, prec, rel),
                packing.pack_conv_module_train(layer.conv_module, prec), packing.pack_ffn_train(layer.feed_forward, prec))
            return _layer_composite_forward(ctx, _f32c(x.reshape(B * T, D)), layer, prec, mask8, m_str, keep, pks, B, T, D).view(B, T, D)
        pks = (packing.pack_ffn_train(layer.feed_forward_macaron, prec), packing.pack_mhsa_train(layer.self_attn, prec, rel),
               packing.pack_conv_module_train(layer.conv_module, prec), packing.pack_ffn_train(layer.feed_forward, prec))
        ln = lambda m: (m.weight.detach(), m.bias.detach())
        x0 = _f32c(x.reshape(B * T, D))
        # dropout (encoder_layer.py:56-69 under module.train()): the shared nn.Dropout(feedforward_dropout) on each of the four branch
        # outputs, each FFN's own dropout on its hidden activation, the attention's on its probabilities (and, plain MHSA only, on its output)
        p_br, p_a = layer.dropout.p, layer.self_attn.dropout.p
        p_hm, p_h = layer.feed_forward_macaron.dropout.p, layer.feed_forward.dropout.p
        seed = draw_seed() if max(p_br, p_a, p_hm, p_h) > 0 else 0
        dr = dict(hm=_drop(p_hm, seed, 1), om=_drop(p_br, seed, 2), a=_drop(p_a, seed, 3), oa=_drop(p_br, seed, 4),
                  oa2=None if rel else _drop(p_a, seed, 5), oc=_drop(p_br, seed, 6), h=_drop(p_h, seed, 7), o=_drop(p_br, seed, 8))
        x1, s1 = ffn_fwd(pks[0], x0, ln(layer.norm_ff_macaron), prec, 0.5, drop_h=dr["hm"], drop_o=dr["om"])
        x2, s2 = mhsa_fwd(layer.self_attn, pks[1], x1, ln(layer.norm_mha), B, T, mask8, m_str, prec, rel, dr["a"], dr["oa"], dr["oa2"])
        x3, s3 = conv_module_fwd(layer.conv_module, pks[2], x2, ln(layer.norm_conv), B, T, keep, prec, drop_o=dr["oc"])
        x4, s4 = ffn_fwd(pks[3], x3, ln(layer.norm_ff), prec, 0.5, drop_h=dr["h"], drop_o=dr["o"])
        y = cfm.layernorm(x4, *ln(layer.norm_final))[0]
        ctx.args = (layer, prec, mask8, m_str, keep, pks, (s1, s2, s3, s4, x4), B, T, D, dr)
        return y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        if ctx.comp is not None:
            layer, B, T, D = ctx.comp[0], ctx.comp[-3], ctx.comp[-2], ctx.comp[-1]
            dx, slab, lay = _layer_composite_backward(ctx, dy)
            if ctx.flat:                                              # one gradient for the block's flat parameter leaf (trainer.py)
                return (dx.view(B, T, D), None, None, None, None, None, slab)
            grads = []
            for name, p in layer.named_parameters():
                off, n = lay["layout"][name]
                grads.append(slab[off:off + n].view(p.shape) if p.requires_grad else None)
            return (dx.view(B, T, D), None, None, None, None, None) + tuple(grads)
        layer, prec, mask8, m_str, keep, pks, (s1, s2, s3, s4, x4), B, T, D, dr = ctx.args
        rel = layer.use_relative
        ln = lambda m: (m.weight.detach(), m.bias.detach())
        grads = {}

        def put(prefix, g, norm_name, lng):
            for k, v in g.items():
                grads[prefix + k] = v
            grads[norm_name + ".weight"], grads[norm_name + ".bias"] = lng

        d, dgf, dbf = cfm.layernorm_bwd(x4, _f32c(dy.reshape(B * T, D)), layer.norm_final.weight.detach())
        grads["norm_final.weight"], grads["norm_final.bias"] = dgf, dbf
        d, g, lng = ffn_bwd(pks[3], s4, d, ln(layer.norm_ff), prec, 0.5, drop_h=dr["h"], drop_o=dr["o"])
        put("feed_forward.", g, "norm_ff", lng)
        d, g, lng = conv_module_bwd(layer.conv_module, pks[2], s3, d, ln(layer.norm_conv), B, T, keep, prec, drop_o=dr["oc"])
        put("conv_module.", g, "norm_conv", lng)
        d, g, lng = mhsa_bwd(layer.self_attn, pks[1], s2, d, ln(layer.norm_mha), B, T, mask8, m_str, prec, rel, dr["a"], dr["oa"], dr["oa2"])
        put("self_attn.", g, "norm_mha", lng)
        d, g, lng = ffn_bwd(pks[0], s1, d, ln(layer.norm_ff_macaron), prec, 0.5, drop_h=dr["hm"], drop_o=dr["om"])
        put("feed_forward_macaron.", g, "norm_ff_macaron", lng)
        names, tensors = _params(layer)
        return (d.view(B, T, D), None, None, None, None, None) + tuple(_ordered(names, grads, tensors))


# ----------------------------------------------------------------------------------------------------------------------
# the whole block STACK over an accumulation WINDOW (round 3): one host call each way (csrc/train_layer.cpp cfm_encoder_train_forward /
# _backward), the micro-batches of the window concatenated along the row axis (cfm.h cfm_train_group), each block's weight gradients as ONE
# grouped launch at the end of its backward, gradients written straight into the data-parallel trainer's flat buffer.
# ----------------------------------------------------------------------------------------------------------------------
_SAVED_ACT = ("xn1", "z1", "h1", "xn2", "qkv", "ctx", "xn3", "u", "glu", "s", "xn4", "z2", "h2")


def stack_supported(layers, flat):
    """The stack path needs every block on the composite train path with the same sizes and dropout rates (it shares one io struct)."""
    l0 = layers[0]
    bn0 = l0.conv_module.norm
    key = lambda l: (l.encoder_dim, l.hidden_dim, l.num_heads, l.kernel_size, l.use_relative, l.dropout.p, l.self_attn.dropout.p,
                     l.feed_forward_macaron.dropout.p, l.feed_forward.dropout.p)
    return (USE_COMPOSITE and all(_composite_ok(l, None, flat) and key(l) == key(l0) and l.conv_module.norm.momentum is not None and
                                  l.conv_module.norm.track_running_stats == bn0.track_running_stats for l in layers))


def _stack_weights(owner, layers, prec, flat):
    """ctypes array of the blocks' weight structs (cached while the packs stay the same objects)."""
    rel = layers[0].use_relative
    pks = packing.pack_stack_train(owner, layers, prec, rel, flat) if USE_PACK_KERNEL else tuple(packing.pack_layer_train(l, prec, rel) for l in layers)
    hit = owner.__dict__.get("_stack_w")
    if hit is not None and len(hit[0]) == len(pks) and all(a is b for a, b in zip(hit[0], pks)):
        return hit[1], pks
    arr = (cfm.LayerTrainWeights * len(layers))()
    for i, (l, pk) in enumerate(zip(layers, pks)):
        arr[i] = _train_weights_struct(l, pk)
    owner.__dict__["_stack_w"] = (pks, arr)
    return arr, pks


def _stack_grads(owner, layers, bases, lays, tag):
    """ctypes array of the blocks' gradient destinations: block i's slab starts at address bases[i]; cached per destination."""
    key = (tag, tuple(bases))
    hit = owner.__dict__.get("_stack_g")
    if hit is not None and hit[0] == key:
        return hit[1]
    arr = (cfm.LayerTrainGrads * len(layers))()
    for g, base, lay in zip(arr, bases, lays):
        g.slab = base
        for name, field in _GRAD_FIELDS.items():
            if name in lay["layout"]:
                setattr(g, field, base + 4 * lay["layout"][name][0])
        g.q_bias = base + 4 * lay["layout"]["self_attn.linear_q.bias"][0]
        g.qkv_row_off, g.qkv_bias_off, g.pw1_row_off, g.pw1_bias_off = (lay[k].data_ptr() for k in ("qkv_row", "qkv_bias", "pw1_row", "pw1_bias"))
        if "qkv_bias2" in lay:
            g.qkv_bias_off2 = lay["qkv_bias2"].data_ptr()
    owner.__dict__["_stack_g"] = (key, arr)
    return arr


class EncoderStackFn(torch.autograd.Function):
    """All conformer blocks of the encoder in train mode over one accumulation window (encoder.py:72-73 under module.train())."""

    @staticmethod
    def forward(ctx, x, owner, layers, prec, groups, keep, flat, *params):
        """x f32 [M,D]: the window's rows, micro-batch after micro-batch; groups: [(B, T, mask8 | None, (m_sb, m_sq))]; keep u8 [M] | None."""
        dev, adt = x.device, prec.act_dtype
        L, G = len(layers), len(groups)
        l0 = layers[0]
        D, FF, H = l0.encoder_dim, l0.hidden_dim, l0.num_heads
        M = sum(B * T for B, T, _, _ in groups)
        BHT = sum(B * H * T for B, T, _, _ in groups)
        if tuple(x.shape) != (M, D) or x.dtype != torch.float32 or not x.is_contiguous():
            raise RuntimeError("EncoderStackFn: rows must be contiguous float32 (%d,%d), got %s %s" % (M, D, tuple(x.shape), x.dtype))
        if G > 8:
            raise RuntimeError("EncoderStackFn: at most 8 micro-batches per window")
        w_arr, pks = _stack_weights(owner, layers, prec, flat)
        esz = 4 if adt == torch.float32 else 2
        widths = dict(xn1=D, z1=FF, h1=FF, xn2=D, qkv=3 * D, ctx=D, xn3=D, u=2 * D, glu=D, s=D, xn4=D, z2=FF, h2=FF)
        per_act = M * sum(widths.values())
        per_f32 = 5 * M * D + BHT + G * 4 * D
        act = torch.empty((L * per_act,), dtype=adt, device=dev)
        f32 = torch.empty((L * per_f32 + L * M * D,), dtype=torch.float32, device=dev)
        sv = (cfm.LayerTrainSaved * L)()
        xs = (ctypes.c_void_p * (L + 1))()
        xs[0] = x.data_ptr()
        pa, pf = act.data_ptr(), f32.data_ptr()
        for l in range(L):
            o = pa + l * per_act * esz
            for name in _SAVED_ACT:
                setattr(sv[l], name, o)
                o += M * widths[name] * esz
            o = pf + l * per_f32 * 4
            for name, n in (("x1", M * D), ("x2", M * D), ("x3", M * D), ("x4", M * D), ("c", M * D), ("lse", BHT), ("stats", G * 4 * D)):
                setattr(sv[l], name, o)
                o += n * 4
            xs[l + 1] = pf + (L * per_f32 + l * M * D) * 4
        garr = (cfm.TrainGroup * G)()
        row0 = 0
        for g, (B, T, m8, m_str) in zip(garr, groups):
            g.B, g.T, g.row0, g.attn_mask, g.am_sb, g.am_sq = B, T, row0, cfm.ptr(m8), m_str[0], m_str[1]
            row0 += B * T
        io = cfm.LayerTrainIO()
        io.D, io.H, io.FF, io.ktaps, io.act_dtype, io.w_dtype = D, H, FF, l0.kernel_size, prec.act_code, prec.w_code
        io.pad_valid = cfm.ptr(keep)
        p_br, p_a = l0.dropout.p, l0.self_attn.dropout.p
        io.p_hidden_m, io.p_hidden, io.p_branch, io.p_attn = l0.feed_forward_macaron.dropout.p, l0.feed_forward.dropout.p, p_br, p_a
        io.p_attn_out = 0.0 if l0.use_relative else p_a
        io.seed = draw_seed() if max(io.p_hidden_m, io.p_hidden, p_br, p_a) > 0 else 0
        io.deterministic = 1 if cfm.ops._deterministic[0] else 0
        io.n_groups, io.groups, io.defer_wgrad = G, garr, 1
        ws = sum(cfm.lib().cfm_dwconv_bn_ws(B, T, D) for B, T, _, _ in groups)
        sc = cfm.LayerTrainScratch()
        sc.dwbn_ws = cfm.scratch("dwbn", ws, torch.float32, dev).data_ptr()
        cfm.check(cfm.lib().cfm_encoder_train_forward(L, w_arr, ctypes.byref(io), sv, ctypes.byref(sc), xs, cfm.stream()), "cfm_encoder_train_forward")
        if l0.conv_module.norm.track_running_stats:
            torch._foreach_add_([l.conv_module.norm.num_batches_tracked for l in layers], G)
        ctx.st = (owner, layers, prec, groups, keep, flat, w_arr, pks, io, garr, sv, xs, (act, f32, x), M, BHT, ws)
        return f32[L * per_f32 + (L - 1) * M * D:].view(M, D)

    @staticmethod
    def backward(ctx, dy):
        owner, layers, prec, groups, keep, flat, w_arr, pks, io, garr, sv, xs, held, M, BHT, ws = ctx.st
        dev, adt = dy.device, prec.act_dtype
        L = len(layers)
        l0 = layers[0]
        D, FF = l0.encoder_dim, l0.hidden_dim
        det = cfm.ops._deterministic[0]
        # where the gradients go: the trainer's flat gradient buffer itself (each block registered with a sink: trainer.py) -- nothing is
        # returned to autograd for the leaves, the trainer's ready hook is called from the per-block callback -- or one zero-filled slab per
        # block (plain autograd / deterministic sums), returned as the leaves' / parameters' gradients
        sinks = [l.__dict__.get("_flat_grad_sink") for l in layers] if (flat and not det) else None
        if sinks is not None and any(s is None or s[0].device != dev for s in sinks):
            sinks = None
        lays = [layer_grad_layout(l, l.__dict__.get("_flat_grad_offsets") if flat else None) for l in layers]
        for lay, l in zip(lays, layers):
            if "qkv_bias2" not in lay and "self_attn.pos_bias_u" in lay["layout"]:
                u0 = lay["layout"]["self_attn.pos_bias_u"][0]
                lay["qkv_bias2"] = torch.cat([u0 + torch.arange(D, dtype=torch.int64), torch.full((2 * D,), -1, dtype=torch.int64)]).to(dev)
        if sinks is not None:
            slabs = None
            g_arr = _stack_grads(owner, layers, [s[0].data_ptr() for s in sinks], lays, "sink")
            io.grads_accumulate = 1
        else:
            sizes = [(l.__dict__["_flat_leaf"].numel() if flat else lay["numel"]) for l, lay in zip(layers, lays)]
            slab_all = torch.zeros((sum(sizes),), dtype=torch.float32, device=dev)
            slabs, o = [], 0
            for n in sizes:
                slabs.append(slab_all[o:o + n])
                o += n
            g_arr = _stack_grads(owner, layers, [s.data_ptr() for s in slabs], lays, "slab")
            io.grads_accumulate = 0
        # the grouped weight-gradient launch of each block on a side stream, beside the next block's chain (two scratch sets: its operands
        # must outlive the block); measured at config 3 -- see WGRAD_BESIDE
        beside = WGRAD_BESIDE and adt != torch.float32
        io.side_stream = _side_stream(dev).cuda_stream if beside else None
        n_sc = 2 if beside else 1
        scs = (cfm.LayerTrainScratch * n_sc)()
        for si, sc in enumerate(scs):
            tag = "t%d_" % si
            sc.dxn = cfm.scratch("t_dxn", M * D, torch.float32, dev).data_ptr()
            for name, wd in (("dz", FF), ("dz2", FF), ("dyb", D), ("dyb2", D), ("dyb3", D), ("dyb4", D), ("du", 2 * D), ("dqkv", 3 * D)):
                setattr(sc, name, cfm.scratch(tag + name, M * wd, adt, dev).data_ptr())         # operands of the deferred products: per set
            for name, wd in (("ds", D), ("dglu", D), ("dctx", D)):
                setattr(sc, name, cfm.scratch("t_" + name, M * wd, adt, dev).data_ptr())
            sc.delta = cfm.scratch("attn_delta", BHT, torch.float32, dev).data_ptr()
            sc.ln_ws = cfm.scratch("ln_bwd", cfm.lib().cfm_layernorm_bwd_ws(M, D), torch.float32, dev).data_ptr()
            sc.dwbn_ws = cfm.scratch("dwbn", ws, torch.float32, dev).data_ptr()
            sc.dy_ws = cfm.scratch("dwbn_dy", M * D, torch.float32, dev).data_ptr()
        dyc = _f32c(dy.reshape(M, D))
        bufs = torch.empty((2, M, D), dtype=torch.float32, device=dev)
        failed = []

        def done(layer, _user):
            if sinks is not None:
                try:
                    sinks[layer][1]()                                   # the trainer's ready hook: this block's bucket may be all-reduced
                except BaseException as e:                              # noqa: BLE001 -- ctypes would swallow it: re-raised below
                    failed.append(e)

        cb = cfm.LAYER_DONE_FN(done)
        out = ctypes.c_void_p()
        cfm.check(cfm.lib().cfm_encoder_train_backward(L, w_arr, ctypes.byref(io), sv, scs, n_sc, g_arr, xs, dyc.data_ptr(), bufs[0].data_ptr(),
                                                       bufs[1].data_ptr(), cb, None, ctypes.byref(out), cfm.stream()), "cfm_encoder_train_backward")
        if failed:
            raise failed[0]
        dx = bufs[0] if out.value == bufs[0].data_ptr() else bufs[1]
        head = (dx, None, None, None, None, None, None)
        if sinks is not None:
            return head + (None,) * L
        if flat:
            return head + tuple(slabs)
        grads = []
        for l, slab, lay in zip(layers, slabs, lays):
            for name, p in l.named_parameters():
                off, n = lay["layout"][name]
                grads.append(slab[off:off + n].view(p.shape) if p.requires_grad else None)
        return head + tuple(grads)


class CTCLossFn(torch.autograd.Function):
    """sum_b nll_b / padded label length on top of ctc_lo (decoder.py:19-22)."""

    @staticmethod
    def forward(ctx, enc_out, mod, prec, enc_lens, labels, label_lens, weight, bias):
        pk = packing.pack_ctc_train(mod, prec)
        B, T, D = enc_out.shape
        x2 = _f32c(enc_out.reshape(B * T, D))
        logits = _gemm(x2, pk.w, bias=pk.b, w_lo=pk.w_lo, out_dtype=torch.float32).view(B, T, pk.Vp)
        nll, state = cfm.ctc_nll_train(logits, pk.V, enc_lens, labels, label_lens)
        ctx.args = (prec, pk, x2, logits, state, enc_lens, labels, label_lens, B, T, D)
        return nll.sum() / labels.size(1)

    @staticmethod
    def backward(ctx, gout):
        prec, pk, x2, logits, state, enc_lens, labels, label_lens, B, T, D = ctx.args
        gdev = _f32c(gout.reshape(1))
        dlog = cfm.ctc_grad(logits, pk.V, enc_lens, labels, label_lens, state, gscale=1.0 / labels.size(1), gscale_dev=gdev).view(B * T, pk.Vp)
        dW, db = cfm.gemm_tn(dlog, x2, want_colsum=True, mma_code=prec.w_code, split=prec.split)
        dx = _gemm(dlog, pk.wt, w_lo=pk.wt_lo, out_dtype=torch.float32)
        return dx.view(B, T, D), None, None, None, None, None, dW[:pk.V], db[:pk.V]


class CTCWindowLossFn(torch.autograd.Function):
    """The CTC heads of an accumulation window in one projection: the window's rows are one [M, D] matrix (ConformerEncoder.forward_window), so the
    vocabulary projection, its weight gradient and its input gradient each run ONCE over all micro-batches' rows; the recursions and the per-row
    gradients run per micro-batch on slices of the one logits matrix (each micro-batch has its own T', label width and normaliser).
    Returns the micro-batches' losses (sum_b nll_b / padded label length each, decoder.py:19-22) as a 1-D tensor."""

    @staticmethod
    def forward(ctx, rows, mod, prec, groups, weight, bias):
        """rows f32 [M, D]; groups: [(B, T, enc_lens i32 [B], labels i32 [B,U], label_lens i32 [B])], rows of micro-batch g at sum of earlier B*T."""
        pk = packing.pack_ctc_train(mod, prec)
        x2 = _f32c(rows)
        logits = _gemm(x2, pk.w, bias=pk.b, w_lo=pk.w_lo, out_dtype=torch.float32)
        problems, r0 = [], 0
        for B, T, enc_lens, labels, label_lens in groups:
            problems.append((logits[r0:r0 + B * T].view(B, T, pk.Vp), enc_lens, labels, label_lens))
            r0 += B * T
        res = cfm.ctc_nll_train_groups(problems, pk.V)         # the recursions of all micro-batches in one launch
        losses = [nll.sum() / pr[2].size(1) for (nll, _), pr in zip(res, problems)]
        states = [st for _, st in res]
        if r0 != x2.shape[0]:
            raise RuntimeError("CTCWindowLossFn: the micro-batches cover %d rows, the row matrix has %d" % (r0, x2.shape[0]))
        ctx.args = (prec, pk, x2, logits, states, groups)
        return torch.stack(losses)

    @staticmethod
    def backward(ctx, gout):
        prec, pk, x2, logits, states, groups = ctx.args
        gdev = _f32c(gout.reshape(-1))
        dlog = torch.empty_like(logits)
        r0 = 0
        for gi, ((B, T, enc_lens, labels, label_lens), state) in enumerate(zip(groups, states)):
            cfm.ctc_grad(logits[r0:r0 + B * T].view(B, T, pk.Vp), pk.V, enc_lens, labels, label_lens, state, gscale=1.0 / labels.size(1),
                         gscale_dev=gdev[gi:gi + 1], out=dlog[r0:r0 + B * T].view(B, T, pk.Vp))
            r0 += B * T
        dW, db = cfm.gemm_tn(dlog, x2, want_colsum=True, mma_code=prec.w_code, split=prec.split)
        dx = _gemm(dlog, pk.wt, w_lo=pk.wt_lo, out_dtype=torch.float32)
        return dx, None, None, None, dW[:pk.V], db[:pk.V]
