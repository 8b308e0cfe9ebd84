"""Tensor-level wrappers over the C ABI.  Each function validates device / dtype / layout on the host,
allocates its output with torch.empty on the same device, and enqueues on the current HIP stream."""
import ctypes

import torch

import cfm as _c

__all__ = ["gemm", "ffn_fused", "ffn_fused_supported", "rowchain", "rowchain_supported", "ffn_partial", "layernorm", "attention", "kv_cache_pack", "dwconv_bn_silu", "conv1_relu", "conv1_relu_mma_supported", "ctc_nll", "joint_act", "valid_mask", "chunk_mask",
           "attn_mask_combine", "cast", "add_rows", "scratch", "prof_enable", "prof_reset", "prof_table", "as_u8_mask"]


def _rows2d(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("cfm.%s: expected a 2-D tensor with unit inner stride, got shape %s strides %s" % (name, tuple(t.shape), t.stride()))
    return t


# a grow-only arena of reusable device buffers (workspace owned by the extension side of the boundary)
_arena = {}


def scratch(tag, numel, dtype, device):
    key = (tag, dtype, device)
    buf = _arena.get(key)
    if buf is None or buf.numel() < numel:
        buf = torch.empty(max(int(numel), 1), dtype=dtype, device=device)
        _arena[key] = buf
    return buf[:numel]


def gemm(a, w, bias=None, w_lo=None, out=None, out_dtype=None, act=_c.ACT_NONE, residual=None, alpha=1.0, row_mask=None,
         mask_mode=0, conv=None, tile=0, n_out=None, w_frag=None):
    """out = epilogue(a[M,K] . w[N,K]^T); see include/cfm.h cfm_gemm.  conv=(C,T1,F1,T2,F2,M) selects the implicit
    3x3/stride-2 convolution over a channels-last image `a` of shape [B,T1,F1,C]."""
    _c.require_hip(a, w, bias, w_lo, out, residual, row_mask)
    w = _rows2d(w, "gemm(w)")
    N, K = w.shape
    d = _c.GemmDesc()
    if conv is None:
        a = _rows2d(a, "gemm(a)")
        M = a.shape[0]
        if a.shape[1] != K:
            raise ValueError("cfm.gemm: a is %s but w is %s" % (tuple(a.shape), tuple(w.shape)))
        d.lda = a.stride(0)
    else:
        C, T1, F1, T2, F2, M = conv
        if not a.is_contiguous():
            raise ValueError("cfm.gemm(conv): image must be contiguous channels-last [B,T1,F1,C]")
        d.conv_C, d.conv_T1, d.conv_F1, d.conv_T2, d.conv_F2 = C, T1, F1, T2, F2
        d.lda = 0
    cols = N // 2 if act == _c.ACT_GLU else N
    if out is None:
        odt = out_dtype if out_dtype is not None else (residual.dtype if residual is not None else torch.float32)
        out = torch.empty((M, cols), dtype=odt, device=a.device)
    else:
        out = _rows2d(out, "gemm(out)")
        if out.shape[0] != M or out.shape[1] != cols:
            raise ValueError("cfm.gemm: out is %s, expected (%d,%d)" % (tuple(out.shape), M, cols))
    if residual is not None:
        residual = _rows2d(residual, "gemm(residual)")
        if residual.dtype != torch.float32 or out.dtype != torch.float32 or tuple(residual.shape) != (M, cols):
            raise ValueError("cfm.gemm: residual accumulate needs f32 residual/out of shape (%d,%d)" % (M, cols))
        d.ldr = residual.stride(0)
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("cfm.gemm: bias must be contiguous f32 [N]")
    if row_mask is not None and (row_mask.dtype != torch.uint8 or row_mask.numel() != M or not row_mask.is_contiguous()):
        raise ValueError("cfm.gemm: row_mask must be contiguous uint8 [M]")
    if w_lo is not None and (w_lo.shape != w.shape or w_lo.dtype != w.dtype or not w_lo.is_contiguous()):
        raise ValueError("cfm.gemm: w_lo must match w")
    if w_frag is not None:
        _c.require_hip(w_frag)
        if w_frag.dtype != w.dtype or w_frag.numel() < N * K:
            raise ValueError("cfm.gemm: w_frag must be the fragment-major pack of w")
        d.W_frag = _c.ptr(w_frag)
    d.A, d.W, d.W_lo, d.bias, d.residual, d.row_mask, d.C = _c.ptr(a), _c.ptr(w), _c.ptr(w_lo), _c.ptr(bias), _c.ptr(residual), _c.ptr(row_mask), _c.ptr(out)
    d.ldc = out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.a_dtype, d.w_dtype, d.c_dtype = _c.dt_code(a), _c.dt_code(w), _c.dt_code(out)
    d.act, d.alpha, d.tile, d.mask_mode = act, alpha, tile, mask_mode
    _c.check(_c.lib().cfm_gemm(ctypes.byref(d), _c.stream()), "cfm_gemm")
    return out


def ffn_fused_supported(D, FF, prec):
    return (not prec.split) and D in (144, 256) and FF % 32 == 0 and FF <= 2048


def ffn_fused(x, w1f, w2f, b1, b2, FF, act=_c.ACT_SILU, ln=None, alpha=1.0, add_x=False, ln1=None, ln2=None, out_f32=None,
              want_f32=True, out16_dtype=None, eps=1e-5):
    """One-launch feed-forward block (include/cfm.h cfm_ffn_fused).  x f32 [M,D]; ln/ln1/ln2 are (gain, bias) pairs or None.
    Returns (out_f32 | None, out16 | None)."""
    _c.require_hip(x, w1f, w2f, b1, b2, out_f32)
    x = _rows2d(x, "ffn_fused(x)")
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("cfm.ffn_fused: x must be contiguous float32 [M,D]")
    M, D = x.shape
    d = _c.FfnDesc()
    d.x, d.w1f, d.w2f, d.b1, d.b2 = _c.ptr(x), _c.ptr(w1f), _c.ptr(w2f), _c.ptr(b1), _c.ptr(b2)
    for name, pair in (("ln", ln), ("ln1", ln1), ("ln2", ln2)):
        if pair is not None:
            setattr(d, name + "_g", _c.ptr(pair[0]))
            setattr(d, name + "_b", _c.ptr(pair[1]))
    if want_f32 and out_f32 is None:
        out_f32 = torch.empty((M, D), dtype=torch.float32, device=x.device)
    out16 = torch.empty((M, D), dtype=out16_dtype, device=x.device) if out16_dtype is not None else None
    d.out_f32, d.out16 = _c.ptr(out_f32), _c.ptr(out16)
    d.M, d.D, d.FF = M, D, FF
    d.w_dtype = _c.dt_code(w1f)
    d.out16_dtype = _c.dt_code(out16) if out16 is not None else 0
    d.act, d.add_x, d.alpha, d.eps = act, 1 if add_x else 0, alpha, eps
    _c.check(_c.lib().cfm_ffn_fused(ctypes.byref(d), _c.stream()), "cfm_ffn_fused")
    return out_f32, out16


def rowchain_supported(D, FF, prec):
    return (not prec.split) and bool(_c.lib().cfm_rowchain_supported(D, FF))


def ffn_partial(x, ln, w1f, w2f, b1, FF, y0, y1, pending=None, head=None, x_out=None, eps=1e-5):
    """Partial feed-forward over FF halves on 64-row tiles (include/cfm.h cfm_ffn_partial): writes y0, y1 (f32 [M,D]).
    pending = (py0, py1, pb2, palpha, pln | None);  head = (a16, w_frag, bias, mask | None)."""
    _c.require_hip(x, w1f, w2f, b1, y0, y1, x_out)
    M, D = x.shape
    d = _c.FfnPartialDesc()
    d.x, d.ln_g, d.ln_b, d.w1f, d.w2f, d.b1, d.y0, d.y1, d.x_out = (_c.ptr(x), _c.ptr(ln[0]), _c.ptr(ln[1]), _c.ptr(w1f), _c.ptr(w2f),
                                                                     _c.ptr(b1), _c.ptr(y0), _c.ptr(y1), _c.ptr(x_out))
    if pending is not None:
        d.py0, d.py1, d.pb2, d.palpha = _c.ptr(pending[0]), _c.ptr(pending[1]), _c.ptr(pending[2]), pending[3]
        if pending[4] is not None:
            d.pln_g, d.pln_b = _c.ptr(pending[4][0]), _c.ptr(pending[4][1])
    if head is not None:
        d.head_a, d.head_w, d.head_b, d.head_mask = _c.ptr(head[0]), _c.ptr(head[1]), _c.ptr(head[2]), _c.ptr(head[3])
    d.M, d.D, d.FF, d.w_dtype, d.eps = M, D, FF, _c.dt_code(w1f), eps
    _c.check(_c.lib().cfm_ffn_partial(ctypes.byref(d), _c.stream()), "cfm_ffn_partial")


def rowchain(M, D, w_code, x=None, head=None, ln=None, ln_mask=None, ffn=None, alpha=1.0, ln1=None, ln2=None, out_f32=None, out16=None,
             tail=None, eps=1e-5, pending=None, dw=None):
    """One-launch row-local chain (include/cfm.h cfm_rowchain).
    dw = (taps f32 [D,15], bias, bn_scale, bn_shift, T): depthwise conv + BatchNorm + SiLU applied to the head input first;
    head = (a16 [M,D], w_frag, bias, residual f32 [M,D], out_mask u8 [M] | None);  ffn = (w1f, w2n, b1, b2, FF) with
    w1f = pack_frag_major(W1), w2n = pack_frag_major(W2) (natural k order, NOT ffn_fused's permuted w2f);
    tail = (w_frag, bias, N, glu, out 16-bit [M, N or N/2]);  ln/ln1/ln2 = (gain, bias)."""
    d = _c.RowChainDesc()
    keep = [x, out_f32, out16, ln_mask]
    d.x, d.out_f32, d.out16, d.ln_mask = _c.ptr(x), _c.ptr(out_f32), _c.ptr(out16), _c.ptr(ln_mask)
    if head is not None:
        a16, hw, hb, res, hm = head
        _c.require_hip(a16, hw, hb, res, hm)
        d.head_a, d.head_w, d.head_b, d.head_res, d.head_mask = _c.ptr(a16), _c.ptr(hw), _c.ptr(hb), _c.ptr(res), _c.ptr(hm)
        keep += list(head)
    for name, pair in (("ln", ln), ("ln1", ln1), ("ln2", ln2)):
        if pair is not None:
            setattr(d, name + "_g", _c.ptr(pair[0]))
            setattr(d, name + "_b", _c.ptr(pair[1]))
    if ffn is not None:
        d.w1f, d.w2n, d.b1, d.b2, d.FF = _c.ptr(ffn[0]), _c.ptr(ffn[1]), _c.ptr(ffn[2]), _c.ptr(ffn[3]), ffn[4]
    if tail is not None:
        d.tail_w, d.tail_b, d.tail_N, d.tail_glu, d.tail_out = _c.ptr(tail[0]), _c.ptr(tail[1]), tail[2], 1 if tail[3] else 0, _c.ptr(tail[4])
    if pending is not None:                     # (py0, py1, pb2, palpha, pln | None): finish a partial FFN while loading the rows
        d.py0, d.py1, d.pb2, d.palpha = _c.ptr(pending[0]), _c.ptr(pending[1]), _c.ptr(pending[2]), pending[3]
        if pending[4] is not None:
            d.pln_g, d.pln_b = _c.ptr(pending[4][0]), _c.ptr(pending[4][1])
    if dw is not None:
        _c.require_hip(*dw[:4])
        d.dw_w, d.dw_b, d.dw_scale, d.dw_shift, d.dw_T, d.dw_K = _c.ptr(dw[0]), _c.ptr(dw[1]), _c.ptr(dw[2]), _c.ptr(dw[3]), dw[4], dw[0].shape[1]
    _c.require_hip(x, out_f32, out16)
    d.M, d.D, d.w_dtype, d.alpha, d.eps = M, D, w_code, alpha, eps
    _c.check(_c.lib().cfm_rowchain(ctypes.byref(d), _c.stream()), "cfm_rowchain")


def layernorm(x, g1, b1, out1=None, out1_dtype=None, g2=None, b2=None, out2=None, out2_dtype=None, row_mask=None, eps=1e-5,
              want1=True):
    """y1 = LN(x;g1,b1) -> out1 (if want1);  y2 = LN(y1;g2,b2) (or y1) masked -> out2 (if out2/out2_dtype given)."""
    _c.require_hip(x, g1, b1, g2, b2, out1, out2, row_mask)
    x = _rows2d(x, "layernorm(x)")
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("cfm.layernorm: x must be contiguous float32 [M,D]")
    M, D = x.shape
    if want1 and out1 is None:
        out1 = torch.empty((M, D), dtype=out1_dtype or torch.float32, device=x.device)
    if out2 is None and out2_dtype is not None:
        out2 = torch.empty((M, D), dtype=out2_dtype, device=x.device)
    for o in (out1, out2):
        if o is not None and (tuple(o.shape) != (M, D) or not o.is_contiguous()):
            raise ValueError("cfm.layernorm: outputs must be contiguous [M,D]")
    _c.check(_c.lib().cfm_layernorm(_c.ptr(x), _c.ptr(g1), _c.ptr(b1), _c.ptr(out1), _c.dt_code(out1) if out1 is not None else 0,
                                    _c.ptr(g2), _c.ptr(b2), _c.ptr(out2), _c.dt_code(out2) if out2 is not None else 0,
                                    _c.ptr(row_mask), eps, M, D, _c.stream()), "cfm_layernorm")
    return out1, out2


def as_u8_mask(mask):
    """bool / numeric mask -> contiguous uint8 0/1 view (no copy when already a contiguous bool tensor)."""
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    if mask.dtype == torch.uint8:
        return mask.contiguous()
    return (mask != 0).contiguous().view(torch.uint8)


def attention(q, k, v, B, H, Tq, Tk, dk, q_str, k_str, v_str, out, p=None, p_str=(0, 0), bias_u=None, bias_v=None, mask=None,
              mask_str=(0, 0), mma_code=_c.BF16, split=False, scale=None):
    """Fused attention; *_str are (batch stride, time stride[, head stride]) in ELEMENTS (see include/cfm.h)."""
    _c.require_hip(q, k, v, p, out, mask, bias_u, bias_v)
    d = _c.AttnDesc()
    d.q, d.k, d.v, d.p, d.out = _c.ptr(q), _c.ptr(k), _c.ptr(v), _c.ptr(p), _c.ptr(out)
    d.bias_u, d.bias_v, d.mask = _c.ptr(bias_u), _c.ptr(bias_v), _c.ptr(mask)
    d.q_sb, d.q_st = q_str
    d.k_sb, d.k_st, d.k_sh = k_str
    d.v_sb, d.v_st, d.v_sh = v_str
    d.p_sb, d.p_st = p_str
    d.m_sb, d.m_sq = mask_str
    d.B, d.H, d.Tq, d.Tk, d.dk = B, H, Tq, Tk, dk
    d.q_dtype, d.kv_dtype, d.out_dtype = _c.dt_code(q), _c.dt_code(k), _c.dt_code(out)
    d.p_dtype = _c.dt_code(p) if p is not None else 0
    d.mma_dtype, d.split = mma_code, 1 if split else 0
    d.scale = scale if scale is not None else float(dk) ** -0.5
    _c.check(_c.lib().cfm_attention(ctypes.byref(d), _c.stream()), "cfm_attention")
    return out


def kv_cache_pack(old_cache, k, v, k_str, v_str, B, H, Tn, dk):
    """-> f32 (B,H,Tc+Tn,2dk): rows < Tc copied from old_cache, the rest taken from k / v."""
    Tc = old_cache.size(2) if old_cache is not None else 0
    if old_cache is not None:
        if old_cache.dtype != torch.float32 or tuple(old_cache.shape) != (B, H, Tc, 2 * dk):
            raise ValueError("cfm.kv_cache_pack: cache must be float32 (B,H,Tc,2dk); got %s %s" % (old_cache.dtype, tuple(old_cache.shape)))
        old_cache = old_cache.contiguous()
    _c.require_hip(old_cache, k, v)
    out = torch.empty((B, H, Tc + Tn, 2 * dk), dtype=torch.float32, device=k.device)
    _c.check(_c.lib().cfm_kv_cache_pack(_c.ptr(old_cache), Tc, _c.ptr(k), _c.ptr(v), _c.dt_code(k), k_str[0], k_str[1], v_str[0],
                                        v_str[1], _c.ptr(out), B, H, Tn, dk, _c.stream()), "cfm_kv_cache_pack")
    return out


def dwconv_bn_silu(x, w, dw_bias, bn_scale, bn_shift, out=None, out_dtype=None):
    """x [B,T,D] channels-last -> silu(bn(depthwise(x)))."""
    _c.require_hip(x, w, dw_bias, bn_scale, bn_shift, out)
    if x.dim() != 3 or not x.is_contiguous():
        raise ValueError("cfm.dwconv_bn_silu: x must be contiguous [B,T,D]")
    B, T, D = x.shape
    if out is None:
        out = torch.empty((B, T, D), dtype=out_dtype or x.dtype, device=x.device)
    _c.check(_c.lib().cfm_dwconv_bn_silu(_c.ptr(x), _c.dt_code(x), _c.ptr(w), _c.ptr(dw_bias), _c.ptr(bn_scale), _c.ptr(bn_shift),
                                         _c.ptr(out), _c.dt_code(out), B, T, D, w.shape[1], _c.stream()), "cfm_dwconv_bn_silu")
    return out


def conv1_relu_mma_supported(C, out_dtype):
    return C % 16 == 0 and C <= 256 and out_dtype in (torch.bfloat16, torch.float16)


def conv1_relu(x, w9c, bias, out_dtype, cmvn=None, mma=False):
    """x [B,T,F] f32 -> relu(conv3x3 s2) channels-last [B,T1,F1,C].  cmvn = (mean [F], istd [F] | None): global CMVN folded into the
    tap loads, (x - mean) * istd, bit-identical to normalising first.  mma=True: the 9 taps as an MFMA contraction on operands rounded
    to out_dtype (16-bit; include/cfm.h cfm_conv1_relu_mma) instead of f32 FMAs."""
    _c.require_hip(x, w9c, bias)
    mean = istd = None
    if cmvn is not None:
        mean, istd = cmvn
        _c.require_hip(mean, istd)
        for t in (mean, istd):
            if t is not None and (t.dtype != torch.float32 or t.numel() != x.shape[2] or not t.is_contiguous()):
                raise ValueError("cfm.conv1_relu: cmvn statistics must be contiguous float32 [F]")
    if x.dim() != 3 or x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("cfm.conv1_relu: x must be contiguous float32 [B,T,F]")
    B, T, F = x.shape
    C = w9c.shape[1]
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    out = torch.empty((B, T1, F1, C), dtype=out_dtype, device=x.device)
    fn = _c.lib().cfm_conv1_relu_mma if mma else _c.lib().cfm_conv1_relu
    _c.check(fn(_c.ptr(x), _c.ptr(w9c), _c.ptr(bias), _c.ptr(out), _c.dt_code(out), B, T, F, C, _c.ptr(mean), _c.ptr(istd), _c.stream()),
             "cfm_conv1_relu_mma" if mma else "cfm_conv1_relu")
    return out


def ctc_nll(logits, V, enc_lens, labels, label_lens):
    """Per-utterance CTC negative log-likelihood from UN-normalised f32 logits [B,T,ld>=V] (log-softmax applied on the fly); blank 0.
    enc_lens / label_lens int32 [B], labels int32 [B,Umax].  Returns f32 [B] (include/cfm.h cfm_ctc_nll)."""
    _c.require_hip(logits, enc_lens, labels, label_lens)
    if logits.dim() != 3 or logits.dtype != torch.float32 or logits.stride(2) != 1 or logits.stride(0) != logits.size(1) * logits.stride(1):
        raise ValueError("cfm.ctc_nll: logits must be float32 [B,T,>=V] with contiguous rows")
    B, T = logits.shape[:2]
    for t in (enc_lens, labels, label_lens):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise ValueError("cfm.ctc_nll: lengths and labels must be contiguous int32")
    if labels.dim() != 2 or labels.size(0) != B or enc_lens.numel() != B or label_lens.numel() != B:
        raise ValueError("cfm.ctc_nll: batch sizes differ")
    out = torch.empty((B,), dtype=torch.float32, device=logits.device)
    work = scratch("ctc_lp", B * T * (2 * labels.size(1) + 1), torch.float32, logits.device)
    _c.check(_c.lib().cfm_ctc_nll(_c.ptr(logits), logits.stride(1), B, T, V, _c.ptr(enc_lens), _c.ptr(labels), labels.size(1), _c.ptr(label_lens),
                                  _c.ptr(work), _c.ptr(out), _c.stream()), "cfm_ctc_nll")
    return out


def joint_act(enc, pred, B, T, U, out_dtype):
    """tanh(enc[b*T+t] + pred[b*U+u]) for every (b,t,u) as a row-major [B*T*U, J] operand (include/cfm.h cfm_joint_act).
    enc f32 [B*T,J], pred f32 [B*U,J] with unit inner stride."""
    _c.require_hip(enc, pred)
    enc, pred = _rows2d(enc, "joint_act(enc)"), _rows2d(pred, "joint_act(pred)")
    J = enc.shape[1]
    if enc.dtype != torch.float32 or pred.dtype != torch.float32 or tuple(enc.shape) != (B * T, J) or tuple(pred.shape) != (B * U, J):
        raise ValueError("cfm.joint_act: enc must be f32 [%d,J] and pred f32 [%d,J], got %s %s" % (B * T, B * U, tuple(enc.shape), tuple(pred.shape)))
    out = torch.empty((B * T * U, J), dtype=out_dtype, device=enc.device)
    _c.check(_c.lib().cfm_joint_act(_c.ptr(enc), enc.stride(0), _c.ptr(pred), pred.stride(0), _c.ptr(out), _c.dt_code(out), B, T, U, J,
                                    _c.stream()), "cfm_joint_act")
    return out


def valid_mask(lengths, T, first=0, stride=1):
    """bool (B,T): (first + stride*t) < lengths[b]."""
    _c.require_hip(lengths)
    if lengths.dtype not in (torch.int32, torch.int64) or lengths.dim() != 1:
        raise ValueError("cfm.valid_mask: lengths must be a 1-D int32/int64 tensor")
    lengths = lengths.contiguous()
    B = lengths.numel()
    out = torch.empty((B, T), dtype=torch.uint8, device=lengths.device)
    _c.check(_c.lib().cfm_valid_mask(_c.ptr(lengths), 1 if lengths.dtype == torch.int64 else 0, _c.ptr(out), B, T, first, stride,
                                     _c.stream()), "cfm_valid_mask")
    return out.view(torch.bool)


def chunk_mask(size, chunk, left, device):
    out = torch.empty((size, size), dtype=torch.uint8, device=device)
    _c.require_hip(out)
    _c.check(_c.lib().cfm_chunk_mask(_c.ptr(out), size, chunk, left, _c.stream()), "cfm_chunk_mask")
    return out.view(torch.bool)


def attn_mask_combine(valid, chunk):
    """valid bool (B,1,T) [or (B,T)] & chunk bool (T,T) -> bool (B,T,T)."""
    v = as_u8_mask(valid.reshape(valid.size(0), -1))
    c = as_u8_mask(chunk)
    _c.require_hip(v, c)
    B, T = v.shape
    out = torch.empty((B, T, T), dtype=torch.uint8, device=v.device)
    _c.check(_c.lib().cfm_attn_mask(_c.ptr(v), _c.ptr(c), _c.ptr(out), B, T, _c.stream()), "cfm_attn_mask")
    return out.view(torch.bool)


def cast(src, dtype):
    _c.require_hip(src)
    src = src.contiguous()
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    if src.numel():
        _c.check(_c.lib().cfm_cast(_c.ptr(src), _c.dt_code(src), _c.ptr(out), _c.dt_code(out), src.numel(), _c.stream()), "cfm_cast")
    return out


def add_rows(x, add, group):
    """x[r,:] += add[r // group, :] in place (f32)."""
    _c.require_hip(x, add)
    rows, D = x.shape
    _c.check(_c.lib().cfm_add_rows(_c.ptr(x), _c.ptr(add), rows, D, group, _c.stream()), "cfm_add_rows")
    return x


# ----------------------------------------------------------------------------------------------------------------------
def prof_enable(on=True):
    _c.lib().cfm_prof_enable(1 if on else 0)


def prof_reset():
    _c.lib().cfm_prof_reset()


def prof_table():
    """Synchronise the recorded events; returns {kernel name: dict(calls, ms, flops, bytes)}."""
    L = _c.lib()
    n = L.cfm_prof_collect()
    out = {}
    name = ctypes.create_string_buffer(128)
    calls, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    for i in range(n):
        _c.check(L.cfm_prof_entry(i, name, 128, ctypes.byref(calls), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)), "cfm_prof_entry")
        out[name.value.decode()] = dict(calls=calls.value, ms=ms.value, flops=fl.value, bytes=by.value)
    return out
