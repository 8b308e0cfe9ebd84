"""Tensor-level wrappers over the C ABI.  Each function validates device / dtype / layout on the host,
allocates its output with torch.empty on the same device, and enqueues on the current HIP stream."""
import ctypes

import os

import torch

import cfm as _c

__all__ = ["stream_prep", "stream_advance", "dwconv_causal_bn_silu", "conv_cache_update", "dropout_rows", "dropout_mask", "set_deterministic", "gemm_tn", "gemm_tn_group", "layernorm_bwd", "glu_bwd", "dwconv_bn_train", "dwconv_bn_train_bwd", "col2im_relu_bwd", "conv1_wgrad", "attention_bwd",
           "ctc_nll_train", "ctc_nll_train_groups", "ctc_grad", "ffn_split", "adam_step", "adam_clip_step", "sumsq", "scratch_stats",
           "gemm", "ffn_fused", "ffn_fused_supported", "rowchain", "rowchain_supported", "rowchain_pair_supported", "layernorm", "attention", "kv_cache_pack", "dwconv_bn_silu", "conv1_relu", "conv1_relu_mma_supported", "conv12_relu", "conv12_supported", "ctc_nll", "joint_act", "valid_mask", "chunk_mask",
           "attn_mask_combine", "cast", "add_rows", "scratch", "prof_enable", "prof_reset", "prof_table", "as_u8_mask"]


def _rows2d(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("cfm.%s: expected a 2-D tensor with unit inner stride, got shape %s strides %s" % (name, tuple(t.shape), t.stride()))
    return t


# A grow-only arena of reusable device buffers (workspace owned by the extension side of the boundary), keyed per
# (device, STREAM): two streams never share a scratch buffer, so forwards running concurrently on one device cannot overwrite each
# other's xn / hid / qkv.  A buffer that has been handed out is NEVER released: when a larger one is needed the old block is retired,
# not freed -- a captured HIP graph (bench.py, StreamingSession) may still hold its address.  Blocks grow by at least 1.5x, so the
# retired list stays short.
_arena = {}
_retired = []


def scratch(tag, numel, dtype, device, zero=False):
    """zero=True: the block is zero-filled when it is (re)allocated -- for buffers whose never-written parts are read (and must be finite)."""
    device = torch.device(device)
    key = (tag, dtype, device, torch.cuda.current_stream(device).cuda_stream)
    buf = _arena.get(key)
    if buf is None or buf.numel() < numel:
        want = max(int(numel), 1) if buf is None else max(int(numel), buf.numel() * 3 // 2)
        if buf is not None:
            _retired.append(buf)
        buf = (torch.zeros if zero else torch.empty)(want, dtype=dtype, device=device)
        _arena[key] = buf
    return buf[:numel]


def scratch_stats():
    """(live blocks, retired blocks, bytes held) -- for tests and leak hunting."""
    held = sum(b.numel() * b.element_size() for b in list(_arena.values()) + _retired)
    return len(_arena), len(_retired), held


def gemm(a, w, bias=None, w_lo=None, out=None, out_dtype=None, act=_c.ACT_NONE, residual=None, alpha=1.0, row_mask=None,
         mask_mode=0, conv=None, tile=0, n_out=None, pre_out=None, aux=None, drop=None, drop2=None):
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
    if pre_out is not None:                      # training: acc + bias before the activation, all N columns
        _c.require_hip(pre_out)
        pre_out = _rows2d(pre_out, "gemm(pre_out)")
        if tuple(pre_out.shape) != (M, N):
            raise ValueError("cfm.gemm: pre_out is %s, expected (%d,%d)" % (tuple(pre_out.shape), M, N))
        d.C_pre, d.ld_pre, d.pre_dtype = _c.ptr(pre_out), pre_out.stride(0), _c.dt_code(pre_out)
    if aux is not None:                          # backward epilogues (ACT_DSILU / ACT_DRELU): the forward pre-activation / output
        _c.require_hip(aux)
        aux = _rows2d(aux, "gemm(aux)")
        if tuple(aux.shape) != (M, N):
            raise ValueError("cfm.gemm: aux is %s, expected (%d,%d)" % (tuple(aux.shape), M, N))
        d.aux, d.ld_aux, d.aux_dtype = _c.ptr(aux), aux.stride(0), _c.dt_code(aux)
    if drop is not None and drop[0] > 0.0:       # (p, seed): output dropout in the epilogue (train mode)
        d.drop_p, d.drop_seed = float(drop[0]), int(drop[1]) & 0xFFFFFFFF
        if drop2 is not None and drop2[0] > 0.0:
            d.drop2_p, d.drop2_seed = float(drop2[0]), int(drop2[1]) & 0xFFFFFFFF
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


def rowchain_pair_supported(D, FF, prec):
    """The feed-forward split over workgroup pairs (cfm_rowchain_desc.psum_out / psum_in, D = 512)."""
    return (not prec.split) and bool(_c.lib().cfm_rowchain_pair_supported(D, FF))


def ffn_split(x, w_code, mode, psum=None, psum_b2=None, psum_alpha=1.0, ln1=None, ln2=None, rows_out=None, rows2_out=None, ln=None, w1=None, b1=None,
              n1=0, act=_c.ACT_NONE, w2=None, psum_out=None, out16=None, eps=1e-5, ring=None):
    """include/cfm.h cfm_ffn_split: the rows stage (optional reduce of partial slabs psum [G,M,D] + residual x, LN1, rows_out) followed by nothing
    (mode 0: + LN2 -> rows2_out), a projection (mode 1: out16 [M,n1]) or a feed-forward that leaves partial slabs (mode 2: psum_out [n1/256,M,D])."""
    _c.require_hip(x, psum, psum_b2, rows_out, rows2_out, w1, b1, w2, psum_out, out16)
    x = _rows2d(x, "ffn_split(x)")
    M, D = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("cfm.ffn_split: x must be contiguous float32 rows")
    d = _c.FfnSplitDesc()
    d.x, d.M, d.D, d.mode, d.w_dtype, d.eps = x.data_ptr(), M, D, mode, w_code, eps
    if psum is not None:
        if psum.dtype != torch.float32 or psum.dim() != 3 or tuple(psum.shape[1:]) != (M, D) or not psum.is_contiguous():
            raise ValueError("cfm.ffn_split: psum must be contiguous float32 [G, M, D]")
        d.psum, d.psum_b2, d.psum_splits, d.psum_alpha = psum.data_ptr(), _c.ptr(psum_b2), psum.shape[0], psum_alpha
    for name, pair in (("ln1", ln1), ("ln2", ln2), ("ln", ln)):
        if pair is not None:
            setattr(d, name + "_g", pair[0].data_ptr())
            setattr(d, name + "_b", pair[1].data_ptr())
    d.rows_out, d.rows2_out = _c.ptr(rows_out), _c.ptr(rows2_out)
    d.w1, d.b1, d.N1, d.act, d.w2, d.psum_out = _c.ptr(w1), _c.ptr(b1), n1, act, _c.ptr(w2), _c.ptr(psum_out)
    if out16 is not None:
        d.out16, d.ldo = out16.data_ptr(), out16.stride(0)
    if ring is not None:                                  # (kv f32 [B,H,ring_T,2dk], offsets int32 [B], frames per stream): mode 1 as the q|k|v projection
        kv, offs, tq = ring
        _c.require_hip(kv, offs)
        d.kv_ring, d.ring_offsets, d.ring_T, d.ring_H, d.ring_Tq = kv.data_ptr(), offs.data_ptr(), kv.shape[2], kv.shape[1], tq
    _c.check(_c.lib().cfm_ffn_split(ctypes.byref(d), _c.stream()), "cfm_ffn_split")


def rowchain(M, D, w_code, x=None, head=None, ln=None, ln_mask=None, ffn=None, alpha=1.0, ln1=None, ln2=None, out_f32=None, out16=None,
             tail=None, eps=1e-5, dw=None):
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
              mask_str=(0, 0), mma_code=_c.BF16, split=False, scale=None, lse=None, drop=None):
    """Fused attention; *_str are (batch stride, time stride[, head stride]) in ELEMENTS (see include/cfm.h).  lse: optional f32 [B,H,Tq]
    output (training): the log-sum-exp of each row's scaled masked scores."""
    _c.require_hip(q, k, v, p, out, mask, bias_u, bias_v, lse)
    if lse is not None and (lse.dtype != torch.float32 or lse.numel() != B * H * Tq or not lse.is_contiguous()):
        raise ValueError("cfm.attention: lse must be contiguous float32 [B,H,Tq]")
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
    d.lse = _c.ptr(lse)
    if drop is not None and drop[0] > 0.0:       # (p, seed): dropout on the probabilities (train mode)
        d.drop_p, d.drop_seed = float(drop[0]), int(drop[1]) & 0xFFFFFFFF
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


def conv12_supported(C, out_dtype):
    return out_dtype in (torch.bfloat16, torch.float16) and bool(_c.lib().cfm_conv12_supported(C, _c.dt_code(out_dtype)))


def conv12_relu(x, w9c, b1, w2, b2, cmvn=None):
    """x [B,T,F] f32 -> relu(conv3x3 s2(relu(conv3x3 s2(x)))) as [B*T2*F2, C] in w2's 16-bit dtype, the first convolution recomputed inside
    the second one's operand producer (include/cfm.h cfm_conv12_relu; bit-identical to conv1_relu(mma=True) + gemm(conv=...))."""
    _c.require_hip(x, w9c, b1, w2, b2)
    mean = istd = None
    if cmvn is not None:
        mean, istd = cmvn
        _c.require_hip(mean, istd)
        for t in (mean, istd):
            if t is not None and (t.dtype != torch.float32 or t.numel() != x.shape[2] or not t.is_contiguous()):
                raise ValueError("cfm.conv12_relu: cmvn statistics must be contiguous float32 [F]")
    if x.dim() != 3 or x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("cfm.conv12_relu: x must be contiguous float32 [B,T,F]")
    B, T, F = x.shape
    C = w9c.shape[1]
    if w2.shape != (C, 9 * C) or not w2.is_contiguous() or w2.dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("cfm.conv12_relu: w2 must be a contiguous 16-bit [C, 9C] matrix")
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    out = torch.empty((B * T2 * F2, C), dtype=w2.dtype, device=x.device)
    _c.check(_c.lib().cfm_conv12_relu(_c.ptr(x), _c.ptr(w9c), _c.ptr(b1), _c.ptr(w2), _c.ptr(b2), _c.ptr(out), _c.dt_code(out), B, T, F, C,
                                      _c.ptr(mean), _c.ptr(istd), _c.stream()), "cfm_conv12_relu")
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
    work = scratch("ctc_lp", B * T * (2 * labels.size(1) + 2), torch.float32, logits.device)
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


# ----------------------------------------------------------------------------------------------------------------------
# training (include/cfm.h "Training"): thin wrappers, outputs allocated here, workspaces from the per-stream arena
# ----------------------------------------------------------------------------------------------------------------------
_deterministic = [False]


def set_deterministic(on=True):
    """Bitwise-reproducible gradients: the weight-gradient GEMM then runs ONE workgroup per output tile over all M rows (no split-M
    atomics, whose arrival order varies from run to run).  Slower on small batches; for tests and debugging."""
    _deterministic[0] = bool(on)


def gemm_tn(a, b, out=None, want_colsum=False, row_mask=None, alpha=1.0, conv=None, split=False, splits=0, mma_code=_c.BF16, accumulate=False,
            colsum=None):
    """C[N,K] (+)= alpha * a[M,N]^T . b[M,K]  (f32) and optionally colsum[N] = alpha * sum_m a[m,:]: the weight / bias gradient of a
    dense layer from its output gradient `a` and its input rows `b` (include/cfm.h cfm_gemm_tn).  conv=(C,T1,F1,T2,F2): `b` is a
    channels-last image [B,T1,F1,C] read as its 3x3 stride-2 im2col matrix.  Returns (C, colsum | None)."""
    _c.require_hip(a, b, out, row_mask, colsum)
    a = _rows2d(a, "gemm_tn(a)")
    M, N = a.shape
    d = _c.GemmTnDesc()
    if conv is None:
        b = _rows2d(b, "gemm_tn(b)")
        if b.shape[0] != M:
            raise ValueError("cfm.gemm_tn: a is %s but b is %s" % (tuple(a.shape), tuple(b.shape)))
        K = b.shape[1]
        d.ldb = b.stride(0)
    else:
        C, T1, F1, T2, F2 = conv
        if not b.is_contiguous() or b.numel() != (M // (T2 * F2)) * T1 * F1 * C:
            raise ValueError("cfm.gemm_tn(conv): image must be contiguous channels-last [B,T1,F1,C]")
        K = 9 * C
        d.conv_C, d.conv_T1, d.conv_F1, d.conv_T2, d.conv_F2 = C, T1, F1, T2, F2
    if out is None:
        if accumulate:
            raise ValueError("cfm.gemm_tn: accumulate needs an existing out")
        if want_colsum and colsum is None and not _deterministic[0]:
            # weight and bias gradient in ONE zero-filled buffer (one fill launch instead of the library's two memsets), then accumulate
            buf = torch.zeros((N * K + N,), dtype=torch.float32, device=a.device)
            out, colsum, accumulate = buf[:N * K].view(N, K), buf[N * K:], True
        else:
            out = torch.empty((N, K), dtype=torch.float32, device=a.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (N, K) or out.stride(1) != 1:
        raise ValueError("cfm.gemm_tn: out must be float32 (%d,%d)" % (N, K))
    if want_colsum and colsum is None:
        colsum = torch.empty((N,), dtype=torch.float32, device=a.device)
    if colsum is not None and (colsum.dtype != torch.float32 or colsum.numel() != N or not colsum.is_contiguous()):
        raise ValueError("cfm.gemm_tn: colsum must be contiguous float32 [N]")
    if row_mask is not None and (row_mask.dtype != torch.uint8 or row_mask.numel() != M or not row_mask.is_contiguous()):
        raise ValueError("cfm.gemm_tn: row_mask must be contiguous uint8 [M]")
    d.A, d.B, d.C, d.colsum, d.row_mask = _c.ptr(a), _c.ptr(b), _c.ptr(out), _c.ptr(colsum), _c.ptr(row_mask)
    d.lda, d.ldc = a.stride(0), out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.a_dtype, d.b_dtype, d.mma_dtype = _c.dt_code(a), _c.dt_code(b), mma_code
    d.split, d.accumulate, d.splits, d.alpha = 1 if split else 0, 1 if accumulate else 0, (1 if _deterministic[0] else splits), alpha
    _c.check(_c.lib().cfm_gemm_tn(ctypes.byref(d), _c.stream()), "cfm_gemm_tn")
    return out, colsum


def gemm_tn_group(products, mma_code=_c.BF16, splits=0):
    """Several weight-gradient products in ONE launch (include/cfm.h cfm_gemm_tn_group).  products: list of dicts with keys a [M,N], b [M,K],
    out f32 [N,K] (zero-filled or holding a running sum: the products ACCUMULATE), optional colsum f32 [N], alpha, row_off / colsum_off /
    colsum_off2 (int64 device tables; `out` / `colsum` are then the slab the offsets are relative to).  Returns nothing."""
    n = len(products)
    descs = (_c.GemmTnDesc * n)()
    for d, pr in zip(descs, products):
        a, b, out = _rows2d(pr["a"], "gemm_tn_group(a)"), _rows2d(pr["b"], "gemm_tn_group(b)"), pr["out"]
        colsum = pr.get("colsum")
        _c.require_hip(a, b, out, colsum)
        if a.shape[0] != b.shape[0] or out.dtype != torch.float32 or (colsum is not None and colsum.dtype != torch.float32):
            raise ValueError("cfm.gemm_tn_group: a %s, b %s, out %s" % (tuple(a.shape), tuple(b.shape), out.dtype))
        d.A, d.B, d.C, d.colsum = a.data_ptr(), b.data_ptr(), out.data_ptr(), _c.ptr(colsum)
        d.lda, d.ldb = a.stride(0), b.stride(0)
        d.M, d.N, d.K = a.shape[0], a.shape[1], b.shape[1]
        d.ldc = pr.get("ldc", d.K)
        d.a_dtype, d.b_dtype, d.mma_dtype = _c.dt_code(a), _c.dt_code(b), mma_code
        d.accumulate, d.splits, d.alpha = 1, (1 if _deterministic[0] else splits), float(pr.get("alpha", 1.0))
        d.row_off, d.colsum_off, d.colsum_off2 = _c.ptr(pr.get("row_off")), _c.ptr(pr.get("colsum_off")), _c.ptr(pr.get("colsum_off2"))
    _c.check(_c.lib().cfm_gemm_tn_group(descs, n, _c.stream()), "cfm_gemm_tn_group")


def layernorm_bwd(x, dy, gamma, row_mask=None, dres=None, dx=None, eps=1e-5):
    """-> (dx, dgamma, dbeta): dx = (dres or 0) + dLayerNorm(dy) w.r.t. the norm's input x (f32 [M,D]); dx may be dres itself."""
    _c.require_hip(x, dy, gamma, row_mask, dres, dx)
    x = _rows2d(x, "layernorm_bwd(x)")
    M, D = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous() or tuple(dy.shape) != (M, D) or not dy.is_contiguous():
        raise ValueError("cfm.layernorm_bwd: x must be contiguous float32 [M,D] and dy contiguous [M,D]")
    if dx is None:
        dx = torch.empty_like(x)
    for t in (dres, dx):
        if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != (M, D) or not t.is_contiguous()):
            raise ValueError("cfm.layernorm_bwd: dres / dx must be contiguous float32 [M,D]")
    dg = torch.empty((D,), dtype=torch.float32, device=x.device)
    db = torch.empty((D,), dtype=torch.float32, device=x.device)
    ws = scratch("ln_bwd", _c.lib().cfm_layernorm_bwd_ws(M, D), torch.float32, x.device)
    _c.check(_c.lib().cfm_layernorm_bwd(_c.ptr(x), _c.ptr(dy), _c.dt_code(dy), _c.ptr(gamma), _c.ptr(row_mask), _c.ptr(dres), _c.ptr(dx), _c.ptr(dg),
                                        _c.ptr(db), _c.ptr(ws), eps, M, D, _c.stream()), "cfm_layernorm_bwd")
    return dx, dg, db


def glu_bwd(u, dg, out_dtype):
    """u [M,2D] (the interleaved pre-GLU columns from gemm(..., pre_out=)), dg [M,D] -> du [M,2D] same layout."""
    _c.require_hip(u, dg)
    M, D2 = u.shape
    D = D2 // 2
    if not u.is_contiguous() or not dg.is_contiguous() or tuple(dg.shape) != (M, D):
        raise ValueError("cfm.glu_bwd: u must be contiguous [M,2D] and dg contiguous [M,D]")
    du = torch.empty((M, D2), dtype=out_dtype, device=u.device)
    _c.check(_c.lib().cfm_glu_bwd(_c.ptr(u), _c.dt_code(u), _c.ptr(dg), _c.dt_code(dg), _c.ptr(du), _c.dt_code(du), M, D, _c.stream()), "cfm_glu_bwd")
    return du


def dwconv_bn_train(g, w, dw_bias, gamma, beta, running_mean, running_var, momentum, eps, s_dtype):
    """g [B,T,D] -> (c f32 [B,T,D], stats f32 [4,D], s = SiLU(BatchNorm_train(c)) in s_dtype); running statistics updated in place."""
    _c.require_hip(g, w, dw_bias, gamma, beta, running_mean, running_var)
    if g.dim() != 3 or not g.is_contiguous():
        raise ValueError("cfm.dwconv_bn_train: g must be contiguous [B,T,D]")
    B, T, D = g.shape
    c = torch.empty((B, T, D), dtype=torch.float32, device=g.device)
    stats = torch.empty((4, D), dtype=torch.float32, device=g.device)
    s = torch.empty((B, T, D), dtype=s_dtype, device=g.device)
    ws = scratch("dwbn", _c.lib().cfm_dwconv_bn_ws(B, T, D), torch.float32, g.device)
    _c.check(_c.lib().cfm_dwconv_bn_train(_c.ptr(g), _c.dt_code(g), _c.ptr(w), _c.ptr(dw_bias), _c.ptr(gamma), _c.ptr(beta), _c.ptr(running_mean),
                                          _c.ptr(running_var), momentum, eps, _c.ptr(c), _c.ptr(stats), _c.ptr(s), _c.dt_code(s), _c.ptr(ws), B, T, D,
                                          w.shape[1], _c.stream()), "cfm_dwconv_bn_train")
    return c, stats, s


def dwconv_bn_train_bwd(ds, c, stats, g, w, dg_dtype):
    """-> (dg [B,T,D], d taps [D,K], d conv bias [D], d BatchNorm gain [D], d BatchNorm bias [D])."""
    _c.require_hip(ds, c, stats, g, w)
    B, T, D = g.shape
    if not ds.is_contiguous() or ds.numel() != B * T * D or not c.is_contiguous() or not g.is_contiguous():
        raise ValueError("cfm.dwconv_bn_train_bwd: ds, c, g must be contiguous [B,T,D]")
    dev = g.device
    dg = torch.empty((B, T, D), dtype=dg_dtype, device=dev)
    dw_w = torch.empty((D, w.shape[1]), dtype=torch.float32, device=dev)
    dw_b, dgamma, dbeta = (torch.empty((D,), dtype=torch.float32, device=dev) for _ in range(3))
    dy = scratch("dwbn_dy", B * T * D, torch.float32, dev)
    ws = scratch("dwbn", _c.lib().cfm_dwconv_bn_ws(B, T, D), torch.float32, dev)
    _c.check(_c.lib().cfm_dwconv_bn_train_bwd(_c.ptr(ds), _c.dt_code(ds), _c.ptr(c), _c.ptr(stats), _c.ptr(g), _c.dt_code(g), _c.ptr(w), _c.ptr(dg),
                                              _c.dt_code(dg), _c.ptr(dw_w), _c.ptr(dw_b), _c.ptr(dgamma), _c.ptr(dbeta), _c.ptr(dy), _c.ptr(ws), B, T, D,
                                              w.shape[1], _c.stream()), "cfm_dwconv_bn_train_bwd")
    return dg, dw_w, dw_b, dgamma, dbeta


def col2im_relu_bwd(dcol, h1, out_dtype):
    """dcol [B*T2*F2, 9C] (K order (kt,kf,c)), h1 [B,T1,F1,C] -> dh1 = (h1 > 0) * col2im(dcol)."""
    _c.require_hip(dcol, h1)
    B, T1, F1, C = h1.shape
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    if not dcol.is_contiguous() or tuple(dcol.shape) != (B * T2 * F2, 9 * C) or not h1.is_contiguous():
        raise ValueError("cfm.col2im_relu_bwd: dcol must be contiguous [B*T2*F2, 9C] and h1 contiguous [B,T1,F1,C]")
    dh1 = torch.empty((B, T1, F1, C), dtype=out_dtype, device=h1.device)
    _c.check(_c.lib().cfm_col2im_relu_bwd(_c.ptr(dcol), _c.dt_code(dcol), _c.ptr(h1), _c.dt_code(h1), _c.ptr(dh1), _c.dt_code(dh1), B, T1, F1, C,
                                          _c.stream()), "cfm_col2im_relu_bwd")
    return dh1


def conv1_wgrad(dh1, x, cmvn=None):
    """dh1 [B,T1,F1,C], x f32 [B,T,F] -> (dw [9,C] tap-major, db [C])."""
    _c.require_hip(dh1, x)
    B, T, F = x.shape
    C = dh1.shape[3]
    mean, istd = cmvn if cmvn is not None else (None, None)
    _c.require_hip(mean, istd)
    if x.dtype != torch.float32 or not x.is_contiguous() or not dh1.is_contiguous() or tuple(dh1.shape[:3]) != (B, (T - 3) // 2 + 1, (F - 3) // 2 + 1):
        raise ValueError("cfm.conv1_wgrad: x must be contiguous float32 [B,T,F] and dh1 contiguous [B,T1,F1,C]")
    dw = torch.empty((9, C), dtype=torch.float32, device=x.device)
    db = torch.empty((C,), dtype=torch.float32, device=x.device)
    ws = scratch("conv1_wgrad", _c.lib().cfm_conv1_wgrad_ws(B, T, C), torch.float32, x.device)
    _c.check(_c.lib().cfm_conv1_wgrad(_c.ptr(dh1), _c.dt_code(dh1), _c.ptr(x), _c.ptr(mean), _c.ptr(istd), _c.ptr(dw), _c.ptr(db), _c.ptr(ws), B, T, F, C,
                                      _c.stream()), "cfm_conv1_wgrad")
    return dw, db


def attention_bwd(q, k, v, out, dout, lse, B, H, Tq, Tk, dk, q_str, k_str, v_str, dq, dkk, dv, mask=None, mask_str=(0, 0), mma_code=_c.BF16, split=False,
                  scale=None, drop=None):
    """Backward of attention(); q/k/v and dq/dkk/dv share strides ((batch, time) in elements, head h at h*dk); see include/cfm.h."""
    _c.require_hip(q, k, v, out, dout, lse, dq, dkk, dv, mask)
    d = _c.AttnBwdDesc()
    d.q, d.k, d.v, d.mask, d.out, d.dout, d.lse = _c.ptr(q), _c.ptr(k), _c.ptr(v), _c.ptr(mask), _c.ptr(out), _c.ptr(dout), _c.ptr(lse)
    d.grad_q, d.grad_k, d.grad_v = _c.ptr(dq), _c.ptr(dkk), _c.ptr(dv)
    delta = scratch("attn_delta", B * H * Tq, torch.float32, q.device)
    d.delta = _c.ptr(delta)
    d.q_sb, d.q_st = q_str
    d.k_sb, d.k_st = k_str
    d.v_sb, d.v_st = v_str
    d.m_sb, d.m_sq = mask_str
    d.B, d.H, d.Tq, d.Tk, d.dk = B, H, Tq, Tk, dk
    if not (q.dtype == k.dtype == v.dtype == out.dtype == dq.dtype == dkk.dtype == dv.dtype):
        raise ValueError("cfm.attention_bwd: q, k, v, out and the gradients must share one dtype")
    d.io_dtype, d.dout_dtype, d.mma_dtype, d.split = _c.dt_code(q), _c.dt_code(dout), mma_code, 1 if split else 0
    d.scale = scale if scale is not None else float(dk) ** -0.5
    if drop is not None and drop[0] > 0.0:
        d.drop_p, d.drop_seed = float(drop[0]), int(drop[1]) & 0xFFFFFFFF
    _c.check(_c.lib().cfm_attention_bwd(ctypes.byref(d), _c.stream()), "cfm_attention_bwd")


def _ctc_args(logits, enc_lens, labels, label_lens):
    if logits.dim() != 3 or logits.dtype != torch.float32 or logits.stride(2) != 1 or logits.stride(0) != logits.size(1) * logits.stride(1):
        raise ValueError("cfm.ctc: logits must be float32 [B,T,>=V] with contiguous rows")
    for t in (enc_lens, labels, label_lens):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise ValueError("cfm.ctc: lengths and labels must be contiguous int32")
    B = logits.shape[0]
    if labels.dim() != 2 or labels.size(0) != B or enc_lens.numel() != B or label_lens.numel() != B:
        raise ValueError("cfm.ctc: batch sizes differ")


def ctc_nll_train(logits, V, enc_lens, labels, label_lens, beta_now=None):
    """As ctc_nll, keeping what the backward needs: returns (nll [B], state) with state = (work, alpha, lse, nll_shifted, beta | None) owned by the
    caller.  beta_now: run the backward recursion beside the forward one (one launch of 2 B workgroups) instead of inside ctc_grad."""
    _c.require_hip(logits, enc_lens, labels, label_lens)
    _ctc_args(logits, enc_lens, labels, label_lens)
    if beta_now is None:
        beta_now = os.environ.get("CFM_CTC_BETA_NOW", "1") != "0"        # A/B switch (same bits either way)
    B, T = logits.shape[:2]
    SM = 2 * labels.size(1) + 2
    dev = logits.device
    nll = torch.empty((B,), dtype=torch.float32, device=dev)
    nllp = torch.empty((B,), dtype=torch.float32, device=dev)
    work = torch.empty((B, T, SM), dtype=torch.float32, device=dev)
    alpha = torch.empty((B, T, SM), dtype=torch.float32, device=dev)
    lse = torch.empty((B, T), dtype=torch.float32, device=dev)
    beta = torch.empty((B, T, SM), dtype=torch.float32, device=dev) if beta_now else None
    _c.check(_c.lib().cfm_ctc_nll_train(_c.ptr(logits), logits.stride(1), B, T, V, _c.ptr(enc_lens), _c.ptr(labels), labels.size(1), _c.ptr(label_lens),
                                        _c.ptr(work), _c.ptr(alpha), _c.ptr(lse), _c.ptr(nll), _c.ptr(nllp), _c.ptr(beta), _c.stream()), "cfm_ctc_nll_train")
    return nll, (work, alpha, lse, nllp, beta)


def ctc_nll_train_groups(problems, V):
    """ctc_nll_train for several micro-batches with ONE launch for all their recursions (include/cfm.h cfm_ctc_nll_train_groups).
    problems: [(logits [B,T,ld], enc_lens, labels, label_lens)]; returns [(nll [B], state)] as ctc_nll_train does."""
    n = len(problems)
    arr = (_c.CtcGroup * n)()
    outs = []
    for g, (logits, enc_lens, labels, label_lens) in zip(arr, problems):
        _c.require_hip(logits, enc_lens, labels, label_lens)
        _ctc_args(logits, enc_lens, labels, label_lens)
        B, T = logits.shape[:2]
        SM = 2 * labels.size(1) + 2
        dev = logits.device
        nll, nllp = torch.empty((B,), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.float32, device=dev)
        work, alpha, beta = (torch.empty((B, T, SM), dtype=torch.float32, device=dev) for _ in range(3))
        lse = torch.empty((B, T), dtype=torch.float32, device=dev)
        g.logits, g.ld, g.B, g.T, g.Umax = logits.data_ptr(), logits.stride(1), B, T, labels.size(1)
        g.enc_lens, g.labels, g.label_lens = enc_lens.data_ptr(), labels.data_ptr(), label_lens.data_ptr()
        g.work, g.alpha, g.lse, g.nll, g.nll_shifted, g.beta = work.data_ptr(), alpha.data_ptr(), lse.data_ptr(), nll.data_ptr(), nllp.data_ptr(), beta.data_ptr()
        outs.append((nll, (work, alpha, lse, nllp, beta)))
    _c.check(_c.lib().cfm_ctc_nll_train_groups(arr, n, V, _c.stream()), "cfm_ctc_nll_train_groups")
    return outs


def ctc_grad(logits, V, enc_lens, labels, label_lens, state, gscale=1.0, gscale_dev=None, out=None):
    """d (sum_b nll_b) / d logits * gscale * (*gscale_dev), f32 [B,T,ld]; consumes state[1] (alpha is overwritten unless the state carries beta)."""
    _c.require_hip(logits, enc_lens, labels, label_lens, gscale_dev, out)
    _ctc_args(logits, enc_lens, labels, label_lens)
    work, alpha, lse, nllp = state[:4]
    beta = state[4] if len(state) > 4 else None
    B, T = logits.shape[:2]
    if out is None:
        out = torch.empty_like(logits)
    if out.dtype != torch.float32 or out.shape != logits.shape or out.stride() != logits.stride():
        raise ValueError("cfm.ctc_grad: out must match logits")
    _c.check(_c.lib().cfm_ctc_grad(_c.ptr(logits), logits.stride(1), B, T, V, _c.ptr(enc_lens), _c.ptr(labels), labels.size(1), _c.ptr(label_lens),
                                   _c.ptr(work), _c.ptr(alpha), _c.ptr(beta), _c.ptr(lse), _c.ptr(nllp), gscale, _c.ptr(gscale_dev), _c.ptr(out), _c.stream()), "cfm_ctc_grad")
    return out


def adam_step(p, g, m, v, lr, betas, eps, weight_decay, step, grad_scale=None):
    """In-place Adam over flat float32 buffers (torch.optim.Adam's update rule); grad_scale: optional device scalar multiplied into g."""
    _c.require_hip(p, g, m, v, grad_scale)
    n = p.numel()
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
            raise ValueError("cfm.adam_step: p, g, m, v must be contiguous float32 buffers of one size")
    _c.check(_c.lib().cfm_adam_step(_c.ptr(p), _c.ptr(g), _c.ptr(m), _c.ptr(v), n, lr, betas[0], betas[1], eps, weight_decay, step, _c.ptr(grad_scale),
                                    _c.stream()), "cfm_adam_step")


def adam_clip_step(p, g, m, v, lr, betas, eps, weight_decay, step, sumsq_t, clip, inv_world, zero_grad=True):
    """adam_step with the clip coefficient computed on the device from `sumsq_t` (cfm.sumsq of g), the 1/world averaging and the zeroing of g in
    the same launch (include/cfm.h cfm_adam_clip_step).  Returns the averaged gradient's norm as a 1-element device tensor."""
    _c.require_hip(p, g, m, v, sumsq_t)
    n = p.numel()
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
            raise ValueError("cfm.adam_clip_step: p, g, m, v must be contiguous float32 buffers of one size")
    norm = torch.empty((1,), dtype=torch.float32, device=p.device)
    _c.check(_c.lib().cfm_adam_clip_step(_c.ptr(p), _c.ptr(g), _c.ptr(m), _c.ptr(v), n, lr, betas[0], betas[1], eps, weight_decay, step, _c.ptr(sumsq_t),
                                         float(clip or 0.0), float(inv_world), 1 if zero_grad else 0, _c.ptr(norm), _c.stream()), "cfm_adam_clip_step")
    return norm


def sumsq(x):
    """sum(x^2) of a flat float32 buffer as a 1-element device tensor (no host sync)."""
    _c.require_hip(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("cfm.sumsq: x must be a contiguous float32 buffer")
    n = x.numel()
    nb = max(1, min(1024, (n + 4095) // 4096))
    part = scratch("sumsq", nb, torch.float32, x.device)
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    _c.check(_c.lib().cfm_sumsq(_c.ptr(x), n, _c.ptr(part), nb, _c.ptr(out), _c.stream()), "cfm_sumsq")
    return out


def dropout_rows(x, out_dtype, alpha=1.0, drop=None, drop2=None, row_mask=None):
    """y = alpha * x * keep(seed, element) / (1 - p), rows with row_mask == 0 zeroed: the gradient of a residual branch
    x + alpha * dropout(f) as a GEMM operand (include/cfm.h cfm_dropout_rows).  drop / drop2 = (p, seed) or None."""
    _c.require_hip(x, row_mask)
    x = _rows2d(x, "dropout_rows(x)")
    if not x.is_contiguous():
        raise ValueError("cfm.dropout_rows: x must be contiguous")
    M, N = x.shape
    y = torch.empty((M, N), dtype=out_dtype, device=x.device)
    p1, s1 = (float(drop[0]), int(drop[1]) & 0xFFFFFFFF) if drop is not None else (0.0, 0)
    p2, s2 = (float(drop2[0]), int(drop2[1]) & 0xFFFFFFFF) if drop2 is not None else (0.0, 0)
    _c.check(_c.lib().cfm_dropout_rows(_c.ptr(x), _c.dt_code(x), _c.ptr(y), _c.dt_code(y), _c.ptr(row_mask), alpha, p1, s1, p2, s2, M, N, _c.stream()),
             "cfm_dropout_rows")
    return y


def dropout_mask(n, p, seed, device):
    """the 0/1 keep mask the kernels regenerate for (p, seed) over element indices 0..n-1 (bool tensor; for tests)."""
    out = torch.empty((n,), dtype=torch.uint8, device=device)
    _c.require_hip(out)
    _c.check(_c.lib().cfm_dropout_mask(_c.ptr(out), n, float(p), int(seed) & 0xFFFFFFFF, _c.stream()), "cfm_dropout_mask")
    return out.view(torch.bool)


# ----------------------------------------------------------------------------------------------------------------------
# per-stream streaming state (include/cfm.h, csrc/stream.hip)
# ----------------------------------------------------------------------------------------------------------------------
def stream_prep(offsets, T, need, ring_T, pe, slot_mask, pos_rows, abs_rows=None):
    """offsets int32 [B] -> slot_mask u8 [B,ring_T], pos_rows f32 [B,ring_T,D] (= pe[frame held by the slot]), abs_rows f32 [B,D] = pe[offset]."""
    _c.require_hip(offsets, pe, slot_mask, pos_rows, abs_rows)
    B = offsets.numel()
    D = pe.shape[-1]
    if offsets.dtype != torch.int32 or pe.dtype != torch.float32 or not pe.is_contiguous() or slot_mask.numel() != B * ring_T or pos_rows.numel() != B * ring_T * D:
        raise ValueError("cfm.stream_prep: offsets int32 [B], pe contiguous f32 [max_len,D], slot_mask [B,ring_T], pos_rows [B,ring_T,D]")
    _c.check(_c.lib().cfm_stream_prep(_c.ptr(offsets), B, T, need, ring_T, _c.ptr(pe), pe.numel() // D, D, _c.ptr(slot_mask), _c.ptr(pos_rows), _c.ptr(abs_rows),
                                      _c.stream()), "cfm_stream_prep")


def stream_advance(offsets, T, active=None):
    _c.require_hip(offsets, active)
    _c.check(_c.lib().cfm_stream_advance(_c.ptr(offsets), _c.ptr(active), offsets.numel(), T, _c.stream()), "cfm_stream_advance")


def dwconv_causal_bn_silu(x, w, dw_bias, bn_scale, bn_shift, cache=None, out_dtype=None):
    """OPT-IN causal depthwise conv + folded BatchNorm + SiLU over [cache | x]; x [B,T,D], cache f32 [B,K-1,D] or None (zeros)."""
    _c.require_hip(x, w, dw_bias, bn_scale, bn_shift, cache)
    B, T, D = x.shape
    K = w.shape[1]
    if not x.is_contiguous() or (cache is not None and (cache.dtype != torch.float32 or tuple(cache.shape) != (B, K - 1, D) or not cache.is_contiguous())):
        raise ValueError("cfm.dwconv_causal_bn_silu: x contiguous [B,T,D], cache contiguous f32 [B,K-1,D]")
    y = torch.empty((B, T, D), dtype=out_dtype or x.dtype, device=x.device)
    _c.check(_c.lib().cfm_dwconv_causal_bn_silu(_c.ptr(x), _c.dt_code(x), _c.ptr(cache), _c.ptr(w), _c.ptr(dw_bias), _c.ptr(bn_scale), _c.ptr(bn_shift), _c.ptr(y),
                                                _c.dt_code(y), B, T, D, K, _c.stream()), "cfm_dwconv_causal_bn_silu")
    return y


def conv_cache_update(x, cache, ktaps):
    _c.require_hip(x, cache)
    B, T, D = x.shape
    _c.check(_c.lib().cfm_conv_cache_update(_c.ptr(x), _c.dt_code(x), _c.ptr(cache), B, T, D, ktaps, _c.stream()), "cfm_conv_cache_update")
