"""Positional encodings and multi-head self-attention of the conformer, MI355X-native.

Drop-in for the reference's ``src/attention.py``: same four class names, constructor arguments, parameter names
(``linear_{q,k,v,out}``, ``linear_pos`` without bias, ``pos_bias_u/v``) and ``forward`` signatures
(attention.py:6-29, 34-100, 105-127, 130-179).  The reference's behaviours that bind parity are kept on purpose:

* ``forward`` of both encodings slices the table by ``inputs.size(0)`` -- the BATCH size (attention.py:20,119) -- so the
  batch path hands attention a (B,1,D) ``pos_embed`` and the positional score term is one value per query row;
  the streaming path hands it (Tk,1,D) and the term is an absolute position-by-key bias.  No relative shift anywhere.
* the relative encoding neither scales x by sqrt(D) nor adds the table; the absolute one adds a table that was stored
  in float16 (values rounded) (attention.py:113).
* masked scores are -inf and a fully masked row yields a zero context (attention.py:89-92).
* "no mask" / "no cache" are signalled by zero-sized tensors, never None.

Projections are MFMA GEMMs (Q,K,V fused into one N=3D GEMM when query/key/value are the same tensor); scores, the
positional term, scaling, masking, softmax and the value product are ONE fused kernel (csrc/attention.hip) -- no
(B,H,T,T) tensor is ever written to HBM.
"""
import math

import torch
import torch.nn as nn

import cfm
from cfm import packing
from feedforward import _inference_only


def _sinusoid_table(max_len, d_model, dtype=torch.float32):
    """pe[t, 2i] = sin(t * w_i), pe[t, 2i+1] = cos(t * w_i), w_i = exp(-2i ln(1e4)/d): host, torch f32 (built once)."""
    rate = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    angle = torch.arange(max_len).unsqueeze(1) * rate
    table = torch.zeros(max_len, 1, d_model, dtype=dtype)
    table[:, 0, 0::2] = torch.sin(angle)
    table[:, 0, 1::2] = torch.cos(angle)
    return table


class _SinusoidBase(nn.Module):
    _table_dtype = torch.float32

    def __init__(self, d_model, dropout, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        self.pe = _sinusoid_table(max_len, d_model, self._table_dtype)     # plain attribute, not in state_dict

    def _table_like(self, ref):
        if self.pe.device != ref.device or self.pe.dtype != ref.dtype:
            self.pe = self.pe.to(ref.device).to(ref.dtype)
        return self.pe

    def position_encoding(self, offset, size, apply_dropout=True):
        rows = self.pe[offset: offset + size]
        return self.dropout(rows) if apply_dropout else rows


class RelativePositionalEncoding(_SinusoidBase):

    def forward(self, inputs, offset=0):
        self._table_like(inputs)
        rows = self.position_encoding(offset, inputs.size(0), False)
        return self.dropout(inputs), self.dropout(rows)


class PositionalEncoding(_SinusoidBase):
    _table_dtype = torch.float16            # the reference stores this table in half precision (attention.py:113)

    def forward(self, inputs, offset=0, rows=None):
        """rows: the (B,1,D) table rows to add, in a caller-owned buffer, instead of the slice at `offset` (a captured HIP graph must
        not bake the offset in: StreamingSession refreshes that buffer between replays)."""
        self._table_like(inputs)
        if rows is None:
            rows = self.position_encoding(offset, inputs.size(0), False)      # (B,1,D): one row per batch item
        if inputs.is_cuda and inputs.dtype == torch.float32 and inputs.is_contiguous() and rows.size(0) == inputs.size(0):
            x = inputs.clone()
            cfm.add_rows(x.view(-1, x.size(-1)), rows.reshape(rows.size(0), -1).contiguous(), inputs.size(1))
        else:
            cfm.require_hip(inputs)
            x = inputs + rows
        return self.dropout(x), self.dropout(rows)


_NO_CACHE = torch.zeros((0, 0, 0, 0))


def _mask_args(mask, B, Tq, Tk):
    if mask is None or mask.dim() < 3 or mask.size(2) == 0:
        return None, (0, 0)
    m8 = cfm.as_u8_mask(mask)
    bm, qm, km = m8.shape
    if km != Tk or qm not in (1, Tq) or bm not in (1, B):
        raise RuntimeError("attention mask of shape %s does not broadcast to (%d,%d,%d)" % (tuple(mask.shape), B, Tq, Tk))
    return m8, (qm * km if bm == B else 0, km if qm > 1 else 0)


def _attend_train(mod, query, key, value, inputs_attn_mask, cache, relative, prec):
    """module.train(): self-attention over one tensor, no KV cache (encoder.py:73 -- the only way the reference trains it)."""
    if not (key is query and value is query):
        raise NotImplementedError("%s: train mode supports self-attention (query is key is value) only" % type(mod).__name__)
    if cache is not None and cache.dim() == 4 and cache.size(0) > 0:
        raise NotImplementedError("%s: a KV cache in train mode (streaming is inference-only)" % type(mod).__name__)
    from cfm import autograd as ag
    B, T, D = query.shape
    m8, m_str = _mask_args(inputs_attn_mask, B, T, T)
    out = ag.AttentionFn.apply(query, mod, prec, relative, m8, m_str, *mod.parameters())
    return out, torch.zeros((0, 0, 0, 0), dtype=torch.float32, device=query.device)


def _attend(mod, query, key, value, inputs_attn_mask, pos_embed, cache, relative):
    cfm.require_hip(query, key, value)
    prec = cfm.resolve_precision(mod)
    if cfm.check_mode(mod, type(mod).__name__):
        return _attend_train(mod, query, key, value, inputs_attn_mask, cache, relative, prec)
    pk = packing.pack_mhsa(mod, prec, relative)
    B, Tq, D = query.shape
    Tn = key.size(1)
    H, dk = mod.num_heads, mod.d_k
    adt = prec.act_dtype

    def rows(t):
        t = t.reshape(-1, t.size(-1))
        return (t if t.dtype == torch.float32 else t.float()).contiguous()

    same = key is query and value is query
    if same:
        qkv = cfm.gemm(rows(query), pk.qkv_w, bias=pk.qkv_b, w_lo=pk.qkv_w_lo, out_dtype=adt)           # [B*T, 3D]
        q_t, k_t, v_t = qkv, qkv[:, D:], qkv[:, 2 * D:]
        q_str, kv_sb, kv_st = (Tq * 3 * D, 3 * D), Tn * 3 * D, 3 * D
    else:
        q_t = cfm.gemm(rows(query), pk.q_w, bias=pk.q_b, w_lo=pk.q_w_lo, out_dtype=adt)
        kv = torch.empty((B * Tn, 2 * D), dtype=adt, device=query.device)
        cfm.gemm(rows(key), pk.k_w, bias=pk.k_b, w_lo=pk.k_w_lo, out=kv[:, :D])
        cfm.gemm(rows(value), pk.v_w, bias=pk.v_b, w_lo=pk.v_w_lo, out=kv[:, D:])
        k_t, v_t = kv, kv[:, D:]
        q_str, kv_sb, kv_st = (Tq * D, D), Tn * 2 * D, 2 * D

    have_cache = cache is not None and cache.dim() == 4 and cache.size(0) > 0
    old = cache.to(device=query.device, dtype=torch.float32) if have_cache else None
    Tc = old.size(2) if have_cache else 0
    Tk = Tc + Tn
    new_cache = None
    if have_cache or getattr(mod, "return_cache", True):
        new_cache = cfm.kv_cache_pack(old, k_t, v_t, (kv_sb, kv_st), (kv_sb, kv_st), B, H, Tn, dk)      # (B,H,Tk,2dk) f32
    if have_cache:
        k_src, v_src = new_cache, new_cache[..., dk:]
        k_str = v_str = (H * Tk * 2 * dk, 2 * dk, Tk * 2 * dk)
    else:
        k_src, v_src = k_t, v_t
        k_str = v_str = (kv_sb, kv_st, dk)

    p, p_str = None, (0, 0)
    if relative:
        pe = pos_embed.reshape(-1, D)
        R = pe.size(0)
        if R % B != 0 or R // B not in (1, Tk):
            raise RuntimeError("pos_embed with %d rows cannot be viewed as (B=%d, 1 or Tk=%d, H, d_k) (attention.py:78)" % (R, B, Tk))
        P = R // B
        p = cfm.gemm(rows(pe), pk.pos_w, w_lo=pk.pos_w_lo, out_dtype=adt)
        p_str = (P * D, D if P > 1 else 0)

    m8, m_str = _mask_args(inputs_attn_mask, B, Tq, Tk)
    ctx = torch.empty((B * Tq, D), dtype=adt, device=query.device)
    cfm.attention(q_t, k_src, v_src, B, H, Tq, Tk, dk, q_str, k_str, v_str, ctx, p=p, p_str=p_str,
                  bias_u=pk.bias_u, bias_v=pk.bias_v, mask=m8, mask_str=m_str, mma_code=prec.w_code, split=prec.split)
    out = cfm.gemm(ctx, pk.out_w, bias=pk.out_b, w_lo=pk.out_w_lo, out_dtype=torch.float32).view(B, Tq, D)
    if new_cache is None:
        new_cache = torch.zeros((0, 0, 0, 0), dtype=torch.float32, device=query.device)
    return out.to(query.dtype), new_cache


class RelativeMultiHeadSelfAttentionModule(nn.Module):

    def __init__(self, encoder_dim, num_heads, dropout):
        super().__init__()
        self.d_k = encoder_dim // num_heads
        self.num_heads = num_heads
        self.linear_pos = nn.Linear(encoder_dim, encoder_dim, bias=False)
        self.linear_k = nn.Linear(encoder_dim, encoder_dim)
        self.linear_q = nn.Linear(encoder_dim, encoder_dim)
        self.linear_v = nn.Linear(encoder_dim, encoder_dim)
        self.linear_out = nn.Linear(encoder_dim, encoder_dim)
        self.pos_bias_u = nn.Parameter(torch.empty(num_heads, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.empty(num_heads, self.d_k))
        self.dropout = nn.Dropout(dropout)
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)
        self.return_cache = True
        self._pack = packing.PackCache()

    def forward(self, query, key, value, inputs_attn_mask, pos_embed=None, cache=_NO_CACHE):
        return _attend(self, query, key, value, inputs_attn_mask, pos_embed, cache, True)


class MultiHeadSelfAttentionModule(nn.Module):

    def __init__(self, encoder_dim, num_heads, dropout):
        super().__init__()
        self.d_k = encoder_dim // num_heads
        self.num_heads = num_heads
        self.linear_k = nn.Linear(encoder_dim, encoder_dim)
        self.linear_q = nn.Linear(encoder_dim, encoder_dim)
        self.linear_v = nn.Linear(encoder_dim, encoder_dim)
        self.linear_out = nn.Linear(encoder_dim, encoder_dim)
        self.dropout = nn.Dropout(dropout)
        self.return_cache = True
        self._pack = packing.PackCache()

    def forward(self, query, key, value, inputs_attn_mask, pos_embed=None, cache=_NO_CACHE):
        return _attend(self, query, key, value, inputs_attn_mask, None, cache, False)
