"""CPU ORACLE for the conformer encoder hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, as plain functions over a flat ``{name: tensor}`` weight
table, the arithmetic of the reference hot path (Lingeng56/conformer-pytorch-
lightning, files under ``src/``; every function cites the reference file:line
it follows).  It is written independently of the reference's nn.Module code:
convolutions are explicit im2col / tap sums, attention is einsum, masks are closed-form
index arithmetic (numpy), positional tables are built here.

Status: PARITY PINNED.  ``tests/test_oracle_golden.py`` checks every function
below against the fixtures in ``tests/golden/*.npz``, which were produced by
running the reference itself on CPU (``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the timed CPU baseline.
The product path (``conformer-pytorch-lightning_amd/``) never imports it and has
no CPU fallback.

All float functions compute in the dtype of their inputs (float32 for parity
runs, float64 for the "truth" used when measuring reduced-precision error).
"""
import math

import numpy as np
import torch

EMPTY3 = (0, 0, 0)


# =============================================================================
# Row M -- masks: integer / bool arithmetic, bit-exact (numpy)
# =============================================================================
def pad_mask(lengths, max_len):
    """True where a frame is PADDING.  reference src/utils.py:84-93 (arange(T) >= len)."""
    lengths = np.asarray(lengths, dtype=np.int64).reshape(-1, 1)
    return np.arange(max_len, dtype=np.int64)[None, :] >= lengths


def chunk_mask(size, chunk_size, num_left_chunks):
    """Row i may attend [start_i, end_i).  reference src/utils.py:96-111 (a python row loop there).

    start_i = 0 if num_left_chunks < 0 else max((i//c - left)*c, 0);  end_i = min((i//c + 1)*c, size).
    """
    i = np.arange(size, dtype=np.int64)[:, None]
    j = np.arange(size, dtype=np.int64)[None, :]
    blk = i // chunk_size
    end = np.minimum((blk + 1) * chunk_size, size)
    if num_left_chunks < 0:
        start = np.zeros_like(blk)
    else:
        start = np.maximum((blk - num_left_chunks) * chunk_size, 0)
    return (j >= start) & (j < end)


def subsample_mask(valid_mask):
    """(B,1,T) -> (B,1,T'): reference src/convolution.py:76  mask[:, :, 2::2][:, :, 2::2]  ==  mask[:, :, 6::4]."""
    return valid_mask[:, :, 6::4]


def subsampled_len(t):
    """T' of the two stride-2 3x3 convs.  reference src/convolution.py:60-63 (((T-1)//2-1)//2)."""
    return ((t - 1) // 2 - 1) // 2


def attn_mask(valid_mask_sub, max_len, use_dynamic_chunk, use_dynamic_left_chunk, decoding_chunk_size,
              static_chunk_size, num_decoding_left_chunks, rand_chunk=None, rand_left=None):
    """reference src/utils.py:115-160.  ``valid_mask_sub`` is the (B,1,T') True=valid mask.

    The reference draws ``torch.randint`` on the host when decoding_chunk_size == 0 (:131,:139); the oracle
    takes those two draws as arguments (rand_chunk, rand_left) so the branch stays testable.
    """
    m = np.asarray(valid_mask_sub, dtype=bool)
    if use_dynamic_chunk:
        if decoding_chunk_size < 0:
            c, left = max_len, -1
        elif decoding_chunk_size > 0:
            c, left = decoding_chunk_size, num_decoding_left_chunks
        else:
            c, left = int(rand_chunk), -1
            if c > max_len // 2:
                c = max_len
            else:
                c = c % 25 + 1
                if use_dynamic_left_chunk:
                    left = int(rand_left)
        return m & chunk_mask(max_len, c, left)[None]
    if static_chunk_size > 0:
        return m & chunk_mask(max_len, static_chunk_size, num_decoding_left_chunks)[None]
    return m


# =============================================================================
# Row D -- positional tables
# =============================================================================
def rel_pos_table(d_model, max_len=5000):
    """(max_len, d_model) float32 sinusoid table.  reference src/attention.py:11-16."""
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def abs_pos_table(d_model, max_len=5000):
    """As above but the table is STORED in float16 (values rounded), reference src/attention.py:110-115."""
    return rel_pos_table(d_model, max_len).to(torch.float16)


# =============================================================================
# small float helpers
# =============================================================================
def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def linear(x, w, b=None):
    y = x @ w.transpose(0, 1)
    return y if b is None else y + b


def silu(x):
    return x * torch.sigmoid(x)


def _w(P, prefix, name):
    return P[prefix + name]


def _is_empty_mask(m):
    return m is None or (m.dim() >= 3 and m.size(2) == 0)


# =============================================================================
# Row B -- feed forward.  reference src/feedforward.py:16-21
# =============================================================================
def ffn(P, prefix, x, activation="swish"):
    h = linear(x, _w(P, prefix, "w_1.weight"), _w(P, prefix, "w_1.bias"))
    h = silu(h) if activation == "swish" else torch.relu(h)
    return linear(h, _w(P, prefix, "w_2.weight"), _w(P, prefix, "w_2.bias"))


# =============================================================================
# Rows C / C' -- attention.  reference src/attention.py:54-100 (relative) and :142-179 (plain)
# =============================================================================
def _split_heads(x, n_heads):
    b, t, d = x.shape
    return x.view(b, t, n_heads, d // n_heads).permute(0, 2, 1, 3)        # (B,H,T,dk)


def _attend(scores, v, mask):
    """masked softmax then context; fully masked rows give an all-zero context (attention.py:89-92, quirk Q2)."""
    if not _is_empty_mask(mask):
        dead = (mask.unsqueeze(1) == 0)                                     # (B,1,Tq|1,Tk)
        scores = scores.masked_fill(dead, float("-inf"))
        attn = torch.softmax(scores, dim=-1)
        attn = torch.where(dead.expand_as(attn), torch.zeros((), dtype=attn.dtype), attn)
    else:
        attn = torch.softmax(scores, dim=-1)
    ctx = torch.einsum("bhij,bhjd->bhid", attn, v)
    b, h, t, dk = ctx.shape
    return ctx.permute(0, 2, 1, 3).reshape(b, t, h * dk)


def _kv_with_cache(k, v, cache):
    if cache is not None and cache.dim() == 4 and cache.size(0) > 0:       # attention.py:70-74
        dk = cache.size(-1) // 2
        k = torch.cat([cache[..., :dk].to(k.dtype), k], dim=2)
        v = torch.cat([cache[..., dk:].to(v.dtype), v], dim=2)
    return k, v


def rel_mhsa(P, prefix, x, mask, pos_embed, cache=None, n_heads=4):
    """Relative-position MHSA exactly as the reference evaluates it (no rel-shift; quirk Q3).

    pos_embed (R,1,D): p = linear_pos(pos_embed) viewed (B, R/B, H, dk); R/B must be 1 or Tk so that
    bd (B,H,Tq,R/B) broadcasts against ac (B,H,Tq,Tk)  (attention.py:78-88).
    """
    B = x.size(0)
    q = _split_heads(linear(x, _w(P, prefix, "linear_q.weight"), _w(P, prefix, "linear_q.bias")), n_heads)
    k = _split_heads(linear(x, _w(P, prefix, "linear_k.weight"), _w(P, prefix, "linear_k.bias")), n_heads)
    v = _split_heads(linear(x, _w(P, prefix, "linear_v.weight"), _w(P, prefix, "linear_v.bias")), n_heads)
    k, v = _kv_with_cache(k, v, cache)
    new_cache = torch.cat([k, v], dim=-1)                                   # attention.py:76
    dk = q.size(-1)
    p = linear(pos_embed.to(x.dtype), _w(P, prefix, "linear_pos.weight")).reshape(B, -1, n_heads, dk).permute(0, 2, 1, 3)
    u = _w(P, prefix, "pos_bias_u")[None, :, None, :]
    vb = _w(P, prefix, "pos_bias_v")[None, :, None, :]
    ac = torch.einsum("bhid,bhjd->bhij", q + u, k)
    bd = torch.einsum("bhid,bhjd->bhij", q + vb, p)
    scores = (ac + bd) / math.sqrt(dk)
    ctx = _attend(scores, v, mask)
    out = linear(ctx, _w(P, prefix, "linear_out.weight"), _w(P, prefix, "linear_out.bias"))
    return out, new_cache


def mhsa(P, prefix, x, mask, cache=None, n_heads=4):
    """Plain MHSA (use_relative=False).  reference src/attention.py:142-179 (the trailing dropout is identity in eval)."""
    q = _split_heads(linear(x, _w(P, prefix, "linear_q.weight"), _w(P, prefix, "linear_q.bias")), n_heads)
    k = _split_heads(linear(x, _w(P, prefix, "linear_k.weight"), _w(P, prefix, "linear_k.bias")), n_heads)
    v = _split_heads(linear(x, _w(P, prefix, "linear_v.weight"), _w(P, prefix, "linear_v.bias")), n_heads)
    k, v = _kv_with_cache(k, v, cache)
    new_cache = torch.cat([k, v], dim=-1)
    scores = torch.einsum("bhid,bhjd->bhij", q, k) / math.sqrt(q.size(-1))
    ctx = _attend(scores, v, mask)
    out = linear(ctx, _w(P, prefix, "linear_out.weight"), _w(P, prefix, "linear_out.bias"))
    return out, new_cache


# =============================================================================
# Row E -- convolution module.  reference src/convolution.py:34-49
# =============================================================================
def conv_module(P, prefix, x, valid_mask, bn_eps=1e-5, train=False, bn_out=None, momentum=0.1, causal=False, conv_cache=None, cache_out=None):
    """x (B,T,D) time-major.  train=True: BatchNorm1d in training mode (convolution.py:44 under module.train()) -- batch mean and
    BIASED variance over ALL B*T positions of each channel, padded frames included (quirk Q6); the running statistics it would
    leave behind (momentum 0.1, UNBIASED variance, torch's rule) are returned through bn_out[prefix] = (mean, var).
    causal=True is NOT the reference (which has no causal mode, convolution.py:34-39): it restates the build's opt-in extension -- the
    depthwise taps reach back K-1 frames, conv_cache (B,K-1,D) is the left context (zeros when None), cache_out[prefix] receives the next one.
    Channels-last restatement of
    mask -> pointwise(D->2D) -> GLU -> depthwise k (zero pad only at the tensor edges) -> BatchNorm(eval) -> SiLU
    -> pointwise(D->D) -> mask.   Quirk Q5: masking precedes pw1, so padded frames carry GLU(bias) into the halo.
    """
    B, T, D = x.shape
    use_mask = not _is_empty_mask(valid_mask)
    if use_mask:
        keep = valid_mask.reshape(B, T, 1).to(torch.bool)                   # (B,1,T) -> (B,T,1)
        x = torch.where(keep, x, torch.zeros((), dtype=x.dtype))
    w1 = _w(P, prefix, "pointwise_conv1.weight")[:, :, 0]                   # (2D, D)
    y = linear(x, w1, _w(P, prefix, "pointwise_conv1.bias"))
    y = y[..., :D] * torch.sigmoid(y[..., D:])                              # GLU over channels (dim=1 in NCT)
    wd = _w(P, prefix, "depthwise_conv.weight")[:, 0, :]                    # (D, K)
    K = wd.size(1)
    half = (K - 1) // 2
    if causal:
        ctx = torch.zeros(B, K - 1, D, dtype=y.dtype) if conv_cache is None else conv_cache.to(y.dtype)
        ypad = torch.cat([ctx, y], dim=1)
        if cache_out is not None:
            cache_out[prefix] = ypad[:, -(K - 1):].clone()
    else:
        ypad = torch.zeros(B, T + 2 * half, D, dtype=y.dtype)
        ypad[:, half:half + T] = y
    z = torch.zeros_like(y)
    for tap in range(K):                                                    # cross-correlation, like nn.Conv1d
        z = z + ypad[:, tap:tap + T] * wd[:, tap]
    z = z + _w(P, prefix, "depthwise_conv.bias")
    if train:
        n = B * T
        mu = z.reshape(n, D).mean(0)
        var = ((z.reshape(n, D) - mu) ** 2).mean(0)
        if bn_out is not None:
            rm, rv = _w(P, prefix, "norm.running_mean"), _w(P, prefix, "norm.running_var")
            bn_out[prefix] = ((1 - momentum) * rm + momentum * mu.detach().to(rm.dtype),
                              (1 - momentum) * rv + momentum * (var.detach() * n / max(n - 1, 1)).to(rv.dtype))
        z = (z - mu) / torch.sqrt(var + bn_eps)
    else:
        z = (z - _w(P, prefix, "norm.running_mean")) / torch.sqrt(_w(P, prefix, "norm.running_var") + bn_eps)
    z = z * _w(P, prefix, "norm.weight") + _w(P, prefix, "norm.bias")
    z = silu(z)
    w2 = _w(P, prefix, "pointwise_conv2.weight")[:, :, 0]
    out = linear(z, w2, _w(P, prefix, "pointwise_conv2.bias"))
    if use_mask:
        out = torch.where(keep, out, torch.zeros((), dtype=out.dtype))
    return out


# =============================================================================
# Row F -- Conv2d subsampling front-end.  reference src/convolution.py:70-76
# =============================================================================
def _conv3x3_s2_relu(x, w, b):
    """x (B,Ci,T,F), w (Co,Ci,3,3) -> relu(conv stride 2, no padding), restated as im2col + one GEMM:
    patches[b,t,f,(ci,kt,kf)] = x[b,ci,2t+kt,2f+kf];  out = patches . w.reshape(Co, Ci*9)^T + bias."""
    B, Ci, T, F = x.shape
    Co = w.size(0)
    To, Fo = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    sb, sc, st, sf = x.stride()
    patches = x.as_strided((B, To, Fo, Ci, 3, 3), (sb, 2 * st, 2 * sf, sc, st, sf))      # a view: no copy yet
    cols = patches.reshape(B * To * Fo, Ci * 9)                                            # materialises the im2col matrix
    out = cols @ w.reshape(Co, Ci * 9).transpose(0, 1) + b
    return torch.relu(out).reshape(B, To, Fo, Co).permute(0, 3, 1, 2)


def subsampling(P, prefix, x, valid_mask, pe_table, offset=0, relative=True):
    """Returns (x', pos_embed, mask').  pos_embed is pe[offset : offset + B] -- sliced by the BATCH size, as the
    reference does (attention.py:20, quirk Q3); with relative=False the table row is ADDED to x' (attention.py:117-121)."""
    B = x.size(0)
    h = _conv3x3_s2_relu(x.unsqueeze(1), _w(P, prefix, "conv.0.weight"), _w(P, prefix, "conv.0.bias"))
    h = _conv3x3_s2_relu(h, _w(P, prefix, "conv.2.weight"), _w(P, prefix, "conv.2.bias"))
    b, c, t, f = h.shape
    feat = h.permute(0, 2, 1, 3).reshape(b, t, c * f)                        # channel-major features c*F'+f
    y = linear(feat, _w(P, prefix, "out.0.weight"), _w(P, prefix, "out.0.bias"))
    pos = pe_table[offset:offset + B].to(y.dtype).unsqueeze(1)              # (B,1,D)
    if not relative:
        y = y + pos
    return y, pos, valid_mask[:, :, 6::4]


# =============================================================================
# Row A -- one conformer block.  reference src/encoder_layer.py:49-71
# =============================================================================
def encoder_layer(P, prefix, x, attn_msk, pos_embed, pad_msk=None, attn_cache=None, n_heads=4, relative=True, train=False, bn_out=None):
    """train=True is module.train() with every dropout probability 0: the only difference is the BatchNorm of the conv module."""
    def ln(name, t):
        return layer_norm(t, _w(P, prefix, name + ".weight"), _w(P, prefix, name + ".bias"))
    x = x + 0.5 * ffn(P, prefix + "feed_forward_macaron.", ln("norm_ff_macaron", x))
    if relative:
        a, new_cache = rel_mhsa(P, prefix + "self_attn.", ln("norm_mha", x), attn_msk, pos_embed, attn_cache, n_heads)
    else:
        a, new_cache = mhsa(P, prefix + "self_attn.", ln("norm_mha", x), attn_msk, attn_cache, n_heads)
    x = x + a
    x = x + conv_module(P, prefix + "conv_module.", ln("norm_conv", x), pad_msk, train=train, bn_out=bn_out)
    x = x + 0.5 * ffn(P, prefix + "feed_forward.", ln("norm_ff", x))
    return ln("norm_final", x), new_cache


# =============================================================================
# Row N -- encoder driver.  reference src/encoder.py:54-75 (forward), :78-123 (forward_chunk), :125-153
# =============================================================================
class Config:
    def __init__(self, encoder_dim, num_heads, encoder_num_layers, use_relative=True, max_len=5000,
                 use_dynamic_chunk_size=False, use_dynamic_left_chunk=False, static_chunk_size=-1, **_ignored):
        self.d = encoder_dim
        self.h = num_heads
        self.layers = encoder_num_layers
        self.relative = use_relative
        self.max_len = max_len
        self.dyn_chunk = use_dynamic_chunk_size
        self.dyn_left = use_dynamic_left_chunk
        self.static_chunk = static_chunk_size
        self.pe = rel_pos_table(self.d, max_len) if use_relative else abs_pos_table(self.d, max_len)


def encoder_forward(P, cfg, x, lengths, decoding_chunk_size=0, num_decoding_left_chunks=-1,
                    rand_chunk=None, rand_left=None, collect=None, train=False, bn_out=None):
    """Whole-utterance forward.  Returns (y (B,T',D), valid mask (B,1,T') bool)."""
    T = x.size(1)
    valid = torch.from_numpy(~pad_mask(np.asarray(lengths), T)).unsqueeze(1)
    h, pos, valid_s = subsampling(P, "embed.", x, valid, cfg.pe, 0, cfg.relative)
    if collect is not None:
        collect["embed_out"] = h
    am = torch.from_numpy(np.ascontiguousarray(attn_mask(valid_s.numpy(), h.size(1), cfg.dyn_chunk, cfg.dyn_left,
                                                         decoding_chunk_size, cfg.static_chunk,
                                                         num_decoding_left_chunks, rand_chunk, rand_left)))
    for li in range(cfg.layers):
        h, _ = encoder_layer(P, "encoders.%d." % li, h, am, pos, valid_s, None, cfg.h, cfg.relative, train=train, bn_out=bn_out)
        if collect is not None:
            collect["layer_out_%d" % li] = h
    y = layer_norm(h, P["after_norm.weight"], P["after_norm.bias"])
    return y, valid_s


def encoder_forward_chunk(P, cfg, x, offset, required_cache_size, attn_cache):
    """One streaming step at batch 1.  attn_cache (L,H,Tc,2dk) or None/empty.  Returns (y, new_cache).

    No masks are applied (reference passes an empty attention mask and no pad mask, encoder.py:83,110-116) and the
    convolution module keeps NO left-context cache (quirk Q4).  pos_embed = pe[offset-Tc : offset+chunk] (encoder.py:98-100).
    """
    ones = torch.ones(1, 1, x.size(1), dtype=torch.bool)
    h, _, _ = subsampling(P, "embed.", x, ones, cfg.pe, offset, cfg.relative)
    have = attn_cache is not None and attn_cache.dim() == 4 and attn_cache.size(0) > 0
    tc = attn_cache.size(2) if have else 0
    chunk = h.size(1)
    tk = tc + chunk
    pos = cfg.pe[offset - tc:offset - tc + tk].to(h.dtype).unsqueeze(1)
    if required_cache_size < 0:
        start = 0
    elif required_cache_size == 0:
        start = tk
    else:
        start = max(tk - required_cache_size, 0)
    new = []
    for li in range(cfg.layers):
        c = attn_cache[li:li + 1] if have else None
        h, nc = encoder_layer(P, "encoders.%d." % li, h, None, pos, None, c, cfg.h, cfg.relative)
        new.append(nc[:, :, start:, :])
    y = layer_norm(h, P["after_norm.weight"], P["after_norm.bias"])
    return y, torch.cat(new, dim=0)


def encoder_forward_chunk_by_chunk(P, cfg, x, decoding_chunk_size, num_decoding_left_chunks=-1):
    """reference src/encoder.py:125-153: window (c-1)*4+7 frames, hop 4*c, cache c*left (<0: unbounded)."""
    hop = 4 * decoding_chunk_size
    window = (decoding_chunk_size - 1) * 4 + 7
    need = decoding_chunk_size * num_decoding_left_chunks
    cache, outs, offset = None, [], 0
    for cur in range(0, x.size(1) - 7 + 1, hop):
        y, cache = encoder_forward_chunk(P, cfg, x[:, cur:min(cur + window, x.size(1))], offset, need, cache)
        outs.append(y)
        offset += y.size(1)
    return torch.cat(outs, dim=1)


# =============================================================================
# algorithmic work model -- used by bench.py for the roofline figure
# =============================================================================
# =============================================================================
# SURVEY 8(f) rank 1 -- CTC head.  reference src/decoder.py:18-23 (dropout p = 0: the reference applies F.dropout with
# training=True even in eval, quirk Q7, which no deterministic restatement can follow)
# =============================================================================
def ctc_nll(logp, labels, blank=0):
    """-log p(labels | logp) by the alpha recursion over the blank-extended sequence (what nn.CTCLoss computes per item).
    logp (T,V) log-probabilities, labels a 1-D int sequence; float64 inside."""
    lp = np.asarray(logp, dtype=np.float64)
    z = [blank]
    for y in np.asarray(labels).tolist():
        z += [int(y), blank]
    S, T = len(z), lp.shape[0]
    if T == 0:
        return 0.0 if S == 1 else float("inf")
    ninf = -np.inf
    alpha = np.full(S, ninf)
    alpha[0] = lp[0, z[0]]
    if S > 1:
        alpha[1] = lp[0, z[1]]
    for t in range(1, T):
        prev = alpha
        alpha = np.full(S, ninf)
        for s in range(S):
            a = prev[s]
            if s >= 1:
                a = np.logaddexp(a, prev[s - 1])
            if s >= 2 and z[s] != blank and z[s] != z[s - 2]:
                a = np.logaddexp(a, prev[s - 2])
            alpha[s] = a + lp[t, z[s]]
    tot = alpha[S - 1] if S == 1 else np.logaddexp(alpha[S - 1], alpha[S - 2])
    return float(-tot)


def ctc_nll_and_grad(logp, labels, blank=0):
    """(nll, d nll / d logits) of one utterance by the alpha-beta recursions in float64.  logp (T,V) = log_softmax(logits).
    d nll / d logits[t,c] = softmax[t,c] - (1/P) sum_{s: z_s = c} alpha_t(s) beta_t(s) / y[t,c]   (Graves et al. 2006, eq. 16;
    what nn.CTCLoss differentiates to, decoder.py:20-21).  An impossible alignment has nll = inf and (as torch without
    zero_infinity) an undefined gradient: returned as zeros here."""
    lp = np.asarray(logp, dtype=np.float64)
    z = [blank]
    for y in np.asarray(labels).tolist():
        z += [int(y), blank]
    S, T = len(z), lp.shape[0]
    ninf = -np.inf
    alpha = np.full((T, S), ninf)
    beta = np.full((T, S), ninf)
    alpha[0, 0] = lp[0, z[0]]
    if S > 1:
        alpha[0, 1] = lp[0, z[1]]
    for t in range(1, T):
        for s in range(S):
            a = alpha[t - 1, s]
            if s >= 1:
                a = np.logaddexp(a, alpha[t - 1, s - 1])
            if s >= 2 and z[s] != blank and z[s] != z[s - 2]:
                a = np.logaddexp(a, alpha[t - 1, s - 2])
            alpha[t, s] = a + lp[t, z[s]]
    beta[T - 1, S - 1] = lp[T - 1, z[S - 1]]
    if S > 1:
        beta[T - 1, S - 2] = lp[T - 1, z[S - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(S):
            a = beta[t + 1, s]
            if s + 1 < S:
                a = np.logaddexp(a, beta[t + 1, s + 1])
            if s + 2 < S and z[s + 2] != blank and z[s + 2] != z[s]:
                a = np.logaddexp(a, beta[t + 1, s + 2])
            beta[t, s] = a + lp[t, z[s]]
    tot = alpha[T - 1, S - 1] if S == 1 else np.logaddexp(alpha[T - 1, S - 1], alpha[T - 1, S - 2])
    grad = np.exp(lp)
    if not np.isfinite(tot):
        return float("inf"), np.zeros_like(grad)
    occ = np.zeros_like(grad)
    with np.errstate(under="ignore"):
        for s in range(S):
            occ[:, z[s]] += np.exp(alpha[:, s] + beta[:, s] - lp[:, z[s]] - tot)
    return float(-tot), grad - occ


class _CTCHead(torch.autograd.Function):
    """sum_b nll_b as a differentiable function of the logits (B,T,V): forward and backward are the float64 numpy recursions above."""

    @staticmethod
    def forward(ctx, logits, enc_lens, labels, label_lens):
        logp = torch.log_softmax(logits.detach().double(), dim=-1).numpy()
        grad = np.zeros(logp.shape, dtype=np.float64)
        total = 0.0
        for b in range(logp.shape[0]):
            n, u = int(enc_lens[b]), int(label_lens[b])
            nll, g = ctc_nll_and_grad(logp[b, :n], np.asarray(labels)[b, :u])
            grad[b, :n] = g
            total += nll
        ctx.grad = torch.from_numpy(grad).to(logits.dtype)
        return logits.new_tensor(total)

    @staticmethod
    def backward(ctx, gout):
        return ctx.grad * gout, None, None, None


def ctc_head_loss_autograd(P, prefix, enc_out, enc_lens, labels, label_lens):
    """Differentiable CTCDecoder.forward (dropout 0): loss = sum_b nll_b / padded label length (decoder.py:19-22)."""
    logits = linear(enc_out, _w(P, prefix, "ctc_lo.weight"), _w(P, prefix, "ctc_lo.bias"))
    return _CTCHead.apply(logits, np.asarray(enc_lens), np.asarray(labels), np.asarray(label_lens)) / np.asarray(labels).shape[1]


def ctc_head_loss(P, prefix, enc_out, enc_lens, labels, label_lens):
    """CTCDecoder.forward with dropout 0: Linear -> log_softmax over the vocabulary -> sum of the per-utterance CTC negative
    log-likelihoods / padded label length (decoder.py:19-22).  Returns (loss, per-utterance nll)."""
    logits = linear(enc_out, _w(P, prefix, "ctc_lo.weight"), _w(P, prefix, "ctc_lo.bias"))
    logp = torch.log_softmax(logits.double(), dim=-1).numpy()
    lab = np.asarray(labels)
    nll = np.array([ctc_nll(logp[b, :int(enc_lens[b])], lab[b, :int(label_lens[b])]) for b in range(logp.shape[0])])
    return float(nll.sum() / lab.shape[1]), nll


def joint_forward(P, prefix, enc_out, pred_out, pre_project=True):
    """TransducerJoint.forward (joint.py:20-38): enc_ffn / pred_ffn projections (unless already projected), broadcast sum over
    (B, T, 1, J) + (B, 1, U, J), tanh, ffn_out -> logits (B, T, U, V).  3-D inputs get the singleton axis as joint.py:29-33 does."""
    if pre_project:
        enc_out = linear(enc_out, _w(P, prefix, "enc_ffn.weight"), _w(P, prefix, "enc_ffn.bias"))
        pred_out = linear(pred_out, _w(P, prefix, "pred_ffn.weight"), _w(P, prefix, "pred_ffn.bias"))
    if enc_out.dim() != 4:
        enc_out = enc_out.unsqueeze(2)
    if pred_out.dim() != 4:
        pred_out = pred_out.unsqueeze(1)
    return linear(torch.tanh(enc_out + pred_out), _w(P, prefix, "ffn_out.weight"), _w(P, prefix, "ffn_out.bias"))


# ----------------------------------------------------------------------------------------------------------------------
# RNN-T greedy search (test infrastructure, like everything in this file): the reference's basic_greedy_search,
# /root/reference/src/model.py:215-269, restated for ONE utterance with the predictor of src/predictor.py:76-86 (Embedding -> LSTM ->
# Linear, one symbol per call) and the joint of src/joint.py:20-38.  model.py itself cannot be imported here (torchaudio is absent,
# SURVEY.md 8c): the LOOP is pinned by this restatement only; the predictor / joint STEP it calls is pinned by tokens generated with the
# reference's own modules (tests/golden/greedy.npz, tests/golden/make_golden.py gen_greedy).
# ----------------------------------------------------------------------------------------------------------------------
def _lstm_cell(P, prefix, layer, x, h, c):
    w_ih, w_hh = _w(P, prefix, "rnn.weight_ih_l%d" % layer), _w(P, prefix, "rnn.weight_hh_l%d" % layer)
    b_ih, b_hh = _w(P, prefix, "rnn.bias_ih_l%d" % layer), _w(P, prefix, "rnn.bias_hh_l%d" % layer)
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, g, o = gates.chunk(4, dim=-1)
    c1 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c1), c1


def predictor_step(P, prefix, token, h, c):
    """predictor.py:76-86 with padding = 0: token () int, h / c (L, H) -> (projection output (P,), h', c')."""
    x = _w(P, prefix, "embed.weight")[int(token)]
    hs, cs = [], []
    for l in range(h.shape[0]):
        x, c1 = _lstm_cell(P, prefix, l, x, h[l], c[l])
        hs.append(x)
        cs.append(c1)
    return linear(x, _w(P, prefix, "projection.weight"), _w(P, prefix, "projection.bias")), torch.stack(hs), torch.stack(cs)


def rnnt_greedy_search(P, pred_prefix, joint_prefix, enc_out, enc_len, blank=0, n_steps=64, token=None, state=None):
    """model.py:215-269 for one utterance: enc_out (T', E), enc_len valid frames.  Returns (tokens, (token, (h, c)))."""
    dt = enc_out.dtype
    L, H = _num_lstm_layers(P, pred_prefix), _w(P, pred_prefix, "rnn.weight_hh_l0").shape[1]
    h, c = (torch.zeros((L, H), dtype=dt), torch.zeros((L, H), dtype=dt)) if state is None else (state[0].to(dt), state[1].to(dt))
    token = blank if token is None else int(token)
    t, hyps, prev_nonblank, per_frame = 0, [], True, 0
    pred_out, new_h, new_c = None, h, c
    while t < int(enc_len):                                            # model.py:243
        if prev_nonblank:                                              # :245-248
            pred_out, new_h, new_c = predictor_step(P, pred_prefix, token, h, c)
        logits = joint_forward(P, joint_prefix, enc_out[t][None, None, :], pred_out[None, None, :]).reshape(-1)   # :250-252
        k = int(torch.argmax(torch.log_softmax(logits, dim=-1)))       # :254
        if k != blank:                                                 # :255-261
            hyps.append(k)
            prev_nonblank = True
            per_frame += 1
            token, h, c = k, new_h, new_c
        if k == blank or per_frame >= n_steps:                         # :263-267
            if k == blank:
                prev_nonblank = False
            t += 1
            per_frame = 0
    return hyps, (token, (h, c))


def _num_lstm_layers(P, prefix):
    n = 0
    while (prefix + "rnn.weight_ih_l%d" % n) in P:
        n += 1
    return n


def encoder_flops_per_utt(T, F=80, D=256, FF=2048, K=15, L=12):
    """2*MAC of the GEMM/conv/attention contractions of one utterance's forward (elementwise/LN excluded)."""
    t1, f1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    tp, fp = (t1 - 3) // 2 + 1, (f1 - 3) // 2 + 1
    mac = D * t1 * f1 * 9 + D * D * tp * fp * 9 + tp * D * D * fp
    per_layer = 4 * tp * D * FF + 4 * tp * D * D + 2 * tp * tp * D + 2 * tp * D * D + tp * D * K + tp * D * D
    return 2 * (mac + L * per_layer)
