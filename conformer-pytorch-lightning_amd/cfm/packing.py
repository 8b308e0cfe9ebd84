"""Device-side weight packing for the HIP kernels.

The nn.Modules keep the reference's parameter names / shapes / dtypes (so state_dicts, optimizers and DDP see what they
expect); the kernels want: 16-bit (or hi/lo bf16 split) [N,K] matrices, a fused QKV matrix, the pointwise-conv-1
rows interleaved for the in-register GLU, BatchNorm folded to scale/shift, conv kernels reordered to the implicit-GEMM K
order.  A pack is rebuilt only when a source tensor's version counter / storage / device changes or the precision mode
changes (in-place optimizer steps bump ``_version``).
"""
import torch

import cfm as _c


_EPOCH = [0]


def bump_epoch():
    """Invalidate every pack: for weight updates that bypass torch's version counters (the flat-buffer Adam kernel of trainer.py writes
    through raw pointers)."""
    _EPOCH[0] += 1


def _key(tensors, prec):
    return (prec.name, _EPOCH[0]) + tuple((t.data_ptr(), t._version, str(t.device), t.dtype) for t in tensors if t is not None)


class Packed:
    """Attribute bag of packed tensors (keeps them alive while raw pointers are in flight)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class PackCache:
    def __init__(self):
        self._key = None
        self._val = None

    def get(self, tensors, prec, build):
        k = _key(tensors, prec)
        if k != self._key:
            with torch.no_grad():
                self._val = build()
            self._key = k
        return self._val


def f32(t):
    return t.detach().to(torch.float32).contiguous()


def matrix(w, prec):
    """[N,K] float weight -> (W16, W_lo|None)."""
    w32 = w.detach().to(torch.float32)
    if prec.split:
        hi = w32.to(torch.bfloat16)
        lo = (w32 - hi.to(torch.float32)).to(torch.bfloat16)
        return hi.contiguous(), lo.contiguous()
    return w32.to(prec.w_dtype).contiguous(), None


def glu_interleave_index(D, device):
    """GEMM column blk*32 + half*16 + i  takes weight row  half*D + blk*16 + i  (see gemm.hip epilogue)."""
    idx = torch.arange(2 * D, device=device)
    return ((idx % 32) // 16) * D + (idx // 32) * 16 + (idx % 16)


def pack_ffn_fragments(w1, w2, dtype):
    """Fragment-major packs for the fused FFN kernel (include/cfm.h cfm_ffn_fused, csrc/ffn.hip).

    w1 [FF,D] -> w1f[ffb][kk][lane][j] = w1[ffb*16 + (lane&15)][kk*32 + 8*(lane>>4) + j]           (K zero-padded to 32)
    w2 [D,FF] -> w2f[fs][nf][lane][j]  = w2[nf*16 + (lane&15)][fs*32 + (j<4 ? 0 : 16) + 4*(lane>>4) + (j&3)]
    so that one wavefront-load (lane*16 bytes) is one MFMA A-operand fragment, and the second product's k order is the
    order in which the first product's accumulators hold the hidden activation.
    """
    FF, D = w1.shape
    ks1 = (D + 31) // 32
    w1p = torch.zeros((FF, ks1 * 32), dtype=torch.float32, device=w1.device)
    w1p[:, :D] = w1.detach().float()
    # [ffb, l15, kk, g, j] -> [ffb, kk, g, l15, j]   (lane = g*16 + l15)
    w1f = w1p.view(FF // 16, 16, ks1, 4, 8).permute(0, 2, 3, 1, 4).contiguous().to(dtype)
    # ff = fs*32 + hi*16 + g*4 + r ;  j = hi*4 + r :  [nf, l15, fs, hi, g, r] -> [fs, nf, g, l15, hi, r]
    w2f = w2.detach().float().view(D // 16, 16, FF // 32, 2, 4, 4).permute(2, 0, 4, 1, 3, 5).contiguous().to(dtype)
    return w1f.view(-1), w2f.view(-1)


def pack_frag_major(w, dtype):
    """[N,K] -> 16-bit fragment-major w[nfrag][kk][lane][j] = W[nfrag*16 + (lane&15)][kk*32 + 8*(lane>>4) + j], K zero-padded to a
    multiple of 32 (the operand layout of the row-chain kernels: one wavefront-load of lane*16 B = one MFMA A fragment)."""
    N, K = w.shape
    ks = (K + 31) // 32
    wp = torch.zeros((N, ks * 32), dtype=torch.float32, device=w.device)
    wp[:, :K] = w.detach().float()
    return wp.view(N // 16, 16, ks, 4, 8).permute(0, 2, 3, 1, 4).contiguous().to(dtype).view(-1)


def pack_ffn(mod, prec):
    def build():
        w1, w1l = matrix(mod.w_1.weight, prec)
        w2, w2l = matrix(mod.w_2.weight, prec)
        pk = Packed(w1=w1, w1_lo=w1l, b1=f32(mod.w_1.bias), w2=w2, w2_lo=w2l, b2=f32(mod.w_2.bias), w1f=None, w2f=None, w2n=None)
        FF, D = mod.w_1.weight.shape
        if _c.ffn_fused_supported(D, FF, prec):
            pk.w1f, pk.w2f = pack_ffn_fragments(mod.w_1.weight, mod.w_2.weight, prec.w_dtype)
        if _c.rowchain_supported(D, FF, prec):
            pk.w2n = pack_frag_major(mod.w_2.weight, prec.w_dtype)      # the row chains read W2 in natural k order
            if pk.w1f is None:                                          # (D = 512: no stand-alone fused kernel, the chains only)
                pk.w1f = pack_frag_major(mod.w_1.weight, prec.w_dtype)
        return pk
    return mod._pack.get([mod.w_1.weight, mod.w_1.bias, mod.w_2.weight, mod.w_2.bias], prec, build)


def pack_mhsa(mod, prec, relative):
    srcs = [mod.linear_q.weight, mod.linear_q.bias, mod.linear_k.weight, mod.linear_k.bias, mod.linear_v.weight,
            mod.linear_v.bias, mod.linear_out.weight, mod.linear_out.bias]
    if relative:
        srcs += [mod.linear_pos.weight, mod.pos_bias_u, mod.pos_bias_v]

    def build():
        qkv, qkvl = matrix(torch.cat([mod.linear_q.weight, mod.linear_k.weight, mod.linear_v.weight], 0), prec)
        out, outl = matrix(mod.linear_out.weight, prec)
        p = Packed(qkv_w=qkv, qkv_w_lo=qkvl, qkv_b=f32(torch.cat([mod.linear_q.bias, mod.linear_k.bias, mod.linear_v.bias], 0)),
                   out_w=out, out_w_lo=outl, out_b=f32(mod.linear_out.bias), pos_w=None, pos_w_lo=None, bias_u=None, bias_v=None)
        D = mod.linear_q.weight.shape[0]
        # row views of the fused matrix for the non-self-attention call pattern (query/key/value differ)
        p.q_w, p.k_w, p.v_w = qkv[:D], qkv[D:2 * D], qkv[2 * D:]
        p.q_w_lo, p.k_w_lo, p.v_w_lo = (None, None, None) if qkvl is None else (qkvl[:D], qkvl[D:2 * D], qkvl[2 * D:])
        p.q_b, p.k_b, p.v_b = p.qkv_b[:D], p.qkv_b[D:2 * D], p.qkv_b[2 * D:]
        if relative:
            p.pos_w, p.pos_w_lo = matrix(mod.linear_pos.weight, prec)
            p.bias_u, p.bias_v = f32(mod.pos_bias_u), f32(mod.pos_bias_v)
        p.qkv_wf = p.out_wf = None
        if not prec.split and D % 16 == 0:
            p.qkv_wf = pack_frag_major(torch.cat([mod.linear_q.weight, mod.linear_k.weight, mod.linear_v.weight], 0), prec.w_dtype)
            p.out_wf = pack_frag_major(mod.linear_out.weight, prec.w_dtype)
        return p
    return mod._pack.get(srcs, prec, build)


def pack_conv_module(mod, prec):
    bn = mod.norm
    srcs = [mod.pointwise_conv1.weight, mod.pointwise_conv1.bias, mod.depthwise_conv.weight, mod.depthwise_conv.bias, bn.weight,
            bn.bias, bn.running_mean, bn.running_var, mod.pointwise_conv2.weight, mod.pointwise_conv2.bias]

    def build():
        D = mod.pointwise_conv2.weight.shape[0]
        dev = mod.pointwise_conv2.weight.device
        idx = glu_interleave_index(D, dev)
        w1 = mod.pointwise_conv1.weight.detach()[:, :, 0]
        b1 = mod.pointwise_conv1.bias.detach() if mod.pointwise_conv1.bias is not None else torch.zeros(2 * D, device=dev)
        pw1, pw1l = matrix(w1[idx], prec)
        pw2, pw2l = matrix(mod.pointwise_conv2.weight.detach()[:, :, 0], prec)
        dwb = mod.depthwise_conv.bias.detach() if mod.depthwise_conv.bias is not None else torch.zeros(D, device=dev)
        # BatchNorm1d (eval): y = (x - mean) / sqrt(var + eps) * gamma + beta   ->   x * scale + shift
        gamma = bn.weight.detach().float() if bn.weight is not None else torch.ones(D, device=dev)
        beta = bn.bias.detach().float() if bn.bias is not None else torch.zeros(D, device=dev)
        scale = gamma / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        shift = beta - bn.running_mean.detach().float() * scale
        pk = Packed(pw1_w=pw1, pw1_w_lo=pw1l, pw1_b=f32(b1[idx]), pw2_w=pw2, pw2_w_lo=pw2l, pw2_b=f32(mod.pointwise_conv2.bias),
                    dw_w=f32(mod.depthwise_conv.weight.detach()[:, 0, :]), dw_b=f32(dwb), bn_scale=f32(scale), bn_shift=f32(shift),
                    pw1_wf=None, pw2_wf=None)
        if not prec.split and D % 16 == 0:
            pk.pw1_wf = pack_frag_major(w1[idx], prec.w_dtype)          # same value/gate interleave as the GEMM path
            pk.pw2_wf = pack_frag_major(mod.pointwise_conv2.weight.detach()[:, :, 0], prec.w_dtype)
        return pk
    return mod._pack.get(srcs, prec, build)


def pack_subsampling(mod, prec):
    c1, c2, lin = mod.conv[0], mod.conv[2], mod.out[0]
    srcs = [c1.weight, c1.bias, c2.weight, c2.bias, lin.weight, lin.bias]

    def build():
        C = c1.weight.shape[0]
        w1 = f32(c1.weight.detach().reshape(C, 9).t())                               # [9, C] tap-major
        w2, w2l = matrix(c2.weight.detach().permute(0, 2, 3, 1).reshape(C, 9 * C), prec)   # [co][kt][kf][ci]
        Dout, CF = lin.weight.shape
        Fp = CF // C
        # reference feature order is c*F'+f (convolution.py:74); the implicit GEMM writes [.., f, c]
        wl, wll = matrix(lin.weight.detach().reshape(Dout, C, Fp).permute(0, 2, 1).reshape(Dout, Fp * C), prec)
        return Packed(w1=w1, b1=f32(c1.bias), w2=w2, w2_lo=w2l, b2=f32(c2.bias), wl=wl, wl_lo=wll, bl=f32(lin.bias), C=C, Fp=Fp)
    return mod._pack.get(srcs, prec, build)


def layer_weight_struct(layer, prec):
    """cfm.LayerWeights for one ConformerEncoderLayer (+ the Packed objects it points into)."""
    relative = layer.use_relative
    ffm = pack_ffn(layer.feed_forward_macaron, prec)
    ff = pack_ffn(layer.feed_forward, prec)
    att = pack_mhsa(layer.self_attn, prec, relative)
    cv = pack_conv_module(layer.conv_module, prec)
    norms = {}
    for short, name in (("ffm", "norm_ff_macaron"), ("mha", "norm_mha"), ("conv", "norm_conv"), ("ff", "norm_ff"), ("final", "norm_final")):
        ln = getattr(layer, name)
        norms[short] = (ln.weight.detach(), ln.bias.detach())
    for g, b in norms.values():
        if g.dtype != torch.float32 or not g.is_contiguous() or not b.is_contiguous():
            raise TypeError("LayerNorm parameters must be contiguous float32")
    w = _c.LayerWeights()
    for short, (g, b) in norms.items():
        setattr(w, "ln_%s_g" % short, g.data_ptr())
        setattr(w, "ln_%s_b" % short, b.data_ptr())
    for pre, pk in (("ffm", ffm), ("ff", ff)):
        setattr(w, pre + "_w1", pk.w1.data_ptr())
        setattr(w, pre + "_w1_lo", _c.ptr(pk.w1_lo))
        setattr(w, pre + "_w2", pk.w2.data_ptr())
        setattr(w, pre + "_w2_lo", _c.ptr(pk.w2_lo))
        setattr(w, pre + "_b1", pk.b1.data_ptr())
        setattr(w, pre + "_b2", pk.b2.data_ptr())
        setattr(w, pre + "_w1f", _c.ptr(pk.w1f))
        setattr(w, pre + "_w2f", _c.ptr(pk.w2f))
        setattr(w, pre + "_w2n", _c.ptr(pk.w2n))
    w.qkv_wf, w.out_wf, w.pw1_wf, w.pw2_wf = _c.ptr(att.qkv_wf), _c.ptr(att.out_wf), _c.ptr(cv.pw1_wf), _c.ptr(cv.pw2_wf)
    w.qkv_w, w.qkv_w_lo, w.qkv_b = att.qkv_w.data_ptr(), _c.ptr(att.qkv_w_lo), att.qkv_b.data_ptr()
    w.pos_w, w.pos_w_lo = _c.ptr(att.pos_w), _c.ptr(att.pos_w_lo)
    w.out_w, w.out_w_lo, w.out_b = att.out_w.data_ptr(), _c.ptr(att.out_w_lo), att.out_b.data_ptr()
    w.bias_u, w.bias_v = _c.ptr(att.bias_u), _c.ptr(att.bias_v)
    w.pw1_w, w.pw1_w_lo, w.pw1_b = cv.pw1_w.data_ptr(), _c.ptr(cv.pw1_w_lo), cv.pw1_b.data_ptr()
    w.pw2_w, w.pw2_w_lo, w.pw2_b = cv.pw2_w.data_ptr(), _c.ptr(cv.pw2_w_lo), cv.pw2_b.data_ptr()
    w.dw_w, w.dw_b, w.bn_scale, w.bn_shift = cv.dw_w.data_ptr(), cv.dw_b.data_ptr(), cv.bn_scale.data_ptr(), cv.bn_shift.data_ptr()
    return w, (ffm, ff, att, cv, norms)


# ----------------------------------------------------------------------------------------------------------------------
# training packs: every dense layer needs its weight twice -- [N,K] for the forward product and [K,N] for the input gradient
# (both K-contiguous for cfm_gemm) -- rebuilt when the optimizer has stepped (the parameter's version counter moves)
# ----------------------------------------------------------------------------------------------------------------------
def matrix_t(w, prec):
    """[N,K] float weight -> the packs of its transpose [K,N]."""
    return matrix(w.detach().t(), prec)


def _train_cache(mod):
    pc = mod.__dict__.get("_pack_train")
    if pc is None:
        pc = PackCache()
        mod.__dict__["_pack_train"] = pc
    return pc


def pack_ffn_train(mod, prec):
    def build():
        w1, w1l = matrix(mod.w_1.weight, prec)
        w2, w2l = matrix(mod.w_2.weight, prec)
        w1t, w1tl = matrix_t(mod.w_1.weight, prec)
        w2t, w2tl = matrix_t(mod.w_2.weight, prec)
        return Packed(w1=w1, w1_lo=w1l, b1=f32(mod.w_1.bias), w2=w2, w2_lo=w2l, b2=f32(mod.w_2.bias), w1t=w1t, w1t_lo=w1tl, w2t=w2t, w2t_lo=w2tl)
    return _train_cache(mod).get([mod.w_1.weight, mod.w_1.bias, mod.w_2.weight, mod.w_2.bias], prec, build)


def pack_mhsa_train(mod, prec, relative):
    srcs = [mod.linear_q.weight, mod.linear_q.bias, mod.linear_k.weight, mod.linear_k.bias, mod.linear_v.weight, mod.linear_v.bias,
            mod.linear_out.weight, mod.linear_out.bias]
    if relative:
        srcs.append(mod.pos_bias_u)

    def build():
        wcat = torch.cat([mod.linear_q.weight.detach(), mod.linear_k.weight.detach(), mod.linear_v.weight.detach()], 0)
        qkv, qkvl = matrix(wcat, prec)
        qkvt, qkvtl = matrix_t(wcat, prec)
        out, outl = matrix(mod.linear_out.weight, prec)
        outt, outtl = matrix_t(mod.linear_out.weight, prec)
        bq = mod.linear_q.bias.detach().float()
        if relative:                                   # q + pos_bias_u (attention.py:81) rides in the projection's bias; the batch path's
            bq = bq + mod.pos_bias_u.detach().float().reshape(-1)      # positional term is softmax-invariant and is not evaluated
        qkv_b = torch.cat([bq, mod.linear_k.bias.detach().float(), mod.linear_v.bias.detach().float()], 0).contiguous()
        return Packed(qkv_w=qkv, qkv_w_lo=qkvl, qkv_t=qkvt, qkv_t_lo=qkvtl, qkv_b=qkv_b, out_w=out, out_w_lo=outl, out_t=outt, out_t_lo=outtl,
                      out_b=f32(mod.linear_out.bias))
    return _train_cache(mod).get(srcs, prec, build)


def pack_conv_module_train(mod, prec):
    bn = mod.norm
    srcs = [mod.pointwise_conv1.weight, mod.pointwise_conv1.bias, mod.depthwise_conv.weight, mod.depthwise_conv.bias, bn.weight, bn.bias,
            mod.pointwise_conv2.weight, mod.pointwise_conv2.bias]

    def build():
        D = mod.pointwise_conv2.weight.shape[0]
        dev = mod.pointwise_conv2.weight.device
        idx = glu_interleave_index(D, dev)
        w1 = mod.pointwise_conv1.weight.detach()[:, :, 0][idx]
        b1 = (mod.pointwise_conv1.bias.detach() if mod.pointwise_conv1.bias is not None else torch.zeros(2 * D, device=dev))[idx]
        w2 = mod.pointwise_conv2.weight.detach()[:, :, 0]
        pw1, pw1l = matrix(w1, prec)
        pw1t, pw1tl = matrix_t(w1, prec)
        pw2, pw2l = matrix(w2, prec)
        pw2t, pw2tl = matrix_t(w2, prec)
        dwb = mod.depthwise_conv.bias.detach() if mod.depthwise_conv.bias is not None else torch.zeros(D, device=dev)
        gamma = bn.weight.detach() if bn.weight is not None else torch.ones(D, device=dev)
        beta = bn.bias.detach() if bn.bias is not None else torch.zeros(D, device=dev)
        return Packed(pw1_w=pw1, pw1_w_lo=pw1l, pw1_t=pw1t, pw1_t_lo=pw1tl, pw1_b=f32(b1), pw2_w=pw2, pw2_w_lo=pw2l, pw2_t=pw2t, pw2_t_lo=pw2tl,
                      pw2_b=f32(mod.pointwise_conv2.bias), dw_w=f32(mod.depthwise_conv.weight.detach()[:, 0, :]), dw_b=f32(dwb), gamma=f32(gamma),
                      beta=f32(beta), idx=idx)
    return _train_cache(mod).get(srcs, prec, build)


class _LayerPackPlan:
    """The 8 weight matrices of one conformer block (two feed-forwards, fused q|k|v, out-projection, the GLU-interleaved pointwise-conv-1,
    pointwise-conv-2) as ONE cfm_pack_matrices launch per optimizer step: the destination tensors and the job table are built once (the
    parameters' addresses do not move: the optimizer writes in place), a rebuild is one kernel + the three small bias vectors."""

    def __init__(self, layer, prec, relative):
        ffm, att, cv, ff = layer.feed_forward_macaron, layer.self_attn, layer.conv_module, layer.feed_forward
        dev = ffm.w_1.weight.device
        D = cv.pointwise_conv2.weight.shape[0]
        self.idx = glu_interleave_index(D, dev)
        self.relative = relative
        self.srcs = [ffm.w_1.weight, ffm.w_2.weight, att.linear_q.weight, att.linear_k.weight, att.linear_v.weight, att.linear_out.weight,
                     cv.pointwise_conv1.weight, cv.pointwise_conv2.weight, ff.w_1.weight, ff.w_2.weight]
        self.ptrs = tuple(t.data_ptr() for t in self.srcs)
        for t in self.srcs:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise TypeError("the training packs need contiguous float32 parameters")
        wdt = torch.bfloat16 if prec.split else prec.w_dtype

        def rows_of(t, n, k, perm=None):
            r = t.data_ptr() + torch.arange(n, device=dev, dtype=torch.int64) * (k * 4)
            return r if perm is None else r[perm]

        def alloc(n, k):
            mk = lambda: torch.empty((n, k), dtype=wdt, device=dev)
            return mk(), (mk() if prec.split else None), torch.empty((k, n), dtype=wdt, device=dev), (torch.empty((k, n), dtype=wdt, device=dev) if prec.split else None)

        FF = ffm.w_1.weight.shape[0]
        specs = [("ffm_w1", rows_of(ffm.w_1.weight, FF, D), FF, D), ("ffm_w2", rows_of(ffm.w_2.weight, D, FF), D, FF),
                 ("qkv", torch.cat([rows_of(att.linear_q.weight, D, D), rows_of(att.linear_k.weight, D, D), rows_of(att.linear_v.weight, D, D)]), 3 * D, D),
                 ("out", rows_of(att.linear_out.weight, D, D), D, D),
                 ("pw1", rows_of(cv.pointwise_conv1.weight, 2 * D, D, self.idx), 2 * D, D), ("pw2", rows_of(cv.pointwise_conv2.weight, D, D), D, D),
                 ("ff_w1", rows_of(ff.w_1.weight, FF, D), FF, D), ("ff_w2", rows_of(ff.w_2.weight, D, FF), D, FF)]
        self.keep, self.out, jobs, tile0 = [], {}, [], 0
        for name, rows, n, k in specs:
            if n % 8 or k % 8:
                raise ValueError("training packs: matrix dims must be multiples of 8")
            rows = rows.contiguous()
            w, wl, wt, wtl = alloc(n, k)
            self.keep.append(rows)
            self.out[name] = (w, wl, wt, wtl)
            jobs.append([rows.data_ptr(), n, k, w.data_ptr(), _c.ptr(wl) or 0, wt.data_ptr(), _c.ptr(wtl) or 0, tile0])
            tile0 += ((n + 63) // 64) * ((k + 63) // 64)
        self.tiles = tile0
        self.jobs = torch.tensor(jobs, dtype=torch.int64).to(dev)
        self.prec = prec

    def valid_for(self, prec):
        return prec.name == self.prec.name and tuple(t.data_ptr() for t in self.srcs) == self.ptrs

    def run(self, layer, launch=True, vectors=None):
        """launch=False: the matrices were packed by a launch covering several blocks (_StackPackPlan); vectors = (qkv_b, pw1_b) likewise."""
        prec = self.prec
        if launch:
            _c.check(_c.lib().cfm_pack_matrices(self.jobs.data_ptr(), self.jobs.shape[0], self.tiles, _c.BF16 if prec.split else prec.w_code, 1 if prec.split else 0,
                                                _c.stream()), "cfm_pack_matrices")
        ffm, att, cv, ff = layer.feed_forward_macaron, layer.self_attn, layer.conv_module, layer.feed_forward
        o = self.out

        def ffn(mod, pre):
            (w1, w1l, w1t, w1tl), (w2, w2l, w2t, w2tl) = o[pre + "_w1"], o[pre + "_w2"]
            return Packed(w1=w1, w1_lo=w1l, b1=f32(mod.w_1.bias), w2=w2, w2_lo=w2l, b2=f32(mod.w_2.bias), w1t=w1t, w1t_lo=w1tl, w2t=w2t, w2t_lo=w2tl)
        if vectors is not None:
            qkv_b = vectors[0]
        else:
            bq = att.linear_q.bias.detach().float()
            if self.relative:
                bq = bq + att.pos_bias_u.detach().float().reshape(-1)
            qkv_b = torch.cat([bq, att.linear_k.bias.detach().float(), att.linear_v.bias.detach().float()], 0).contiguous()
        (qkv, qkvl, qkvt, qkvtl), (out, outl, outt, outtl) = o["qkv"], o["out"]
        pa = Packed(qkv_w=qkv, qkv_w_lo=qkvl, qkv_t=qkvt, qkv_t_lo=qkvtl, qkv_b=qkv_b, out_w=out, out_w_lo=outl, out_t=outt, out_t_lo=outtl,
                    out_b=f32(att.linear_out.bias))
        D = cv.pointwise_conv2.weight.shape[0]
        dev = qkv.device
        bn = cv.norm
        if vectors is not None:
            b1 = vectors[1]
        else:
            b1 = (cv.pointwise_conv1.bias.detach() if cv.pointwise_conv1.bias is not None else torch.zeros(2 * D, device=dev))[self.idx]
        dwb = cv.depthwise_conv.bias.detach() if cv.depthwise_conv.bias is not None else torch.zeros(D, device=dev)
        gamma = bn.weight.detach() if bn.weight is not None else torch.ones(D, device=dev)
        beta = bn.bias.detach() if bn.bias is not None else torch.zeros(D, device=dev)
        (pw1, pw1l, pw1t, pw1tl), (pw2, pw2l, pw2t, pw2tl) = o["pw1"], o["pw2"]
        pc = Packed(pw1_w=pw1, pw1_w_lo=pw1l, pw1_t=pw1t, pw1_t_lo=pw1tl, pw1_b=f32(b1), pw2_w=pw2, pw2_w_lo=pw2l, pw2_t=pw2t, pw2_t_lo=pw2tl,
                    pw2_b=f32(cv.pointwise_conv2.bias), dw_w=f32(cv.depthwise_conv.weight.detach()[:, 0, :]), dw_b=f32(dwb), gamma=f32(gamma),
                    beta=f32(beta), idx=self.idx)
        return ffn(ffm, "ffm"), pa, pc, ffn(ff, "ff")


def pack_layer_train(layer, prec, relative):
    """(macaron FFN, attention, conv module, FFN) training packs of one block, the 8 matrices through one cfm_pack_matrices launch.
    Same values as pack_ffn_train / pack_mhsa_train / pack_conv_module_train (tests compare them)."""
    st = layer.__dict__.get("_pack_layer_train")
    srcs = layer.__dict__.get("_pack_layer_srcs")
    if srcs is None:                                   # every parameter a pack depends on (LayerNorm parameters are read in place)
        srcs = [p for n, p in layer.named_parameters() if not n.startswith("norm_")]
        layer.__dict__["_pack_layer_srcs"] = srcs
    key = _key(srcs, prec)
    if st is not None and st[0] == key:
        return st[2]
    with torch.no_grad():
        plan = st[1] if st is not None and st[1].valid_for(prec) else _LayerPackPlan(layer, prec, relative)
        val = plan.run(layer)
    layer.__dict__["_pack_layer_train"] = (key, plan, val)
    return val


import os as _os
# each feed-forward's training forward as ONE launch (csrc/ffn.hip cfm_ffn_train_forward): built, parity-tested (the reference gradient goldens pass through
# it), and measured SLOWER inside the config-3 step -- 8.16 against 7.94 ms per optimizer step: alone the launch is 25 us against 41 for LayerNorm + W1 + W2,
# but with the 8.7 KB per row it must write for the backward (pre-activation, activation, LayerNorm output) from 107 workgroups' main loops it loses -- opt-in
FFN_TRAIN_FUSED = _os.environ.get("CFM_FFN_TRAIN_FUSED", "0") != "0"


class _StackPackPlan:
    """The training packs of ALL blocks of an encoder in two launches per optimizer step -- the blocks' matrix jobs in one table
    (cfm_pack_matrices), their two small gathered vectors (fused q|k|v bias + pos_bias_u, interleaved pointwise-conv-1 bias) through
    element-pointer tables (cfm_pack_vectors) -- instead of one matrix launch and ~4 torch operations per block (12 blocks: 60 launches)."""

    def __init__(self, layers, prec, relative):
        self.plans = [_LayerPackPlan(l, prec, relative) for l in layers]
        dev = self.plans[0].jobs.device
        tables, off = [], 0
        for pl in self.plans:
            j = pl.jobs.clone()
            j[:, 7] += off
            off += pl.tiles
            tables.append(j)
        self.jobs, self.tiles = torch.cat(tables, 0).contiguous(), off
        pa, pb, self.vec_slices, n = [], [], [], 0
        el = lambda t: t.data_ptr() + 4 * torch.arange(t.numel(), device=dev, dtype=torch.int64)
        for l, pl in zip(layers, self.plans):
            att, cv = l.self_attn, l.conv_module
            D = cv.pointwise_conv2.weight.shape[0]
            for t in (att.linear_q.bias, att.linear_k.bias, att.linear_v.bias, cv.pointwise_conv1.bias) + ((att.pos_bias_u,) if relative else ()):
                if t is None or t.dtype != torch.float32 or not t.is_contiguous():
                    raise TypeError("the stack pack needs contiguous float32 bias parameters")
            zeros = torch.zeros(D, device=dev, dtype=torch.int64)
            pa += [el(att.linear_q.bias), el(att.linear_k.bias), el(att.linear_v.bias), el(cv.pointwise_conv1.bias)[pl.idx]]
            pb += [el(att.pos_bias_u) if relative else zeros, zeros, zeros, zeros, zeros]
            self.vec_slices.append((n, n + 3 * D, n + 5 * D))
            n += 5 * D
        self.pa, self.pb = torch.cat(pa).contiguous(), torch.cat(pb).contiguous()
        self.vec = torch.empty(n, dtype=torch.float32, device=dev)
        self.prec, self.layers, self.val = prec, list(layers), None
        self.srcs = [t for pl in self.plans for t in pl.srcs]
        # fragment-major packs of the feed-forwards for the one-launch training forward (csrc/ffn.hip cfm_ffn_train_forward): all of them in one launch
        self.frag_jobs, self.frags = None, None
        ffns = [f for l in layers for f in (l.feed_forward_macaron, l.feed_forward)]
        FF, D = ffns[0].w_1.weight.shape
        if FFN_TRAIN_FUSED and not prec.split and _c.lib().cfm_ffn_train_supported(D, FF) == 1 and all(tuple(f.w_1.weight.shape) == (FF, D) for f in ffns):
            self.frags = [(torch.empty(FF * D, dtype=prec.w_dtype, device=dev), torch.empty(FF * D, dtype=prec.w_dtype, device=dev)) for _ in ffns]
            self.frag_jobs = torch.tensor([[f.w_1.weight.data_ptr(), f.w_2.weight.data_ptr(), a.data_ptr(), b.data_ptr()] for f, (a, b) in zip(ffns, self.frags)],
                                          dtype=torch.int64).to(dev)
            self.frag_dims = (D, FF)
        self.ptrs = tuple(t.data_ptr() for l in layers for t in l.parameters())

    def valid_for(self, layers, prec):
        return (prec.name == self.prec.name and len(layers) == len(self.layers) and all(a is b for a, b in zip(layers, self.layers)) and
                tuple(t.data_ptr() for l in layers for t in l.parameters()) == self.ptrs)

    def run(self):
        prec = self.prec
        _c.check(_c.lib().cfm_pack_matrices(self.jobs.data_ptr(), self.jobs.shape[0], self.tiles, _c.BF16 if prec.split else prec.w_code, 1 if prec.split else 0,
                                            _c.stream()), "cfm_pack_matrices")
        _c.check(_c.lib().cfm_pack_vectors(self.pa.data_ptr(), self.pb.data_ptr(), self.vec.data_ptr(), self.vec.numel(), _c.stream()), "cfm_pack_vectors")
        if self.frag_jobs is not None:
            _c.check(_c.lib().cfm_pack_ffn_fragments(self.frag_jobs.data_ptr(), self.frag_jobs.shape[0], self.frag_dims[0], self.frag_dims[1], prec.w_code,
                                                     _c.stream()), "cfm_pack_ffn_fragments")
        if self.val is None:        # the destinations never move (valid_for checks the sources' addresses): the same Packed objects every step, so
            # whoever caches on their identity (the stack's ctypes weight structs, cfm/autograd.py) keeps its cache across optimizer steps
            self.val = tuple(pl.run(l, launch=False, vectors=(self.vec[a:b], self.vec[b:c])) for l, pl, (a, b, c) in zip(self.layers, self.plans, self.vec_slices))
            if self.frags is not None:
                for i, pks in enumerate(self.val):                         # (macaron FFN, attention, conv module, FFN)
                    pks[0].w1f, pks[0].w2f = self.frags[2 * i]
                    pks[3].w1f, pks[3].w2f = self.frags[2 * i + 1]
        return self.val


def pack_stack_train(owner, layers, prec, relative, flat=False):
    """((macaron FFN, attention, conv module, FFN) training packs) per block for the whole stack, two launches when anything changed.
    flat: every parameter lives in the data-parallel trainer's flat buffer and changes only through its step (which bumps the pack epoch) --
    the per-tensor version walk (~300 tensors) is skipped."""
    st = owner.__dict__.get("_pack_stack_train")
    if flat and st is not None and st[0][:3] == ("flat", prec.name, _EPOCH[0]) and st[1].valid_for(layers, prec):
        return st[2]
    if flat:
        key = ("flat", prec.name, _EPOCH[0])
    else:
        key = ("walk",) + _key([p for l in layers for n, p in l.named_parameters() if not n.startswith("norm_")], prec)
        if st is not None and st[0] == key:
            return st[2]
    with torch.no_grad():
        plan = st[1] if st is not None and st[1].valid_for(layers, prec) else _StackPackPlan(layers, prec, relative)
        val = plan.run()
    owner.__dict__["_pack_stack_train"] = (key, plan, val)
    return val


def pack_subsampling_train(mod, prec):
    c1, c2, lin = mod.conv[0], mod.conv[2], mod.out[0]

    def build():
        C = c1.weight.shape[0]
        w2f = c2.weight.detach().permute(0, 2, 3, 1).reshape(C, 9 * C)                      # [co][kt][kf][ci]
        Dout, CF = lin.weight.shape
        Fp = CF // C
        wlf = lin.weight.detach().reshape(Dout, C, Fp).permute(0, 2, 1).reshape(Dout, Fp * C)   # feature order f*C + c
        w2, w2l = matrix(w2f, prec)
        w2t, w2tl = matrix_t(w2f, prec)
        wl, wll = matrix(wlf, prec)
        wlt, wltl = matrix_t(wlf, prec)
        return Packed(w1=f32(c1.weight.detach().reshape(C, 9).t()), b1=f32(c1.bias), w2=w2, w2_lo=w2l, w2t=w2t, w2t_lo=w2tl, b2=f32(c2.bias),
                      wl=wl, wl_lo=wll, wlt=wlt, wlt_lo=wltl, bl=f32(lin.bias), C=C, Fp=Fp)
    return _train_cache(mod).get([c1.weight, c1.bias, c2.weight, c2.bias, lin.weight, lin.bias], prec, build)


def pack_ctc_train(mod, prec):
    def build():
        V, D = mod.ctc_lo.weight.shape
        Vp = (V + 7) // 8 * 8                          # cfm_gemm_tn wants N % 8 == 0; pad rows are zero and never read by the CTC kernels
        w = torch.zeros((Vp, D), dtype=torch.float32, device=mod.ctc_lo.weight.device)
        w[:V] = mod.ctc_lo.weight.detach()
        b = torch.zeros((Vp,), dtype=torch.float32, device=w.device)
        b[:V] = mod.ctc_lo.bias.detach()
        wm, wlo = matrix(w, prec)
        wt, wtlo = matrix_t(w, prec)
        return Packed(w=wm, w_lo=wlo, wt=wt, wt_lo=wtlo, b=b, V=V, Vp=Vp)
    return _train_cache(mod).get([mod.ctc_lo.weight, mod.ctc_lo.bias], prec, build)
