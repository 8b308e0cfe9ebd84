"""One conformer block, MI355X-native.

Drop-in for the reference's ``src/encoder_layer.py`` (constructor arguments, child-module and parameter names, and the
``forward(inputs, inputs_attn_mask, pos_embed, inputs_pad_mask, attn_cache, cnn_cache)`` ->
``(out, inputs_attn_mask, new_attn_cache, new_cnn_cache)`` contract of encoder_layer.py:11-71).

The whole block -- LN, 1/2 macaron FFN, LN, MHSA, LN, convolution module, LN, 1/2 FFN, final LN with every residual -- is
ONE call into libconformer_gfx950 (``cfm_encoder_layer_forward``): 5 kernel launches (row chains; 17 on the general path) enqueued from C++ with no host
synchronisation, the residual stream in f32, every bias / activation / GLU / mask / residual add fused into a GEMM
epilogue, and LayerNorm writing the next GEMM's operand dtype directly.  The child modules (feedforward / attention /
convolution) only own the parameters here; called on their own they run the same kernels op by op.
"""
import ctypes

import torch
import torch.nn as nn

import cfm
from cfm import packing
from attention import MultiHeadSelfAttentionModule, RelativeMultiHeadSelfAttentionModule, _mask_args
from convolution import ConvolutionModule
from feedforward import PositionwiseFeedForwardModule, _inference_only

_ABSENT = torch.ones((0, 0, 0), dtype=torch.bool)
PAIR_MAX_ROWS = 4096       # include/cfm.h CFM_PAIR_MAX_ROWS: the pair-split feed-forward of the D = 512 row chains (one workgroup per CU up to 128 row tiles)
SPLIT_FFN_FEW_ROWS = True  # STREAMING steps of <= 1536 rows run their feed-forwards split over FF/256 workgroups per row tile (csrc/ffnsplit.hip).  Only
# the streaming entry points ask for it (split_ffn=True): a whole-utterance forward keeps one algorithm for every batch size, so that a batch shard
# reproduces the batch bit for bit


import os as _os
_SPLIT_MAX_ROWS = int(_os.environ.get("CFM_FFSPLIT_MAX_ROWS", "1536"))     # the library reads the same variable (csrc/encoder.cpp): experiments only


def split_rows(M, D, FF):
    """True where cfm_encoder_layer_forward takes the split feed-forward when given the slabs (cfm.h CFM_FFSPLIT_MAX_ROWS)."""
    return M <= _SPLIT_MAX_ROWS and D == 256 and FF % 256 == 0 and FF > 0        # measured crossover (scripts/bench_small_batch.py): 996 rows -21 %, 1992 rows +4 %


CHAIN_BLOCKS = True        # the final chain of block i also runs the macaron chain of block i+1 (one launch and one residual round trip less)
MERGE_ATTENTION = False    # attention as the input stage of the conv-in chain (3 launches per block instead of 4): built, parity-tested and
                           # measured SLOWER at config 2 (27.7 us vs 9.7 + 13.1 us, rowchain.hip) -- opt-in


class ConformerEncoderLayer(nn.Module):

    def __init__(self, encoder_dim, kernel_size, feedforward_dropout, attention_dropout, hidden_dim, num_heads, use_relative):
        super().__init__()
        self.feed_forward = PositionwiseFeedForwardModule(encoder_dim, feedforward_dropout, hidden_dim)
        attn_cls = RelativeMultiHeadSelfAttentionModule if use_relative else MultiHeadSelfAttentionModule
        self.self_attn = attn_cls(encoder_dim, num_heads, attention_dropout)
        self.conv_module = ConvolutionModule(encoder_dim, kernel_size, hidden_dim)     # sic: lands in `bias` (quirk Q1)
        self.feed_forward_macaron = PositionwiseFeedForwardModule(encoder_dim, feedforward_dropout, hidden_dim)
        self.norm_ff = nn.LayerNorm(encoder_dim, eps=1e-5)
        self.norm_ff_macaron = nn.LayerNorm(encoder_dim, eps=1e-5)
        self.norm_mha = nn.LayerNorm(encoder_dim, eps=1e-5)
        self.norm_conv = nn.LayerNorm(encoder_dim, eps=1e-5)
        self.norm_final = nn.LayerNorm(encoder_dim, eps=1e-5)
        self.dropout = nn.Dropout(feedforward_dropout)
        self.use_relative = bool(use_relative)
        self.encoder_dim, self.hidden_dim, self.num_heads, self.kernel_size = encoder_dim, hidden_dim, num_heads, kernel_size
        self.return_cache = True           # the reference always materialises cat(k,v) (attention.py:76)
        self._fused = None                 # (key, LayerWeights, keepalive)

    def _apply(self, fn, *a, **kw):        # .to() / .cuda() / .half(): drop every packed copy
        self._fused = None
        return super()._apply(fn, *a, **kw)

    def _weights(self, prec):
        # the tensor list is rebuilt on every call: replacing a Parameter OBJECT (layer.norm_ff.weight = nn.Parameter(...), pruning /
        # reparametrisation, swapping a submodule) must change the key too; walking ~60 tensors is cheap next to a 5-launch block
        plist = list(self.parameters()) + list(self.buffers())
        key = (prec.name, packing._EPOCH[0]) + tuple((t.data_ptr(), t._version) for t in plist)   # in-place updates bump _version; `p.data = ...` moves the pointer
        if self._fused is None or self._fused[0] != key:
            struct, keep = packing.layer_weight_struct(self, prec)
            self._fused = (key, struct, keep)
        return self._fused[1]

    def chain_ready(self, prec):
        """True when this block's weight struct carries every fragment-major pack of the row-chain path (cfm_encoder_layer_forward then
        takes that path, a precondition for chaining consecutive blocks)."""
        w = self._weights(prec)
        return all(getattr(w, f) for f in ("ffm_w1f", "ffm_w2n", "ff_w1f", "ff_w2n", "qkv_wf", "out_wf", "pw1_wf", "pw2_wf"))

    def fused_forward(self, x, attn_mask, pos_embed, pad_mask, attn_cache, xn_ready=False, next_norm=None, out=None,
                      want_cache=True, pos_proj=None, pos_shared=False, after=None, ring=None, conv_cache=None, chain_next=None,
                      macaron_done=False, split_ffn=False):
        """x (B,T,D) float32 on an MI355X -> (norm_final(block(x)), new_attn_cache | None).  ``x`` is not modified.
        ring = (kv_ring f32 [B,H,ring_T,2dk], offsets int32 [B]): per-stream streaming state (include/cfm.h cfm_layer_io.kv_ring);
        attn_mask is then the (B,1,ring_T) slot mask and pos_embed the B*ring_T positional rows.  conv_cache f32 [B,K-1,D]: the
        opt-in causal convolution's left context (only with conv_module.causal).
        chain_next = (next block, its output buffer): this block's last launch also runs the NEXT block's macaron chain (cfm.h
        cfm_layer_io.next_w); the next block is then called with macaron_done=True and the same output buffer, and THIS call's
        returned tensor does not hold the block output.  Only ConformerEncoder._run_blocks uses it."""
        _inference_only(self, "ConformerEncoderLayer.fused_forward")
        cfm.require_hip(x)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        prec = cfm.resolve_precision(self)
        w = self._weights(prec)
        B, T, D = x.shape
        H, FF = self.num_heads, self.hidden_dim
        dk = D // H
        dev = x.device
        M = B * T

        if ring is not None:
            attn_cache, want_cache = None, False
        have_cache = attn_cache is not None and attn_cache.dim() == 4 and attn_cache.size(0) > 0
        cache = attn_cache.to(device=dev, dtype=torch.float32).contiguous() if have_cache else None
        Tc = cache.size(2) if have_cache else 0
        Tk = Tc + T if ring is None else ring[0].size(2)
        if have_cache and tuple(cache.shape) != (B, H, Tc, 2 * dk):
            raise RuntimeError("attn_cache of shape %s, expected (%d,%d,Tc,%d)" % (tuple(cache.shape), B, H, 2 * dk))
        new_cache = torch.empty((B, H, Tk, 2 * dk), dtype=torch.float32, device=dev) if (have_cache or want_cache) else None

        pos = None
        R = 0
        if self.use_relative:
            pos = pos_embed.reshape(-1, D)
            pos = (pos if pos.dtype == torch.float32 else pos.float()).contiguous()
            cfm.require_hip(pos)
            R = pos.size(0)

        m8, (m_sb, m_sq) = _mask_args(attn_mask, B, T, Tk)
        keep = None
        if pad_mask is not None and pad_mask.dim() >= 3 and pad_mask.size(2) > 0:
            keep = cfm.as_u8_mask(pad_mask).reshape(-1)
            if keep.numel() != M:
                raise RuntimeError("pad mask %s does not match inputs %s" % (tuple(pad_mask.shape), tuple(x.shape)))
        cfm.require_hip(m8, keep)

        adt = prec.act_dtype
        s = cfm.LayerScratch()
        s.xn = cfm.scratch("xn", M * D, adt, dev).data_ptr()
        s.hid = cfm.scratch("hid", M * FF, adt, dev).data_ptr()
        s.qkv = cfm.scratch("qkv", M * 3 * D, adt, dev).data_ptr()
        s.pos = cfm.scratch("pos", max(R, 1) * D, adt, dev).data_ptr()
        s.ctx = cfm.scratch("ctx", M * D, adt, dev).data_ptr()
        s.glu = cfm.scratch("glu", M * D, adt, dev).data_ptr()
        s.dw = cfm.scratch("dw", M * D, adt, dev).data_ptr()
        if MERGE_ATTENTION and D == 256 and H == 4 and T <= 256 and adt != torch.float32:
            # transposed values for the attention stage of the conv-in chain (cfm.h cfm_layer_scratch.vt); key columns past T are read, never written
            s.vt, s.vt_ld = cfm.scratch("vt", B * D * 256, adt, dev, zero=True).data_ptr(), 256
        if split_ffn and SPLIT_FFN_FEW_ROWS and split_rows(M, D, FF) and adt != torch.float32 and chain_next is None and not macaron_done:
            # partial slabs of the feed-forward split over FF (cfm.h cfm_layer_scratch.psum, csrc/ffnsplit.hip): few rows, e.g. a streaming step
            s.psum, s.psum_splits = cfm.scratch("psum", (FF // 256) * M * D, torch.float32, dev).data_ptr(), FF // 256
        elif M <= PAIR_MAX_ROWS and cfm.rowchain_pair_supported(D, FF, prec) and chain_next is None and not macaron_done:
            # D = 512, at most 128 row tiles: feed-forwards split over workgroup pairs -- two partial slabs + the parked rows of the final chain
            s.psum, s.psum_splits = cfm.scratch("psum", 3 * M * D, torch.float32, dev).data_ptr(), 3
        io = cfm.LayerIO()
        io.B, io.T, io.D, io.H, io.FF, io.ktaps = B, T, D, H, FF, self.kernel_size
        io.act_dtype, io.w_dtype = prec.act_code, prec.w_code
        io.attn_mask, io.am_sb, io.am_sq = cfm.ptr(m8), m_sb, m_sq
        io.pad_valid = cfm.ptr(keep)
        io.pos_embed, io.pos_rows = cfm.ptr(pos), R
        if pos_proj is not None:                       # (tensor view [R, D] of the driver's all-blocks projection, row stride)
            io.pos_proj, io.pos_proj_ld = pos_proj[0].data_ptr(), pos_proj[1]
        io.attn_cache, io.cache_T = cfm.ptr(cache), Tc
        io.pos_shared = 1 if pos_shared else 0
        if after is not None:                          # (gain, bias, f32 output [B,T,D]): the encoder's after_norm in the final chain
            io.after_g, io.after_b, io.after_out = after[0].data_ptr(), after[1].data_ptr(), after[2].data_ptr()
        io.new_cache = cfm.ptr(new_cache)
        if ring is not None:
            kv, offs = ring
            if kv.dtype != torch.float32 or tuple(kv.shape) != (B, H, Tk, 2 * dk) or not kv.is_contiguous() or offs.dtype != torch.int32 or offs.numel() != B:
                raise RuntimeError("ring: kv must be contiguous float32 (B,H,ring_T,2dk) and offsets int32 (B,)")
            cfm.require_hip(kv, offs)
            io.kv_ring, io.stream_offset, io.ring_T = kv.data_ptr(), offs.data_ptr(), Tk
        if getattr(self.conv_module, "causal", False):
            io.causal_conv = 1
            if conv_cache is not None:
                if conv_cache.dtype != torch.float32 or tuple(conv_cache.shape) != (B, self.kernel_size - 1, D) or not conv_cache.is_contiguous():
                    raise RuntimeError("conv_cache must be contiguous float32 (B, kernel_size-1, D)")
                cfm.require_hip(conv_cache)
                io.conv_cache = conv_cache.data_ptr()
        elif conv_cache is not None:
            raise RuntimeError("a conv cache needs the opt-in causal convolution (conv_module.causal = True)")
        if out is None:
            out = torch.empty_like(x)
        if chain_next is not None:
            nxt_w = chain_next[0]._weights(prec)
            if chain_next[1].data_ptr() == out.data_ptr() or chain_next[1].shape != out.shape or chain_next[1].dtype != torch.float32:
                raise RuntimeError("chain_next: the next block's output buffer must be a distinct float32 tensor of the same shape")
            io.next_w, io.next_x_out = ctypes.cast(ctypes.pointer(nxt_w), ctypes.c_void_p), chain_next[1].data_ptr()
        io.macaron_done = 1 if macaron_done else 0
        ng = nb = None
        if next_norm is not None:
            ng, nb = next_norm.weight.data_ptr(), next_norm.bias.data_ptr()
        cfm.check(cfm.lib().cfm_encoder_layer_forward(ctypes.byref(w), ctypes.byref(s), ctypes.byref(io), x.data_ptr(),
                                                      out.data_ptr(), 1 if xn_ready else 0, ng, nb, cfm.stream()),
                  "cfm_encoder_layer_forward")
        return out, new_cache

    def train_forward(self, inputs, inputs_attn_mask, inputs_pad_mask):
        """module.train(): the block as ONE autograd node (cfm/autograd.py EncoderLayerFn) -- BatchNorm batch statistics, every
        parameter's gradient returned from its backward.  encoder_layer.py:49-71 with the shared nn.Dropout(feedforward_dropout)."""
        from cfm import autograd as ag
        cfm.require_hip(inputs)
        B, T, D = inputs.shape
        m8, m_str = _mask_args(inputs_attn_mask, B, T, T)
        keep = None
        if inputs_pad_mask is not None and inputs_pad_mask.dim() >= 3 and inputs_pad_mask.size(2) > 0:
            keep = cfm.as_u8_mask(inputs_pad_mask).reshape(-1)
            if keep.numel() != B * T:
                raise RuntimeError("pad mask %s does not match inputs %s" % (tuple(inputs_pad_mask.shape), tuple(inputs.shape)))
        leaf = self.__dict__.get("_flat_leaf")         # set by trainer.DataParallelTrainer: the block's parameters as ONE autograd leaf
        params = (leaf,) if leaf is not None else tuple(self.parameters())
        return ag.EncoderLayerFn.apply(inputs, self, cfm.resolve_precision(self), m8, m_str, keep, *params)

    def forward(self, inputs, inputs_attn_mask, pos_embed, inputs_pad_mask=_ABSENT, attn_cache=_ABSENT, cnn_cache=_ABSENT):
        if cfm.check_mode(self, "ConformerEncoderLayer"):
            if attn_cache is not None and attn_cache.dim() == 4 and attn_cache.size(0) > 0:
                raise NotImplementedError("ConformerEncoderLayer: a KV cache in train mode (streaming is inference-only)")
            out = self.train_forward(inputs, inputs_attn_mask, inputs_pad_mask)
            return (out, inputs_attn_mask, torch.zeros((0, 0, 0, 0), dtype=torch.float32, device=inputs.device),
                    torch.zeros((0, 0, 0), dtype=inputs.dtype, device=inputs.device))
        out, new_attn_cache = self.fused_forward(inputs, inputs_attn_mask, pos_embed, inputs_pad_mask, attn_cache,
                                                 want_cache=self.return_cache)
        if new_attn_cache is None:
            new_attn_cache = torch.zeros((0, 0, 0, 0), dtype=torch.float32, device=inputs.device)
        new_cnn_cache = torch.zeros((0, 0, 0), dtype=inputs.dtype, device=inputs.device)    # convolution.py:39
        return out.to(inputs.dtype), inputs_attn_mask, new_attn_cache, new_cnn_cache
