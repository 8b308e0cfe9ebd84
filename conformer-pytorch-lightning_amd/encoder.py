"""Conformer encoder driver, MI355X-native: subsampling front-end, N conformer blocks, final LayerNorm; whole-utterance
``forward`` and the streaming pair ``forward_chunk`` / ``forward_chunk_by_chunk``.

Mirrors the public surface of the reference's ``src/encoder.py`` (constructor arguments :11-27, ``forward`` :54-75,
``forward_chunk`` :78-123, ``forward_chunk_by_chunk`` :125-153) so it can stand in for it; the reference's own
encoder.py also runs unchanged on top of the drop-in modules of this directory.  Differences are in HOW, not WHAT:

* masks are built on the device from the lengths (one launch each; the subsampled padding mask directly as
  ``6 + 4 j < len``), bit-identical to the reference's arange / slicing / python row loop;
* each block is one C call (2 launches on the row-chain path with consecutive blocks chained -- attention, then conv-in + depthwise + final + the next block's macaron chain --, 4 without, 17 on the general one); consecutive blocks hand over the already-normalised macaron-FFN operand
  (norm_final of block i and norm_ff_macaron of block i+1 are chained in registers);
* the batch path does not materialise the per-layer cat(k,v) cache that the reference builds and discards.
"""
import torch
import torch.nn as nn

import cfm
from attention import PositionalEncoding, RelativePositionalEncoding
from convolution import ConvolutionSubSampling
from encoder_layer import ConformerEncoderLayer
from utils import make_attn_mask, make_pad_mask

_NO_MASK = torch.ones((0, 0, 0))


class ConformerEncoder(nn.Module):

    def __init__(self, input_dim, kernel_size, encoder_dim, dropout, attention_dropout, pos_enc_dropout, hidden_dim,
                 num_heads, encoder_num_layers, cmvn=None, max_len=5000, use_relative=False, use_dynamic_chunk_size=False,
                 use_dynamic_left_chunk=False, static_chunk_size=-1):
        super().__init__()
        pe_cls = RelativePositionalEncoding if use_relative else PositionalEncoding
        self.position_encoding = pe_cls(encoder_dim, pos_enc_dropout, max_len)
        self.embed = ConvolutionSubSampling(input_dim=input_dim, output_dim=encoder_dim, pos_enc=self.position_encoding)
        self.encoders = nn.ModuleList([
            ConformerEncoderLayer(encoder_dim, kernel_size, dropout, attention_dropout, hidden_dim, num_heads, use_relative)
            for _ in range(encoder_num_layers)])
        self.encoder_dim = encoder_dim
        self.after_norm = nn.LayerNorm(encoder_dim, eps=1e-5)
        self.global_cmvn = cmvn
        self.use_dynamic_chunk_size = use_dynamic_chunk_size
        self.use_dynamic_left_chunk = use_dynamic_left_chunk
        self.static_chunk_size = static_chunk_size

    # ------------------------------------------------------------------------------------------------------------
    def _run_blocks(self, x, attn_mask, pos_embed, pad_mask, caches, keep_from, pos_shared=False, proj=None, ring=None, conv=None, streaming=False):
        """x (B,T',D) f32 -> after_norm(blocks(x)); returns (y, [trimmed per-layer caches] | None).
        ring = (kv f32 [L,B,H,ring_T,2dk], offsets int32 [B]), conv = f32 [L,B,K-1,D] | None: per-stream streaming state (StreamingBatch)."""
        n = len(self.encoders)
        if proj is None:
            proj = self._project_positions(pos_embed, x)
        bufs = [torch.empty_like(x), torch.empty_like(x)]
        new_caches = [] if caches is not None else None
        cur, ready = x, False
        prec = cfm.resolve_precision(self)
        # after_norm rides in the last block's final chain when that block runs on the row chains (one launch less)
        fuse_after = (cfm.rowchain_supported(self.encoder_dim, self.encoders[0].hidden_dim, prec) and
                      self.after_norm.weight.dtype == torch.float32 and self.after_norm.eps == 1e-5)
        y_after = torch.empty_like(x) if fuse_after else None
        # consecutive blocks chained: block i's last launch runs block i+1's macaron chain too (batch path on the row chains at the
        # config-2 width; encoder_layer.CHAIN_BLOCKS)
        import encoder_layer as _el
        e0 = self.encoders[0]
        chain = (_el.CHAIN_BLOCKS and not _el.MERGE_ATTENTION and fuse_after and caches is None and conv is None and self.encoder_dim == 256 and
                 e0.hidden_dim == 2048 and e0.kernel_size == 15 and not getattr(e0.conv_module, "causal", False) and
                 all(b.chain_ready(prec) for b in self.encoders) and
                 not (streaming and _el.SPLIT_FFN_FEW_ROWS and _el.split_rows(x.size(0) * x.size(1), self.encoder_dim, e0.hidden_dim)))   # few rows: the split feed-forward instead
        for i, block in enumerate(self.encoders):
            nxt = self.encoders[i + 1].norm_ff_macaron if i + 1 < n else None
            cache_i = None
            if caches is not None and caches.dim() == 5 and caches.size(0) > 0:
                cache_i = caches[i]                                   # batched streaming: (L,B,H,Tc,2dk)
            elif caches is not None and caches.dim() == 4 and caches.size(0) > 0:
                cache_i = caches[i:i + 1]
            pp = None if proj is None else (proj[:, i * self.encoder_dim:], proj.stride(0))
            out, nc = block.fused_forward(cur, attn_mask, pos_embed, pad_mask, cache_i, xn_ready=ready, next_norm=nxt,
                                          out=bufs[i & 1], want_cache=caches is not None, pos_proj=pp, pos_shared=pos_shared,
                                          ring=None if ring is None else (ring[0][i], ring[1]), conv_cache=None if conv is None else conv[i],
                                          chain_next=(self.encoders[i + 1], bufs[(i + 1) & 1]) if chain and i + 1 < n else None,
                                          macaron_done=chain and i > 0, split_ffn=streaming,
                                          after=(self.after_norm.weight.detach(), self.after_norm.bias.detach(), y_after)
                                          if fuse_after and i + 1 == n else None)
            if new_caches is not None:
                new_caches.append(nc[:, :, keep_from:, :])
            cur, ready = out, nxt is not None
        if fuse_after:
            return y_after, new_caches
        y, _ = cfm.layernorm(cur.view(-1, cur.size(-1)), self.after_norm.weight.detach(), self.after_norm.bias.detach(),
                             eps=self.after_norm.eps)
        return y.view_as(cur), new_caches

    def _cmvn_args(self, inputs):
        """(inputs, cmvn tuple for the front-end): a GlobalCMVN-like module with float32 `mean` / `istd` buffers on the input's device is
        folded into the first convolution (bit-identical); anything else is applied as the module it is."""
        m = self.global_cmvn
        if m is None:
            return inputs, None
        mean, istd = getattr(m, "mean", None), getattr(m, "istd", None)
        ok = (isinstance(mean, torch.Tensor) and isinstance(istd, torch.Tensor) and mean.dtype == torch.float32 and istd.dtype == torch.float32 and
              mean.dim() == 1 and mean.shape == istd.shape and mean.numel() == inputs.size(-1) and mean.device == inputs.device and
              inputs.dtype == torch.float32)
        if not ok:
            return m(inputs), None
        return inputs, (mean.contiguous(), istd.contiguous() if getattr(m, "norm_var", True) else None)

    def _project_positions(self, pos_embed, x):
        """linear_pos of EVERY block applied to pos_embed in one GEMM: [R, D] . [L*D, D]^T -> [R, L*D] (block i reads columns
        i*D .. (i+1)*D).  The per-block projections are 8 us launches of a 32-row GEMM in the batch path otherwise."""
        if not isinstance(self.position_encoding, RelativePositionalEncoding) or pos_embed is None:
            return None
        from cfm import packing
        prec = cfm.resolve_precision(self)
        ws = [blk.self_attn.linear_pos.weight for blk in self.encoders]
        if not hasattr(self, "_pos_pack"):
            self._pos_pack = packing.PackCache()
        pk = self._pos_pack.get(ws, prec, lambda: packing.Packed(w=packing.matrix(torch.cat([w.detach() for w in ws], 0), prec)))
        pe = pos_embed.reshape(-1, self.encoder_dim)
        pe = (pe if pe.dtype == torch.float32 else pe.float()).contiguous()
        return cfm.gemm(pe, pk.w[0], w_lo=pk.w[1], out_dtype=prec.act_dtype)

    def set_precision(self, name):
        """Pin this encoder (every drop-in module under it) to a precision mode, independent of the process default
        (cfm.set_precision): 'bf16' | 'fp16' | 'fp32' | None (follow the default again)."""
        prec = None if name is None else (name if isinstance(name, cfm.Precision) else cfm.Precision(name))
        for m in self.modules():
            m.precision = prec
        return self

    def forward(self, inputs, input_lengths, decoding_chunk_size=0, num_decoding_chunk_size=-1):
        if self.training:
            # train mode (encoder.py:54-75 under module.train()): an accumulation window of one micro-batch
            return self.forward_window([(inputs, input_lengths)], decoding_chunk_size, num_decoding_chunk_size)[0]
        inputs, cmvn = self._cmvn_args(inputs)
        cfm.require_hip(inputs, input_lengths)
        frames = inputs.size(1)
        x = self.embed.embed_frames(inputs, cmvn)
        x, pos_embed = self.position_encoding(x, 0)
        # (~make_pad_mask(len, T))[:, None, :][:, :, 2::2][:, :, 2::2]  ==  (6 + 4 j < len), built in one launch
        pad_mask = cfm.valid_mask(input_lengths, x.size(1), first=6, stride=4).unsqueeze(1)
        # (Tried: positional projection + mask on a side stream beside the front-end.  Under graph replay the fork/join costs more
        # than the ~11 us it hides and the concurrent kernels slow the first convolution: 1.697 vs 1.668 ms per step.)
        attn_mask = make_attn_mask(x, pad_mask, self.use_dynamic_chunk_size, self.use_dynamic_left_chunk,
                                   decoding_chunk_size, self.static_chunk_size, num_decoding_chunk_size)
        # split_small_batches (attribute, default False): a whole-utterance forward of <= 1536 rows (up to 6 x 1000 frames) takes the split
        # feed-forward path of the streaming steps (csrc/ffnsplit.hip) -- lower latency for a small batch; off by default because the result then
        # depends on which side of 1536 rows the batch falls (to rounding, 1e-3 of the output), and a batch shard no longer reproduces the batch bit for bit
        y, _ = self._run_blocks(x, attn_mask, pos_embed, pad_mask, None, 0, streaming=bool(getattr(self, "split_small_batches", False)))
        return y.to(inputs.dtype), pad_mask

    def forward_window(self, batches, decoding_chunk_size=0, num_decoding_chunk_size=-1, return_rows=False):
        """Train mode: the micro-batches of ONE accumulation window (train.sh:36 accum_grad; the weights do not change between them) in a single
        pass.  batches: [(inputs (B_g,T_g,F), lengths (B_g,)), ...]; returns [(outputs (B_g,T'_g,D), pad mask (B_g,1,T'_g)), ...], item g equal to
        what `forward(*batches[g])` returns when the micro-batches are run one after the other as the reference does (encoder.py:54-75 called
        accum_grad times; BatchNorm batch statistics per micro-batch, running statistics updated in micro-batch order; dropout masks
        differ, as between any two calls).  The rows of all micro-batches go through the block stack TOGETHER (cfm/autograd.py EncoderStackFn:
        the dense products, LayerNorms and the weight gradients see one [sum B_g*T'_g, D] row matrix; attention, depthwise convolution and
        BatchNorm run per micro-batch), so a window costs about half the launches of two passes and its weight gradient is one product.
        return_rows: also return the outputs as ONE row matrix f32 [sum B_g*T'_g, D] (what the per-micro-batch outputs are views of) -- a head that is
        row-local up to its loss (CTCDecoder.forward_window) then projects all micro-batches in one launch: returns (rows, outs)."""
        if not self.training:
            raise RuntimeError("ConformerEncoder.forward_window is the train-mode pass over an accumulation window; in eval mode call forward per batch")
        from cfm import autograd as ag
        from encoder_layer import _mask_args
        layers = list(self.encoders)
        flat = all(l.__dict__.get("_flat_leaf") is not None for l in layers)
        if not ag.stack_supported(layers, flat):
            # blocks the stack path does not take (other kernel sizes, parameter-free BatchNorm ...): micro-batch after micro-batch, block by block
            outs = []
            for inputs, lengths in batches:
                inputs, cmvn = self._cmvn_args(inputs)
                cfm.require_hip(inputs, lengths)
                x, pos_embed = self.position_encoding(self.embed.embed_frames(inputs, cmvn), 0)
                pad_mask = cfm.valid_mask(lengths, x.size(1), first=6, stride=4).unsqueeze(1)
                attn_mask = make_attn_mask(x, pad_mask, self.use_dynamic_chunk_size, self.use_dynamic_left_chunk, decoding_chunk_size,
                                           self.static_chunk_size, num_decoding_chunk_size)
                for block in layers:
                    x, attn_mask, _, _ = block(x, attn_mask, pos_embed, pad_mask)
                y = ag.LayerNormFn.apply(x, self.after_norm.weight, self.after_norm.bias, self.after_norm.eps)
                outs.append((y.to(inputs.dtype), pad_mask))
            if return_rows:
                return torch.cat([y.float().reshape(-1, y.size(-1)) for y, _ in outs], 0), outs
            return outs
        prec = cfm.resolve_precision(self)
        rows, groups, keeps, shapes, pads = [], [], [], [], []
        for gi, (inputs, lengths) in enumerate(batches):
            inputs, cmvn = self._cmvn_args(inputs)
            cfm.require_hip(inputs, lengths)
            x, pos_embed = self.position_encoding(self.embed.embed_frames(inputs, cmvn), 0)
            B, T, D = x.shape
            pad_mask = cfm.valid_mask(lengths, T, first=6, stride=4).unsqueeze(1)
            attn_mask = make_attn_mask(x, pad_mask, self.use_dynamic_chunk_size, self.use_dynamic_left_chunk, decoding_chunk_size,
                                       self.static_chunk_size, num_decoding_chunk_size)
            m8, m_str = _mask_args(attn_mask, B, T, T)
            cfm.require_hip(m8)
            rows.append(x.reshape(B * T, D))
            groups.append((B, T, m8, m_str))
            keeps.append(cfm.as_u8_mask(pad_mask).reshape(-1))
            shapes.append((B, T, inputs.dtype))
            pads.append(pad_mask)
        x_rows = rows[0] if len(rows) == 1 else torch.cat(rows, 0)
        x_rows = (x_rows if x_rows.dtype == torch.float32 else x_rows.float()).contiguous()
        keep = keeps[0] if len(keeps) == 1 else torch.cat(keeps, 0)
        params = tuple(l.__dict__["_flat_leaf"] for l in layers) if flat else tuple(p for l in layers for p in l.parameters())
        y = ag.EncoderStackFn.apply(x_rows, self, layers, prec, groups, keep, flat, *params)
        y = ag.LayerNormFn.apply(y, self.after_norm.weight, self.after_norm.bias, self.after_norm.eps)
        outs, r0 = [], 0
        for (B, T, dt), pad_mask in zip(shapes, pads):
            outs.append((y[r0:r0 + B * T].view(B, T, -1).to(dt), pad_mask))
            r0 += B * T
        return (y, outs) if return_rows else outs

    def forward_chunk(self, inputs, offset, required_cache_size, attn_cache, cnn_cache, inputs_attn_mask=_NO_MASK, pos_rows=None, abs_rows=None):
        """One streaming step.  Batch 1 as in the reference: attn_cache (L,H,Tc,2dk) or empty; returns (chunk output, new attn
        cache (L,H,Tc',2dk), cnn cache (L,0,0,0) -- the reference keeps no conv context).
        Batch B > 1 (beyond the reference, whose forward_chunk only works at batch 1: SURVEY 8 row S): B streams in lockstep at
        the same `offset`, attn_cache (L,B,H,Tc,2dk) or empty, returned cache (L,B,H,Tc',2dk); item b of the result equals the
        batch-1 call on inputs[b:b+1] with attn_cache[:, b].
        pos_rows (cached + chunk, 1, D): the positional rows for this step in a caller-owned buffer instead of a slice taken at `offset`
        (StreamingSession replays one captured graph per step and refreshes that buffer between replays)."""
        if self.training:
            raise NotImplementedError("ConformerEncoder.forward_chunk: streaming is inference-only; call .eval()")
        inputs, cmvn = self._cmvn_args(inputs)
        cfm.require_hip(inputs)
        dev = inputs.device
        attn_cache = attn_cache.to(dev)
        x = self.embed.embed_frames(inputs, cmvn)
        if abs_rows is not None:                    # absolute encoding, rows in a caller-owned buffer (StreamingSession)
            x, _ = self.position_encoding(x, offset, rows=abs_rows)
        else:
            x, _ = self.position_encoding(x, offset)
        batched = inputs.size(0) > 1
        have = attn_cache.dim() == (5 if batched else 4) and attn_cache.size(0) > 0
        if batched and attn_cache.dim() == 4 and attn_cache.size(0) > 0:
            raise RuntimeError("forward_chunk with %d streams needs attn_cache of shape (L,B,H,Tc,2dk)" % inputs.size(0))
        cached = attn_cache.size(-2) if have else 0
        span = cached + x.size(1)
        pos_embed = pos_rows if pos_rows is not None else self.embed.position_encoding(offset=offset - cached, size=span)
        if pos_embed.size(0) != span:
            raise RuntimeError("forward_chunk: %d positional rows for %d keys" % (pos_embed.size(0), span))
        if required_cache_size < 0:
            keep_from = 0
        elif required_cache_size == 0:
            keep_from = span
        else:
            keep_from = max(span - required_cache_size, 0)
        caches = attn_cache if have else torch.zeros((0, 0, 0, 0), device=dev)
        y, new = self._run_blocks(x, inputs_attn_mask, pos_embed, None, caches, keep_from, pos_shared=batched, streaming=True)
        r_attn = torch.stack(new, dim=0) if batched else torch.cat(new, dim=0)
        r_cnn = torch.zeros((len(self.encoders), 0, 0, 0), dtype=x.dtype, device=dev)
        return y.to(inputs.dtype), r_attn, r_cnn

    def forward_chunk_by_chunk(self, inputs, decoding_chunk_size, num_decoding_left_chunks=-1):
        """Simulated streaming over a whole utterance: windows of (c-1)*4+7 frames every 4*c frames."""
        hop = 4 * decoding_chunk_size
        window = (decoding_chunk_size - 1) * 4 + 7
        total = inputs.size(1)
        need = decoding_chunk_size * num_decoding_left_chunks
        attn_cache = torch.zeros((0, 0, 0, 0), device=inputs.device)
        cnn_cache = torch.zeros((0, 0, 0, 0), device=inputs.device)
        pieces, offset = [], 0
        for start in range(0, total - 7 + 1, hop):
            y, attn_cache, cnn_cache = self.forward_chunk(inputs[:, start:min(start + window, total), :], offset, need,
                                                          attn_cache, cnn_cache)
            pieces.append(y)
            offset += y.size(1)
        out = torch.cat(pieces, 1)
        return out, torch.ones((out.size(0), 1, out.size(1)))


def _weights_signature(owner):
    """Changes whenever a parameter / buffer of owner.enc is updated in place or repacked: a captured graph holds raw pointers to the PACKED
    weights of that moment and must be re-captured then.  Runs before every replayed step, so it is kept to one attribute read per tensor
    (~50 us for the 12-layer encoder; data pointers are checked when the tensor list is (re)built: .to() / load_state_dict keep storage)."""
    from cfm import packing
    ts = owner.__dict__.get("_sig_tensors")
    if ts is None:
        ts = list(owner.enc.parameters()) + list(owner.enc.buffers())
        owner.__dict__["_sig_tensors"] = ts
        owner.__dict__["_sig_ptrs"] = sum(t.data_ptr() for t in ts)
    v = 0
    for t in ts:
        v += t._version
    return packing._EPOCH[0], v, owner.__dict__["_sig_ptrs"], str(cfm.resolve_precision(owner.enc))


class StreamingSession:
    """B streams advanced in lockstep, chunk by chunk, with ONE captured HIP graph per steady-state step.

    The eager streaming step is launch-bound (~70 launches for a 16-frame chunk).  Once the left-context cache is full every step has
    the same shapes, so the step is captured once -- static buffers for the feature window, the positional rows and the KV cache,
    the new cache copied back into the static one inside the graph -- and replayed; only the window and the positional rows are
    refreshed between replays.  Results are those of ConformerEncoder.forward_chunk (bit for bit: the same kernels on the same data).
    `step` returns a buffer that the next step overwrites."""

    def __init__(self, encoder, decoding_chunk_size, num_decoding_left_chunks):
        if num_decoding_left_chunks <= 0:
            raise ValueError("StreamingSession needs a bounded left context (num_decoding_left_chunks > 0): the step shapes must settle")
        self.enc = encoder
        self.chunk = decoding_chunk_size
        self.need = decoding_chunk_size * num_decoding_left_chunks
        self.window = (decoding_chunk_size - 1) * 4 + 7
        self.hop = 4 * decoding_chunk_size
        self.offset = 0
        self.cache = None
        self.graph = None
        self.relative = isinstance(encoder.position_encoding, RelativePositionalEncoding)

    def _weights_signature(self):
        return _weights_signature(self)

    def _abs_rows(self, batch, dev):
        pe = self.enc.position_encoding._table_like(torch.empty(0, device=dev, dtype=torch.float32))
        return pe[self.offset: self.offset + batch]

    def step(self, frames):
        """frames (B, window, F) on the GPU -> (B, chunk, D)."""
        if frames.size(1) != self.window:
            raise ValueError("StreamingSession.step wants windows of %d frames" % self.window)
        dev = frames.device
        empty = torch.zeros((0, 0, 0, 0), device=dev)
        if self.graph is not None and self._weights_signature() != self._sig:
            self.cache, self.graph = self.kv.clone(), None            # weights changed under the captured graph: back to eager, re-capture
        if self.graph is None:
            cache = self.cache if self.cache is not None else empty
            y, self.cache, _ = self.enc.forward_chunk(frames, self.offset, self.need, cache, empty)
            self.offset += y.size(1)
            if self.cache.size(-2) == self.need:
                self._capture(frames)
            return y
        self.x.copy_(frames)
        self.pos.copy_(self.enc.embed.position_encoding(offset=self.offset - self.need, size=self.need + self.chunk))
        if not self.relative:
            self.abs.copy_(self._abs_rows(frames.size(0), dev))
        self.graph.replay()
        self.offset += self.chunk
        return self.y

    def _capture(self, frames):
        enc, dev = self.enc, frames.device
        empty = torch.zeros((0, 0, 0, 0), device=dev)
        self.x = frames.clone()
        self.kv = self.cache.clone()
        self.pos = enc.embed.position_encoding(offset=self.offset - self.need, size=self.need + self.chunk).clone()
        # absolute encoding (use_relative=False): the table rows added to the chunk live in a static buffer too -- sliced at `offset`
        # inside the captured region they would be frozen at the capture step's offset
        self.abs = None if self.relative else self._abs_rows(frames.size(0), dev).clone()
        self._sig = self._weights_signature()
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(stream):
            enc.forward_chunk(self.x, self.offset, self.need, self.kv, empty, pos_rows=self.pos, abs_rows=self.abs)      # sizes the scratch arena
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                y, new_kv, _ = enc.forward_chunk(self.x, self.offset, self.need, self.kv, empty, pos_rows=self.pos, abs_rows=self.abs)
                self.kv.copy_(new_kv)
            self.y = y
        torch.cuda.current_stream(dev).wait_stream(stream)
        self.kv.copy_(self.cache)                                  # the capture pass and its warm-up advanced the static cache: restore
        self.graph = graph



class StreamingBatch:
    """B independent streams, each at ITS OWN position, advanced together one chunk per step (BASELINE config 5:
    64 streams, chunk 16, cached left context) -- what the reference cannot do: its forward_chunk serves one stream at a time
    (encoder.py:78-123: the positional view fails at batch > 1) and rebuilds the attention cache with cat + slice every step (:117).

      * per-stream state lives on the device: `offsets` int32 [B] (encoder frames consumed) and, per layer, a K/V RING buffer
        f32 [B,H,need+chunk,2dk] -- frame f in slot f mod ring_T; a step writes its `chunk` new frames and moves nothing;
      * a step is: front-end on the B windows, cfm_stream_prep (slot mask + positional rows per stream from `offsets`), one projection
        of those rows for all layers, the blocks with attention over the ring under the slot mask, cfm_stream_advance;
      * no host-side scalar enters the step, so it is captured ONCE as a HIP graph and replayed (graph=True);
      * item b of the output equals the reference's batch-1 forward_chunk on stream b with its own cache and offset (tests: against
        the CPU oracle looped per stream) to the precision mode's tolerance -- not bit for bit, the softmax sums run in slot order;
      * causal_conv=True is the OPT-IN extension (not in the reference, which has no causal convolution and ignores cnn_cache):
        every block's depthwise conv becomes causal with a (K-1)-frame left context per stream.  Off = the reference's arithmetic.

    step(frames (B, (chunk-1)*4+7, F)) -> (B, chunk, D) in a buffer the next step overwrites.  reset(streams) starts new utterances."""

    def __init__(self, encoder, streams, decoding_chunk_size, num_decoding_left_chunks, causal_conv=False, graph=True):
        if num_decoding_left_chunks < 0:
            raise ValueError("StreamingBatch needs a bounded left context (num_decoding_left_chunks >= 0)")
        if encoder.training:
            raise NotImplementedError("StreamingBatch: streaming is inference-only; call encoder.eval()")
        self.enc, self.B = encoder, int(streams)
        self.chunk, self.need = int(decoding_chunk_size), int(decoding_chunk_size) * int(num_decoding_left_chunks)
        self.ring_T = self.need + self.chunk
        self.window = (self.chunk - 1) * 4 + 7
        self.relative = isinstance(encoder.position_encoding, RelativePositionalEncoding)
        blk = encoder.encoders[0]
        L, H, D, K = len(encoder.encoders), blk.num_heads, encoder.encoder_dim, blk.kernel_size
        dev = next(encoder.parameters()).device
        cfm.require_hip(next(encoder.parameters()))
        self.dev = dev
        self.offsets = torch.zeros((self.B,), dtype=torch.int32, device=dev)
        self.kv = torch.zeros((L, self.B, H, self.ring_T, 2 * (D // H)), dtype=torch.float32, device=dev)      # zero: masked slots must stay finite
        self.slot_mask = torch.zeros((self.B, 1, self.ring_T), dtype=torch.uint8, device=dev)
        self.pos_rows = torch.zeros((self.B * self.ring_T, 1, D), dtype=torch.float32, device=dev)
        self.abs_rows = None if self.relative else torch.zeros((self.B, 1, D), dtype=torch.float32, device=dev)
        self.causal = bool(causal_conv)
        self.conv = torch.zeros((L, self.B, K - 1, D), dtype=torch.float32, device=dev) if self.causal else None
        self.pe = encoder.position_encoding.pe.reshape(-1, D).to(device=dev, dtype=torch.float32).contiguous()
        # host mirror of an UPPER BOUND of `offsets` (every step advances every stream by `chunk`; reset() zeroes): lets step() see, without a
        # device sync, that a stream is about to read past the sinusoid table -- the reference's pe[offset:offset+size] comes back short there
        # and fails (attention.py:25-29; max_len = 5000 encoder frames = 200 s of audio); a long-lived stream here gets the table EXTENDED
        # with the same formula instead of wrong rows (cfm_stream_prep used to clamp to the last row, silently)
        self._host_off = [0] * self.B
        self.x = None
        self.use_graph, self.graph, self.y, self._sig = bool(graph), None, None, None
        self.steps = 0

    def reset(self, streams=None):
        """start new utterances on the given streams (all by default): position 0, empty left context."""
        if streams is None:
            self.offsets.zero_()
            self._host_off = [0] * self.B
            if self.conv is not None:
                self.conv.zero_()
        else:
            streams = list(streams)
            idx = torch.as_tensor(streams, dtype=torch.long, device=self.dev)
            for b in streams:
                self._host_off[b] = 0
            self.offsets[idx] = 0
            if self.conv is not None:
                self.conv[:, idx] = 0

    def _signature(self):
        return _weights_signature(self)

    def _grow_table(self, rows_needed):
        """Append sinusoid rows [len, new_len) -- the reference's formula (attention.py:12-16 / :109-115 incl. the absolute table's fp16
        rounding), existing rows untouched -- and drop the captured graph (it holds the old table's address)."""
        from attention import _sinusoid_table
        have = self.pe.size(0)
        new_len = max(2 * have, rows_needed)
        full = _sinusoid_table(new_len, self.pe.size(1), type(self.enc.position_encoding)._table_dtype).reshape(new_len, -1)
        self.pe = torch.cat([self.pe, full[have:].to(device=self.dev, dtype=torch.float32)]).contiguous()
        self.graph = None

    def _step_impl(self):
        enc = self.enc
        for blk in enc.encoders:
            blk.conv_module.causal = self.causal
        try:
            inputs, cmvn = enc._cmvn_args(self.x)
            x = enc.embed.embed_frames(inputs, cmvn)
            cfm.stream_prep(self.offsets, self.chunk, self.need, self.ring_T, self.pe, self.slot_mask, self.pos_rows, self.abs_rows)
            if self.relative:
                x, _ = enc.position_encoding(x, 0)
            else:
                x, _ = enc.position_encoding(x, 0, rows=self.abs_rows)
            proj = enc._project_positions(self.pos_rows, x)
            y, _ = enc._run_blocks(x, self.slot_mask.view(torch.bool), self.pos_rows, None, None, 0, proj=proj, ring=(self.kv, self.offsets), conv=self.conv, streaming=True)
            cfm.stream_advance(self.offsets, self.chunk)
        finally:
            for blk in enc.encoders:
                blk.conv_module.causal = False
        return y

    def step(self, frames):
        if tuple(frames.shape[:2]) != (self.B, self.window):
            raise ValueError("StreamingBatch.step wants (%d, %d, F) frames, got %s" % (self.B, self.window, tuple(frames.shape)))
        cfm.require_hip(frames)
        if self.x is None:
            self.x = torch.empty_like(frames, dtype=torch.float32).contiguous()
        self.x.copy_(frames)
        self.steps += 1
        if max(self._host_off) + self.chunk > self.pe.size(0):
            self._grow_table(max(self._host_off) + self.chunk)
        self._host_off = [o + self.chunk for o in self._host_off]
        with torch.no_grad():
            if not self.use_graph:
                return self._step_impl()
            if self.graph is not None and self._signature() != self._sig:
                self.graph = None                                         # weights changed under the captured graph
            if self.graph is None:
                y = self._step_impl()                                     # this step runs eagerly (it also sizes the arena and packs the weights)
                stream = torch.cuda.Stream(device=self.dev)
                stream.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(stream):
                    saved = self.offsets.clone(), None if self.conv is None else self.conv.clone()
                    self._step_impl()                                     # warm-up ON the capture stream (its own scratch arena) ...
                    stream.synchronize()
                    self.offsets.copy_(saved[0])                          # ... with the state it advanced put back (ring slots rewritten: same data)
                    if self.conv is not None:
                        self.conv.copy_(saved[1])
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=stream):
                        self.y = self._step_impl()
                torch.cuda.current_stream(self.dev).wait_stream(stream)
                self.graph, self._sig = graph, self._signature()
                return y
            self.graph.replay()
            return self.y
