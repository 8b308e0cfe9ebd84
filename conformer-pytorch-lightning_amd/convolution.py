"""Convolution module and Conv2d subsampling front-end of the conformer, MI355X-native.

Drop-in for the reference's ``src/convolution.py`` (class names, constructor arguments, parameter names and ``forward``
signatures of convolution.py:5-49 and :52-79).  The nn.Conv1d / nn.Conv2d / nn.BatchNorm1d children exist only to own the
parameters under the reference's names; their forward is never called.  What runs instead:

ConvolutionModule   (activations stay time-major [B,T,D] -- no transposes)
    pointwise_conv1 + GLU      one MFMA GEMM (N = 2D, weight rows interleaved so value/gate share a lane), epilogue
                               a * sigmoid(g); padded INPUT frames are zeroed in the epilogue (acc = 0, bias kept:
                               exactly the reference's mask-before-pw1, quirk Q5)
    depthwise + BN + SiLU      one HBM-bound kernel (csrc/convmod.hip), BatchNorm folded to scale/shift (eval)
    pointwise_conv2 + mask     one MFMA GEMM, padded OUTPUT frames zeroed in the epilogue
    ``cache`` is ignored and the returned cache is an empty (0,0,0) tensor, as in the reference (quirk Q4).

ConvolutionSubSampling
    Conv2d(1,D,3,2)+ReLU       csrc/convmod.hip, written channels-last [B,T1,F1,D]
    Conv2d(D,D,3,2)+ReLU       implicit GEMM (M = B*T'*F', N = D, K = 9D) on the MFMA GEMM kernel
    Linear(D*F' -> D)          MFMA GEMM; the weight's K axis is permuted once so that the channels-last activation
                               is consumed directly (reference feature order c*F'+f, convolution.py:74)
    mask                       inputs_pad_mask[:, :, 2::2][:, :, 2::2], bit-exact (a strided view, no kernel)
"""
import torch
import torch.nn as nn

import cfm
from cfm import packing

_NO_CACHE = torch.zeros((0, 0, 0, 0))


def _rows_f32(t):
    t = t.reshape(-1, t.size(-1))
    return (t if t.dtype == torch.float32 else t.float()).contiguous()


class ConvolutionModule(nn.Module):

    def __init__(self, input_dim, kernel_size, bias=True):
        super().__init__()
        bias = bool(bias)          # the reference's encoder layer passes hidden_dim here (quirk Q1): any truthy value
        self.pointwise_conv1 = nn.Conv1d(input_dim, 2 * input_dim, kernel_size=1, stride=1, padding=0, bias=bias)
        self.glu = nn.GLU(dim=1)
        self.depthwise_conv = nn.Conv1d(input_dim, input_dim, kernel_size=kernel_size, stride=1,
                                        padding=(kernel_size - 1) // 2, groups=input_dim, bias=bias)
        self.norm = nn.BatchNorm1d(input_dim)
        self.activation = nn.SiLU()
        self.pointwise_conv2 = nn.Conv1d(input_dim, input_dim, kernel_size=1, stride=1, padding=0)
        self._pack = packing.PackCache()
        # OPT-IN extension, not in the reference (which has no causal mode and ignores its cache, convolution.py:34-39): the depthwise taps
        # reach back kernel_size-1 frames instead of (kernel_size-1)/2 each way, and `cache` (B, kernel_size-1, D) carries the left context
        # between chunks -- then chunk-by-chunk output equals the whole-utterance output.  False = the reference's convolution, bit for bit.
        self.causal = False

    def forward(self, inputs, inputs_pad_mask, cache=_NO_CACHE):
        cfm.require_hip(inputs)
        prec = cfm.resolve_precision(self)
        B, T, D = inputs.shape
        keep = None
        if inputs_pad_mask is not None and inputs_pad_mask.dim() >= 3 and inputs_pad_mask.size(2) > 0:
            keep = cfm.as_u8_mask(inputs_pad_mask).reshape(-1)
            if keep.numel() != B * T:
                raise RuntimeError("pad mask %s does not match inputs %s" % (tuple(inputs_pad_mask.shape), tuple(inputs.shape)))
        if cfm.check_mode(self, "ConvolutionModule"):
            # BatchNorm batch statistics over all B*T positions (padded ones included, quirk Q6), running statistics updated
            from cfm import autograd as ag
            out = ag.ConvModuleFn.apply(inputs, self, prec, keep, *self.parameters())
            return out, torch.zeros((0, 0, 0), dtype=inputs.dtype, device=inputs.device)
        pk = packing.pack_conv_module(self, prec)
        x = _rows_f32(inputs)
        glu = cfm.gemm(x, pk.pw1_w, bias=pk.pw1_b, w_lo=pk.pw1_w_lo, act=cfm.ACT_GLU, row_mask=keep, mask_mode=1,
                       out_dtype=prec.act_dtype)
        if self.causal:
            K = self.depthwise_conv.kernel_size[0]
            have = cache is not None and cache.dim() == 3 and cache.size(0) > 0
            ctx = cache.to(device=inputs.device, dtype=torch.float32).contiguous() if have else None
            dw = cfm.dwconv_causal_bn_silu(glu.view(B, T, D), pk.dw_w, pk.dw_b, pk.bn_scale, pk.bn_shift, cache=ctx)
            new_ctx = ctx.clone() if have else torch.zeros((B, K - 1, D), dtype=torch.float32, device=inputs.device)
            cfm.conv_cache_update(glu.view(B, T, D), new_ctx, K)
            out = cfm.gemm(dw.view(B * T, D), pk.pw2_w, bias=pk.pw2_b, w_lo=pk.pw2_w_lo, row_mask=keep, mask_mode=0, out_dtype=torch.float32)
            return out.view(B, T, D).to(inputs.dtype), new_ctx
        dw = cfm.dwconv_bn_silu(glu.view(B, T, D), pk.dw_w, pk.dw_b, pk.bn_scale, pk.bn_shift)
        out = cfm.gemm(dw.view(B * T, D), pk.pw2_w, bias=pk.pw2_b, w_lo=pk.pw2_w_lo, row_mask=keep, mask_mode=0,
                       out_dtype=torch.float32)
        new_cache = torch.zeros((0, 0, 0), dtype=inputs.dtype, device=inputs.device)
        return out.view(B, T, D).to(inputs.dtype), new_cache


FUSE_CONVS = True     # tests switch it off to compare with the two-kernel path


class ConvolutionSubSampling(nn.Module):

    def __init__(self, input_dim, output_dim, pos_enc):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(1, output_dim, 3, 2),
            nn.ReLU(),
            nn.Conv2d(output_dim, output_dim, 3, 2),
            nn.ReLU(),
        )
        self.out = nn.Sequential(nn.Linear(output_dim * (((input_dim - 1) // 2 - 1) // 2), output_dim))
        self.pos_enc = pos_enc
        self._pack = packing.PackCache()

    def embed_frames(self, inputs, cmvn=None):
        """(B,T,F) fbank -> (B,T',D) f32: the two stride-2 convolutions and the output projection.  cmvn = (mean, istd | None):
        global CMVN (cmvn.py:22-33) folded into the first convolution's tap loads instead of a pass of its own."""
        cfm.require_hip(inputs)
        prec = cfm.resolve_precision(self)
        if cfm.check_mode(self, "ConvolutionSubSampling"):
            from cfm import autograd as ag
            return ag.SubsamplingFn.apply(inputs, self, prec, cmvn, *ag.subsampling_params(self))
        pk = packing.pack_subsampling(self, prec)
        x = (inputs if inputs.dtype == torch.float32 else inputs.float()).contiguous()
        B, T, F = x.shape
        C = pk.C
        T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        if T2 < 1 or F2 != pk.Fp:
            raise RuntimeError("input of shape %s is too short / has the wrong feature size for this front-end" % (tuple(inputs.shape),))
        if FUSE_CONVS and pk.w2_lo is None and cfm.conv12_supported(C, prec.act_dtype):
            # 16-bit modes: both convolutions in one kernel, the first one recomputed in the second one's operand producer (bit-identical)
            h2 = cfm.conv12_relu(x, pk.w1, pk.b1, pk.w2, pk.b2, cmvn=cmvn)                         # [B*T2*F2, C]
        else:
            # the 9 taps on the matrix pipe in the 16-bit modes (operands rounded to the mode's type, as every other contraction of the mode)
            h1 = cfm.conv1_relu(x, pk.w1, pk.b1, prec.act_dtype, cmvn=cmvn,
                                mma=cfm.conv1_relu_mma_supported(C, prec.act_dtype))             # [B,T1,F1,C]
            h2 = cfm.gemm(h1, pk.w2, bias=pk.b2, w_lo=pk.w2_lo, act=cfm.ACT_RELU, conv=(C, T1, F1, T2, F2, B * T2 * F2),
                          out_dtype=prec.act_dtype)                                              # [B*T2*F2, C]
        y = cfm.gemm(h2.view(B * T2, F2 * C), pk.wl, bias=pk.bl, w_lo=pk.wl_lo, out_dtype=torch.float32)
        return y.view(B, T2, -1)

    def forward(self, inputs, inputs_pad_mask, offset=0):
        y = self.embed_frames(inputs).to(inputs.dtype)
        y, pos_embed = self.pos_enc(y, offset)
        return y, pos_embed, inputs_pad_mask[:, :, 2::2][:, :, 2::2]

    def position_encoding(self, offset, size):
        return self.pos_enc.position_encoding(offset, size)
