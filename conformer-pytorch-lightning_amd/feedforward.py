"""Position-wise feed-forward block of the conformer, MI355X-native.

Drop-in for the reference's ``src/feedforward.py`` (class name, constructor arguments, parameter names
``w_1.{weight,bias}`` / ``w_2.{weight,bias}`` and ``forward(inputs)`` are the reference's, feedforward.py:4-21).
The arithmetic is ONE launch of the fused feed-forward kernel of libconformer_gfx950 (csrc/ffn.hip: W1, bias, SiLU/ReLU, W2,
bias with the hidden activation kept in registers) when the shape has a fused instance, otherwise two MFMA GEMMs whose
epilogues carry the bias / activation.  In train mode (module.train()) the forward is a torch.autograd.Function over the train kernels
(cfm/autograd.py): pre-activation kept for the backward, gradients returned for every parameter.
"""
import torch
import torch.nn as nn

import cfm
from cfm import packing


def _inference_only(module, what):
    """Entry points that only exist for inference (KV-cache streaming): refuse train mode loudly."""
    if module.training:
        raise NotImplementedError("%s: this entry point is inference-only (streaming / KV cache); call .eval()" % what)


class PositionwiseFeedForwardModule(nn.Module):

    def __init__(self, input_dim, dropout, hidden_dim, activation='swish'):
        super().__init__()
        self.w_1 = nn.Linear(input_dim, hidden_dim)
        self.activation = nn.SiLU() if activation == 'swish' else nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        self.w_2 = nn.Linear(hidden_dim, input_dim)
        self._pack = packing.PackCache()

    def forward(self, inputs):
        cfm.require_hip(inputs)
        prec = cfm.resolve_precision(self)
        if cfm.check_mode(self, "PositionwiseFeedForwardModule"):
            from cfm import autograd as ag
            act = cfm.ACT_SILU if isinstance(self.activation, nn.SiLU) else cfm.ACT_RELU
            return ag.FeedForwardFn.apply(inputs, self, prec, act, *self.parameters())
        pk = packing.pack_ffn(self, prec)
        d_in = inputs.shape[-1]
        x = inputs.reshape(-1, d_in)
        if x.dtype != torch.float32 and x.dtype != prec.w_dtype:
            x = x.float()
        if prec.split and x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        act = cfm.ACT_SILU if isinstance(self.activation, nn.SiLU) else cfm.ACT_RELU
        if pk.w1f is not None and x.dtype == torch.float32:
            out, _ = cfm.ffn_fused(x, pk.w1f, pk.w2f, pk.b1, pk.b2, self.w_1.weight.shape[0], act=act)
            return out.view(*inputs.shape[:-1], out.shape[-1]).to(inputs.dtype)
        hid = cfm.gemm(x, pk.w1, bias=pk.b1, w_lo=pk.w1_lo, act=act, out_dtype=prec.act_dtype)
        out = cfm.gemm(hid, pk.w2, bias=pk.b2, w_lo=pk.w2_lo, out_dtype=torch.float32)
        return out.view(*inputs.shape[:-1], out.shape[-1]).to(inputs.dtype)
