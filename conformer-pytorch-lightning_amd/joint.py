"""Transducer joint network -- drop-in for the class of the same name in the reference's src/joint.py:4-38 (forward only).

    e = enc_ffn(enc_out)   [B*T, J]      cfm_gemm, f32
    p = pred_ffn(pred_out) [B*U, J]      cfm_gemm, f32
    a = tanh(e[b,t] + p[b,u])            cfm_joint_act: ONE 16-bit [B*T*U, J] operand (f32 in the accurate mode)
    out = ffn_out(a)       [B,T,U,V]     cfm_gemm, f32 logits with the reference's own layout (row stride V; V = 5002 is written with
                                         column-pair stores, an odd V through a V+1 wide buffer and a view)

Same constructor arguments, parameter names (`enc_ffn`, `pred_ffn`, `ffn_out`) and the attribute `activatoin` (sic) as the reference;
`forward(enc_out, pred_out, pre_project=True)` accepts the 3-D tensors of model.py:102 / :250 and the 4-D singleton-axis form of
joint.py:29-33.  `out_dtype` (default torch.float32, what the reference returns) may be set to a 16-bit type to halve the 3.3 GB the
logits take at BASELINE config 4.  The RNN-T loss itself (torchaudio.functional.rnnt_loss, model.py:107) is outside this repository:
torchaudio is absent here, so nothing pins it.  No backward: calling it with gradients enabled on parameters that
require them raises.
"""
import torch
import torch.nn as nn

import cfm
from cfm import packing


class TransducerJoint(nn.Module):

    def __init__(self, vocab_size, enc_output_size, pred_output_size, join_dim):
        super().__init__()
        self.activatoin = nn.Tanh()
        self.enc_ffn = nn.Linear(enc_output_size, join_dim)
        self.pred_ffn = nn.Linear(pred_output_size, join_dim)
        self.ffn_out = nn.Linear(join_dim, vocab_size)
        self.out_dtype = torch.float32
        self._pack = packing.PackCache()

    def _weights(self, prec):
        def build():
            def lin(m, pad_rows=0):
                w = m.weight.detach().float()
                b = m.bias.detach().float()
                if pad_rows:
                    w = torch.cat([w, w.new_zeros((pad_rows, w.shape[1]))])
                    b = torch.cat([b, b.new_zeros((pad_rows,))])
                wm, wlo = packing.matrix(w.contiguous(), prec)
                return wm, wlo, b.contiguous()
            V = self.ffn_out.weight.shape[0]
            return packing.Packed(enc=lin(self.enc_ffn), pred=lin(self.pred_ffn), out=lin(self.ffn_out, V & 1), V=V, Vp=V + (V & 1))
        params = [self.enc_ffn.weight, self.enc_ffn.bias, self.pred_ffn.weight, self.pred_ffn.bias, self.ffn_out.weight, self.ffn_out.bias]
        return self._pack.get(params, prec, build)

    @staticmethod
    def _rows(t, axis, what):
        """(B, T, X) or the 4-D form with a singleton at `axis` -> (B, T, f32 contiguous [B*T, X])."""
        if t.dim() == 4:
            if t.size(axis) != 1:
                raise ValueError("TransducerJoint: %s of shape %s is already broadcast; pass (B, N, X) or a singleton axis %d"
                                 % (what, tuple(t.shape), axis))
            t = t.squeeze(axis)
        if t.dim() != 3:
            raise ValueError("TransducerJoint: %s must be 3-D or 4-D, got %s" % (what, tuple(t.shape)))
        B, N, X = t.shape
        return B, N, (t if t.dtype == torch.float32 else t.float()).contiguous().view(B * N, X)

    def forward(self, enc_out, pred_out, pre_project=True):
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("TransducerJoint: backward is not built yet (forward only)")
        cfm.require_hip(enc_out, pred_out)
        prec = cfm.resolve_precision(self)
        pk = self._weights(prec)
        B, T, e = self._rows(enc_out, 2, "enc_out")
        Bp, U, p = self._rows(pred_out, 1, "pred_out")
        if Bp != B:
            raise ValueError("TransducerJoint: batch sizes differ (%d, %d)" % (B, Bp))
        if pre_project:
            e = cfm.gemm(e, pk.enc[0], bias=pk.enc[2], w_lo=pk.enc[1], out_dtype=torch.float32)
            p = cfm.gemm(p, pk.pred[0], bias=pk.pred[2], w_lo=pk.pred[1], out_dtype=torch.float32)
        if e.shape[1] != p.shape[1] or e.shape[1] != pk.out[0].shape[1]:
            raise ValueError("TransducerJoint: join dimensions differ (%d, %d, ffn_out expects %d)" % (e.shape[1], p.shape[1], pk.out[0].shape[1]))
        a = cfm.joint_act(e, p, B, T, U, prec.act_dtype)
        out = cfm.gemm(a, pk.out[0], bias=pk.out[2], w_lo=pk.out[1], out_dtype=self.out_dtype)
        out = out.view(B, T, U, pk.Vp)
        return out if pk.Vp == pk.V else out[..., :pk.V]
