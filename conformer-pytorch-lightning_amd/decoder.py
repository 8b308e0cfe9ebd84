"""CTC head -- drop-in for the class of the same name in the reference's src/decoder.py:7-23 (forward / loss evaluation only).

    logits = ctc_lo(dropout(encoder_out))                      cfm_gemm, f32 logits, vocabulary padded to a multiple of 4 columns
    loss   = CTCLoss(sum)(log_softmax(logits), labels, enc_lens, label_lens) / labels.size(1)      cfm_ctc_nll + a host-side sum

Same constructor arguments and parameter names (`ctc_lo.weight`, `ctc_lo.bias`) as the reference.  Quirk Q7 is kept: the reference
calls F.dropout with its default training=True, i.e. it drops activations even in eval mode; with dropout > 0 this module does the same
(and is then as random as the reference), parity is defined and tested at dropout = 0.  In train mode `forward` is differentiable
(cfm/autograd.py CTCLossFn: the beta recursion and d loss / d logits in csrc/ctc.hip, the projection's gradients through cfm_gemm /
cfm_gemm_tn); `nll` is loss evaluation only.
"""
import torch
import torch.nn as nn

import cfm
from cfm import packing


class CTCDecoder(nn.Module):

    def __init__(self, vocab_size, encoder_dim, dropout):
        super().__init__()
        self.ctc_lo = nn.Linear(encoder_dim, vocab_size)
        self.dropout = dropout
        self._pack = packing.PackCache()

    def _weights(self, prec):
        def build():
            V, D = self.ctc_lo.weight.shape
            Vp = (V + 3) // 4 * 4                              # cfm_gemm wants N % 4 == 0; the pad columns are never read by cfm_ctc_nll
            w = torch.zeros((Vp, D), dtype=torch.float32, device=self.ctc_lo.weight.device)
            w[:V] = self.ctc_lo.weight.detach()
            b = torch.zeros((Vp,), dtype=torch.float32, device=w.device)
            b[:V] = self.ctc_lo.bias.detach()
            wm, wlo = packing.matrix(w, prec)
            return packing.Packed(w=wm, w_lo=wlo, b=b, V=V, Vp=Vp)
        return self._pack.get([self.ctc_lo.weight, self.ctc_lo.bias], prec, build)

    def nll(self, encoder_out, encoder_out_lens, padded_labels, label_lengths):
        """Per-utterance CTC negative log-likelihood, f32 [B] (loss evaluation: not differentiable -- use forward in train mode)."""
        cfm.require_hip(encoder_out)
        prec = cfm.resolve_precision(self)
        pk = self._weights(prec)
        x = nn.functional.dropout(encoder_out, self.dropout)      # training=True by default, as decoder.py:19 (quirk Q7)
        B, T, D = x.shape
        x2 = (x if x.dtype == torch.float32 else x.float()).contiguous().view(B * T, D)
        logits = cfm.gemm(x2, pk.w, bias=pk.b, w_lo=pk.w_lo, out_dtype=torch.float32).view(B, T, pk.Vp)
        dev = x.device
        i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
        return cfm.ctc_nll(logits, pk.V, i32(encoder_out_lens), i32(padded_labels), i32(label_lengths))

    def forward(self, encoder_out, encoder_out_lens, padded_labels, label_lengths):
        if self.training:
            # differentiable path (cfm/autograd.py CTCLossFn): projection, alpha AND beta recursions, gradients for ctc_lo and the encoder
            from cfm import autograd as ag
            cfm.require_hip(encoder_out)
            dev = encoder_out.device
            i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
            x = nn.functional.dropout(encoder_out, self.dropout)          # training=True by default, as decoder.py:19 (quirk Q7); a torch op
            return ag.CTCLossFn.apply(x, self, cfm.resolve_precision(self), i32(encoder_out_lens), i32(padded_labels), i32(label_lengths),
                                      self.ctc_lo.weight, self.ctc_lo.bias)
        loss = self.nll(encoder_out, encoder_out_lens, padded_labels, label_lengths).sum()
        return loss / padded_labels.size(1)

    def forward_window(self, rows, groups):
        """Train mode: the CTC losses of an accumulation window in one projection.  rows f32 [sum B_g*T'_g, D]: the encoder outputs of the window's
        micro-batches as ONE row matrix (ConformerEncoder.forward_window(..., return_rows=True)); groups: [(B, T', encoder_out_lens, padded_labels,
        label_lengths)] in row order.  Returns a 1-D tensor of the micro-batches' losses, entry g equal to forward() on micro-batch g alone
        (decoder.py:18-23; the always-on F.dropout of :19 draws one mask over all rows instead of one per micro-batch)."""
        if not self.training:
            raise RuntimeError("CTCDecoder.forward_window is the train-mode path; in eval mode call forward per batch")
        from cfm import autograd as ag
        cfm.require_hip(rows)
        dev = rows.device
        i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
        x = nn.functional.dropout(rows, self.dropout)                    # training=True by default, as decoder.py:19 (quirk Q7)
        gs = [(int(B), int(T), i32(el), i32(lab), i32(ll)) for B, T, el, lab, ll in groups]
        return ag.CTCWindowLossFn.apply(x, self, cfm.resolve_precision(self), gs, self.ctc_lo.weight, self.ctc_lo.bias)
