"""Global cepstral mean/variance normalisation (reference src/cmvn.py:5-33): (x - mean) * istd with the statistics held
as buffers.  Out of the hot-path scope (BASELINE configs run with cmvn=None); provided so that an
encoder built with a stats file still runs.  Elementwise, on whatever device the buffers live on."""
import torch

from utils import load_cmvn


class GlobalCMVN(torch.nn.Module):

    def __init__(self, cmvn_path, norm_var: bool = True):
        super().__init__()
        mean, istd = load_cmvn(cmvn_path)
        if mean.shape != istd.shape:
            raise ValueError("cmvn mean/istd shapes differ: %s vs %s" % (tuple(mean.shape), tuple(istd.shape)))
        self.norm_var = norm_var
        self.register_buffer("mean", mean)
        self.register_buffer("istd", istd)

    def forward(self, x: torch.Tensor):
        y = x - self.mean
        return y * self.istd if self.norm_var else y
