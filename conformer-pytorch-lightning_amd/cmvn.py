"""Global cepstral mean/variance normalisation (reference src/cmvn.py:5-33): (x - mean) * istd with the statistics held
as buffers.  BASELINE configs run with cmvn=None; an encoder built with a stats file folds these two operations into the first
convolution's tap loads (cfm_conv1_relu, bit-identical; encoder.ConformerEncoder._cmvn_args), so this forward is only used when the
module is called on its own or the statistics do not live on the input's device."""
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
