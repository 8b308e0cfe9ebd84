"""Deterministic synthetic weights / inputs shared by the golden generator and the tests.

Everything is drawn from ``numpy.random.RandomState`` (a frozen stream: the same
seed gives the same numbers on every numpy release), never from torch's RNG, so
the committed golden fixtures only need to hold the *outputs* of the reference:
inputs and weights are regenerated bit-identically wherever the tests run.

The fill rule is keyed on the parameter *name* and *shape* of the reference
state_dict (SURVEY.md section 8b lists them); values are deliberately
non-trivial (LayerNorm gain != 1, biases != 0, BatchNorm running stats != (0,1))
so that a kernel that skips an affine term or a bias cannot pass.
"""
import numpy as np


def _fill_one(rs, name, shape):
    leaf = name.split(".")[-1]
    n = int(np.prod(shape)) if len(shape) else 1
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return (0.1 * rs.standard_normal(n)).reshape(shape).astype(np.float32)
    if leaf == "running_var":
        return rs.uniform(0.5, 1.5, n).reshape(shape).astype(np.float32)
    if leaf in ("pos_bias_u", "pos_bias_v"):
        lim = float(np.sqrt(6.0 / (shape[0] + shape[1])))
        return rs.uniform(-lim, lim, n).reshape(shape).astype(np.float32)
    if leaf.startswith("weight_ih_l") or leaf.startswith("weight_hh_l"):      # nn.LSTM matrices (4H, in): torch's own U(-1/sqrt(H), 1/sqrt(H))
        lim = 1.0 / float(np.sqrt(shape[0] // 4))
        return rs.uniform(-lim, lim, n).reshape(shape).astype(np.float32)
    if leaf.startswith("bias_ih_l") or leaf.startswith("bias_hh_l"):
        return rs.uniform(-0.1, 0.1, n).reshape(shape).astype(np.float32)
    if leaf == "weight" and len(shape) == 1:          # LayerNorm / BatchNorm gain
        return (1.0 + 0.1 * rs.standard_normal(n)).reshape(shape).astype(np.float32)
    if leaf == "bias":
        return rs.uniform(-0.1, 0.1, n).reshape(shape).astype(np.float32)
    if leaf == "weight":                               # Linear / ConvNd kernels
        fan_in = int(np.prod(shape[1:]))
        lim = 1.0 / float(np.sqrt(fan_in))
        return rs.uniform(-lim, lim, n).reshape(shape).astype(np.float32)
    raise KeyError("synth: no fill rule for %r %r" % (name, tuple(shape)))


def fill_state(shapes, seed):
    """shapes: {name: shape}.  Returns {name: ndarray}, drawn in sorted-name order."""
    rs = np.random.RandomState(seed)
    return {k: _fill_one(rs, k, tuple(shapes[k])) for k in sorted(shapes)}


def load_synth_(module, seed):
    """Overwrite every entry of module.state_dict() with the synthetic values."""
    import torch
    sd = module.state_dict()
    vals = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed)
    module.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in vals.items()})
    return module


def fbank(seed, batch, frames, dim=80):
    """Synthetic post-CMVN filterbank features ~ N(0,1)."""
    rs = np.random.RandomState(seed)
    return rs.standard_normal((batch, frames, dim)).astype(np.float32)


def normal(seed, shape, scale=1.0):
    rs = np.random.RandomState(seed)
    return (scale * rs.standard_normal(tuple(shape))).astype(np.float32)


def greedy_joint_(joint, vocab_size):
    """After load_synth_: reshape a joint's synthetic parameters so that a greedy search over them is not degenerate (with the plain
    synthetic values the argmax over V classes is almost never the blank and hardly depends on the predictor: every frame would emit
    symbols up to the per-frame cap) -- predictor path x 2, encoder path x 0.7, blank bias raised.  Shared by the golden generator and
    the tests, like every other synthetic value."""
    import torch
    with torch.no_grad():
        joint.pred_ffn.weight.mul_(2.0)
        joint.enc_ffn.weight.mul_(0.7)
        joint.ffn_out.bias[0] += 0.6 if vocab_size < 1000 else 0.8
    return joint
