"""Mask builders (device kernels, bit-exact) and the small host-side helpers of the reference's ``src/utils.py``.

The mask half is on the hot path:
  make_pad_mask          utils.py:84-93   -> one launch of cfm_valid_mask (inverted: True = padding)
  subsequent_chunk_mask  utils.py:96-111  -> one launch of cfm_chunk_mask instead of a T'-iteration python loop of
                                             slice assignments (T' tiny kernels per forward in the reference)
  make_attn_mask         utils.py:115-160 -> same selector logic, host RNG draws kept (torch.randint(...).item())
The reference's label helpers (pad_list, load_vocabs, add_blank, add_sos_eos, reverse_sequence: utils.py:31-81,171-190) are host-side
label preparation, OUT OF SCOPE for this build and deliberately not provided: a deployment keeps the reference's own
``utils.py`` for them and takes only the three mask builders from here.  ``load_cmvn`` stays: GlobalCMVN
(cmvn.py) is on the path's "next" list and is folded into the first convolution.
"""
import json
import math

import torch
import torch.nn as nn

import cfm


def make_pad_mask(input_lengths, max_seq_len):
    """bool (B, max_seq_len), True where the frame index is >= the utterance length."""
    return ~cfm.valid_mask(input_lengths, int(max_seq_len))


def subsequent_chunk_mask(size, chunk_size, num_left_chunks, device):
    """bool (size,size): row i sees columns [max((i//c - left)*c, 0), min((i//c + 1)*c, size)); left < 0: from 0."""
    return cfm.chunk_mask(int(size), int(chunk_size), int(num_left_chunks), device)


def make_attn_mask(inputs, inputs_pad_mask, use_dynamic_chunk, use_dynamic_left_chunk, decoding_chunk_size,
                   static_chunk_size, num_decoding_left_chunks):
    """Select the attention mask for a forward pass: padding only (B,1,T') or padding & chunk window (B,T',T')."""
    frames = inputs.size(1)
    if use_dynamic_chunk:
        left = -1
        if decoding_chunk_size < 0:
            width = frames
        elif decoding_chunk_size > 0:
            width, left = decoding_chunk_size, num_decoding_left_chunks
        else:
            # training-time random chunk width: one host draw; above half the utterance means full context
            width = torch.randint(1, frames, (1,)).item()
            if width > frames // 2:
                width = frames
            else:
                width = width % 25 + 1
                if use_dynamic_left_chunk:
                    left = torch.randint(0, frames - 1, (1,)).item()
        return cfm.attn_mask_combine(inputs_pad_mask, subsequent_chunk_mask(frames, width, left, inputs.device))
    if static_chunk_size > 0:
        window = subsequent_chunk_mask(frames, static_chunk_size, num_decoding_left_chunks, inputs.device)
        return cfm.attn_mask_combine(inputs_pad_mask, window)
    return inputs_pad_mask


# ------------------------------------------------------------------------------------------- global CMVN statistics (cmvn.py)
def load_cmvn(json_cmvn_file):
    """Kaldi-style accumulated stats {mean_stat, var_stat, frame_num} -> (mean, 1/std) float tensors."""
    with open(json_cmvn_file) as f:
        stats = json.load(f)
    n = stats['frame_num']
    mean = [m / n for m in stats['mean_stat']]
    istd = []
    for second, mu in zip(stats['var_stat'], mean):
        var = max(second / n - mu * mu, 1.0e-20)
        istd.append(1.0 / math.sqrt(var))
    return torch.tensor(mean), torch.tensor(istd)
