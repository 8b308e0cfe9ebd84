"""Mask builders (device kernels, bit-exact) and the small host-side helpers of the reference's ``src/utils.py``.

The mask half is on the hot path:
  make_pad_mask          utils.py:84-93   -> one launch of cfm_valid_mask (inverted: True = padding)
  subsequent_chunk_mask  utils.py:96-111  -> one launch of cfm_chunk_mask instead of a T'-iteration python loop of
                                             slice assignments (T' tiny kernels per forward in the reference)
  make_attn_mask         utils.py:115-160 -> same selector logic, host RNG draws kept (torch.randint(...).item())
The label helpers below them are host-side data preparation, outside the accelerated path; they are provided so the
module can stand in for the reference's ``utils`` when this directory shadows it on sys.path.
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


# ------------------------------------------------------------------------------------------- host-side helpers
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


def pad_list(xs, pad_value):
    """List of (T_i, *) tensors -> (B, T_max, *) padded with pad_value."""
    longest = max(x.size(0) for x in xs)
    out = xs[0].new_full((len(xs), longest) + tuple(xs[0].shape[1:]), pad_value)
    for row, x in zip(out, xs):
        row[:x.size(0)] = x
    return out


def load_vocabs(vocab_path):
    table = {}
    with open(vocab_path) as f:
        for line in f:
            token, idx = line.strip().split(' ')
            table[token] = int(idx)
    return table, len(table)


def add_blank(targets, blank, ignore_id):
    """Prepend a blank column and turn padding (ignore_id) into blank."""
    lead = torch.full((targets.size(0), 1), blank, dtype=torch.long, device=targets.device)
    out = torch.cat([lead, targets], dim=1)
    return torch.where(out == ignore_id, blank, out)


def make_subsequent_mask(length, device):
    idx = torch.arange(length, device=device)
    return idx.unsqueeze(0) <= idx.unsqueeze(1)


def add_sos_eos(targets, sos, eos, ignore_id):
    s = torch.tensor([sos], dtype=torch.long, device=targets.device)
    e = torch.tensor([eos], dtype=torch.long, device=targets.device)
    seqs = [row[row != ignore_id] for row in targets]
    return pad_list([torch.cat([s, q]) for q in seqs], eos), pad_list([torch.cat([q, e]) for q in seqs], ignore_id)


def reverse_sequence(targets, target_lengths, ignore_id):
    flipped = [torch.flip(row.int()[:n], [0]) for row, n in zip(targets, target_lengths)]
    return nn.utils.rnn.pad_sequence(flipped, True, ignore_id)
