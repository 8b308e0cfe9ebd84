"""Mask builders (device kernels, bit-exact) and the small host-side helpers of the reference's ``src/utils.py``.

The mask half is on the hot path:
  make_pad_mask          utils.py:84-93   -> one launch of cfm_valid_mask (inverted: True = padding)
  subsequent_chunk_mask  utils.py:96-111  -> one launch of cfm_chunk_mask instead of a T'-iteration python loop of
                                             slice assignments (T' tiny kernels per forward in the reference)
  make_attn_mask         utils.py:115-160 -> same selector logic, host RNG draws kept (torch.randint(...).item())
The reference's label helpers (pad_list, load_vocabs, add_blank, add_sos_eos, reverse_sequence, make_subsequent_mask: utils.py:31-81,
163-190) are host-side label preparation, off the hot path -- but with this directory shadowing ``src/`` on sys.path (the
file-swap route of INTEGRATION.md) the reference's own model.py / decoder.py (`from utils import *`) and executor.py
(`from utils import load_vocabs`) import them from HERE, so they are provided, as plain torch host code with the reference's signatures and
results (pinned by tests/golden/labels.npz, produced by the reference's functions).  ``load_cmvn`` feeds GlobalCMVN (cmvn.py), which is
folded into the first convolution.
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


# ------------------------------------------------------------------- host-side label helpers (utils.py:31-81,163-190), off the hot path
def pad_list(xs, pad_value):
    """[(T_1, *), ..., (T_B, *)] -> (B, max T, *) filled with pad_value behind each sequence; dtype / device of the first entry."""
    longest = max(int(x.size(0)) for x in xs)
    out = xs[0].new_full((len(xs), longest) + tuple(xs[0].shape[1:]), pad_value)
    for row, x in zip(out, xs):
        row[:x.size(0)] = x
    return out


def load_vocabs(vocab_path):
    """`word index` per line -> ({word: index}, size)."""
    table = {}
    with open(vocab_path) as f:
        for line in f:
            word, idx = line.strip().split(' ')
            table[word] = int(idx)
    return table, len(table)


def add_blank(targets, blank, ignore_id):
    """(B, U) -> (B, U+1): a blank in front of every row, padding (ignore_id) turned into blanks (the RNN-T predictor's input)."""
    lead = targets.new_full((targets.size(0), 1), blank, dtype=torch.long)
    out = torch.cat([lead, targets], dim=1)
    return out.masked_fill(out == ignore_id, blank)


def make_subsequent_mask(length, device):
    """bool (length, length), lower triangle incl. the diagonal: position i sees positions <= i."""
    idx = torch.arange(length, device=device)
    return idx.unsqueeze(0) <= idx.unsqueeze(1)


def add_sos_eos(targets, sos, eos, ignore_id):
    """Padded labels -> (sos + labels padded with eos, labels + eos padded with ignore_id)."""
    head = targets.new_tensor([sos], dtype=torch.long)
    tail = targets.new_tensor([eos], dtype=torch.long)
    rows = [row[row != ignore_id] for row in targets]
    return pad_list([torch.cat([head, r]) for r in rows], eos), pad_list([torch.cat([r, tail]) for r in rows], ignore_id)


def reverse_sequence(targets, target_lengths, ignore_id):
    """Each row's first target_lengths[b] labels in reverse order (int32), padded with ignore_id to the longest."""
    flipped = [row.int()[:int(n)].flip(0) for row, n in zip(targets, target_lengths)]
    return nn.utils.rnn.pad_sequence(flipped, batch_first=True, padding_value=ignore_id)
