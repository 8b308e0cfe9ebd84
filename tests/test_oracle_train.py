"""CPU: the oracle's TRAIN-mode path (BatchNorm batch statistics, float64 CTC alpha-beta) under torch.autograd against the losses,
gradients and BatchNorm running statistics produced by running the reference itself in train mode (tests/golden/make_golden.py
gen_train; every dropout probability 0).  This pins the gradient oracle the GPU backward kernels are checked against."""
import numpy as np
import pytest
import torch

import synth
from conftest import grad_err, grad_ref_max, load_golden
from oracle import conformer_oracle as O
from test_oracle_golden import FFN_SHAPES, attn_shapes, conv_shapes, encoder_shapes, layer_shapes, sub_shapes, table

TOL = 2e-4          # fp32 backward through 2..12 layers on two different operation orders


def leaf_table(shapes, seed, prefix=""):
    P = table(shapes, seed, prefix)
    for k, v in P.items():
        if v.dtype == torch.float32 and not any(t in k for t in ("running_", "num_batches")):
            v.requires_grad_(True)
    return P


@pytest.mark.parametrize("name", ["train_cfg1", "train_cfg1_chunk", "train_cfg2s"])
def test_encoder_ctc_gradients(name):
    g, meta = load_golden(name)
    cfg = meta["cfg"]
    P = leaf_table(encoder_shapes(cfg), meta["wseed"], "enc.")
    P.update(leaf_table({"ctc_lo.weight": (meta["V"], cfg["encoder_dim"]), "ctc_lo.bias": (meta["V"],)}, meta["cseed"], "ctc."))
    x = torch.from_numpy(synth.fbank(meta["xseed"], meta["batch"], meta["frames"]))
    bn = {}
    Pe = {k[4:]: v for k, v in P.items() if k.startswith("enc.")}
    y, m = O.encoder_forward(Pe, O.Config(**dict(cfg, **meta["ctor"])), x, meta["lens"], train=True, bn_out=bn,
                             decoding_chunk_size=meta["fw"].get("decoding_chunk_size", 0),
                             num_decoding_left_chunks=meta["fw"].get("num_decoding_chunk_size", -1))
    enc_lens = m.squeeze(1).sum(1).numpy()
    assert np.array_equal(enc_lens, g["enc_lens"])
    assert grad_err(g, "y", y.detach().numpy()) < 5e-5
    loss = O.ctc_head_loss_autograd({k[4:]: v for k, v in P.items() if k.startswith("ctc.")}, "", y, enc_lens, g["labels"], g["label_lens"])
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 2e-5 * abs(float(g["loss"][0]))
    loss.backward()
    names = [k for k, v in P.items() if v.requires_grad]
    floor = 1e-3 * max(grad_ref_max(g, "grad:" + k) for k in names)
    worst = max((grad_err(g, "grad:" + k, P[k].grad.numpy(), floor), k) for k in names)
    assert worst[0] < TOL, worst
    for prefix, (rm, rv) in bn.items():
        assert np.abs(rm.numpy() - g["bn:" + prefix + "norm.running_mean"]).max() < 1e-5
        assert np.abs(rv.numpy() - g["bn:" + prefix + "norm.running_var"]).max() < 1e-5
        assert int(g["bn:" + prefix + "norm.num_batches_tracked"]) == 1


def _check(g, tag, out, x, P, floor_scale=1e-2):   # structurally-zero gradients (depthwise bias under BatchNorm, pos_bias_v) are rounding noise ~1e-5 of the largest one
    assert grad_err(g, tag + ":out", out.detach().numpy()) < 2e-5, tag
    keys = [k for k in P if P[k].requires_grad]
    floor = floor_scale * max([grad_ref_max(g, tag + ":grad:" + k) for k in keys] + [grad_ref_max(g, tag + ":dx") if x.grad is not None else 0.0])
    if x.grad is not None:
        assert grad_err(g, tag + ":dx", x.grad.numpy(), floor) < TOL, tag
    for k in keys:
        assert int(g[tag + ":finite:" + k][0]) == 1, (tag, k)
        assert grad_err(g, tag + ":grad:" + k, P[k].grad.numpy(), floor) < TOL, (tag, k)


def test_module_gradients():
    g, meta = load_golden("train_mods_d144")
    D, H, FF, K, B, T = (meta[k] for k in ("D", "H", "FF", "K", "B", "T"))
    pad = torch.from_numpy(~O.pad_mask(meta["lens"], T)).unsqueeze(1)
    chunk_all = torch.from_numpy(O.chunk_mask(T, 5, 1)).unsqueeze(0).expand(B, T, T)
    rel = O.rel_pos_table(D)
    pos_b = rel[0:B].unsqueeze(1)

    def go(tag, shapes, wseed, xseed, gseed, fn, xshape=(B, T, D), fbank=False):
        P = leaf_table(shapes, wseed)
        x = torch.from_numpy(synth.fbank(xseed, *xshape[:2]) if fbank else synth.normal(xseed, xshape)).requires_grad_(not fbank)
        out = fn(P, x)
        G = torch.from_numpy(synth.normal(gseed, tuple(out.shape)))
        (out * G).sum().backward()
        _check(g, tag, out, x, P)
        return P

    go("ffn", FFN_SHAPES(D, FF), 31, 41, 61, lambda P, x: O.ffn(P, "", x))
    go("relmhsa_pad", attn_shapes(D, H), 32, 42, 62, lambda P, x: O.rel_mhsa(P, "", x, pad, pos_b, None, H)[0])
    go("relmhsa_chunk", attn_shapes(D, H), 32, 42, 63, lambda P, x: O.rel_mhsa(P, "", x, chunk_all, pos_b, None, H)[0])
    go("relmhsa_chunkpad", attn_shapes(D, H), 32, 42, 64, lambda P, x: O.rel_mhsa(P, "", x, chunk_all & pad, pos_b, None, H)[0])
    go("mhsa_pad", attn_shapes(D, H, rel=False), 36, 42, 65, lambda P, x: O.mhsa(P, "", x, pad, None, H)[0])
    bn = {}
    go("conv_pad", conv_shapes(D, K), 33, 43, 66, lambda P, x: O.conv_module(P, "", x, pad, train=True, bn_out=bn))
    assert np.abs(bn[""][0].numpy() - g["conv_pad:running_mean"]).max() < 1e-6
    assert np.abs(bn[""][1].numpy() - g["conv_pad:running_var"]).max() < 1e-6
    padf = torch.from_numpy(~O.pad_mask(meta["sub_lens"], 83)).unsqueeze(1)
    go("sub", sub_shapes(D), 34, 44, 67, lambda P, x: O.subsampling(P, "", x, padf, rel, 0, True)[0], xshape=(3, 83), fbank=True)
    go("layer", layer_shapes(D, H, FF, K), 35, 45, 68, lambda P, x: O.encoder_layer(P, "", x, pad, pos_b, pad, None, H, True, train=True)[0])
    go("layer_norel", layer_shapes(D, H, FF, K, rel=False), 37, 45, 69,
       lambda P, x: O.encoder_layer(P, "", x, pad, None, pad, None, H, False, train=True)[0])


def test_ctc_gradient_matches_torch():
    """the float64 alpha-beta gradient against torch's own CTC backward (the op the reference calls, decoder.py:13,21)."""
    rs = np.random.RandomState(5)
    T, V = 23, 11
    logits = torch.from_numpy(rs.standard_normal((2, T, V)).astype(np.float32)).requires_grad_(True)
    labels = np.array([[3, 3, 5, 1, 0], [7, 2, 0, 0, 0]])
    enc_lens, label_lens = np.array([23, 17]), np.array([4, 2])
    O._CTCHead.apply(logits, enc_lens, labels, label_lens).backward()
    ref = logits.detach().clone().requires_grad_(True)
    torch.nn.functional.ctc_loss(ref.transpose(0, 1).log_softmax(2), torch.from_numpy(labels), torch.from_numpy(enc_lens), torch.from_numpy(label_lens),
                                 reduction="sum").backward()
    assert np.abs(logits.grad.numpy() - ref.grad.numpy()).max() < 1e-5
