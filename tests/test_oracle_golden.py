"""CPU: the oracle (oracle/conformer_oracle.py) against the fixtures produced by the reference itself.

This is what "parity pinned" means for the oracle: every function on the hot path is compared with the
reference's own output on the same (regenerated) inputs and weights.  fp32 tolerance: 2e-5 of max|ref|.
"""
import json

import numpy as np
import pytest
import torch

import synth
from conftest import check_joint_case, joint_case_inputs, load_golden
from oracle import conformer_oracle as O

TOL = 2e-5


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def table(shapes, seed, prefix=""):
    return {prefix + k: torch.from_numpy(v) for k, v in synth.fill_state(shapes, seed).items()}


FFN_SHAPES = lambda D, FF: {"w_1.weight": (FF, D), "w_1.bias": (FF,), "w_2.weight": (D, FF), "w_2.bias": (D,)}


def attn_shapes(D, H, rel=True):
    s = {}
    for n in ("q", "k", "v", "out"):
        s["linear_%s.weight" % n] = (D, D)
        s["linear_%s.bias" % n] = (D,)
    if rel:
        s["linear_pos.weight"] = (D, D)
        s["pos_bias_u"] = (H, D // H)
        s["pos_bias_v"] = (H, D // H)
    return s


def conv_shapes(D, K):
    return {"pointwise_conv1.weight": (2 * D, D, 1), "pointwise_conv1.bias": (2 * D,),
            "depthwise_conv.weight": (D, 1, K), "depthwise_conv.bias": (D,),
            "norm.weight": (D,), "norm.bias": (D,), "norm.running_mean": (D,), "norm.running_var": (D,),
            "norm.num_batches_tracked": (),
            "pointwise_conv2.weight": (D, D, 1), "pointwise_conv2.bias": (D,)}


def sub_shapes(D, F=80):
    fp = ((F - 1) // 2 - 1) // 2
    return {"conv.0.weight": (D, 1, 3, 3), "conv.0.bias": (D,), "conv.2.weight": (D, D, 3, 3), "conv.2.bias": (D,),
            "out.0.weight": (D, D * fp), "out.0.bias": (D,)}


def layer_shapes(D, H, FF, K, rel=True):
    s = {}
    for k, v in FFN_SHAPES(D, FF).items():
        s["feed_forward." + k] = v
        s["feed_forward_macaron." + k] = v
    for k, v in attn_shapes(D, H, rel).items():
        s["self_attn." + k] = v
    for k, v in conv_shapes(D, K).items():
        s["conv_module." + k] = v
    for n in ("ff", "ff_macaron", "mha", "conv", "final"):
        s["norm_%s.weight" % n] = (D,)
        s["norm_%s.bias" % n] = (D,)
    return s


def encoder_shapes(cfg):
    D, H, FF, K, L = cfg["encoder_dim"], cfg["num_heads"], cfg["hidden_dim"], cfg["kernel_size"], cfg["encoder_num_layers"]
    s = {"embed." + k: v for k, v in sub_shapes(D, cfg["input_dim"]).items()}
    for li in range(L):
        for k, v in layer_shapes(D, H, FF, K, cfg["use_relative"]).items():
            s["encoders.%d.%s" % (li, k)] = v
    s["after_norm.weight"] = (D,)
    s["after_norm.bias"] = (D,)
    return s


# ------------------------------------------------------------------ masks (bit-exact)
def test_masks_bit_exact():
    g, _ = load_golden("masks")
    assert np.array_equal(O.pad_mask([5, 0, 9, 3], 9).astype(np.uint8), g["pad_5_0_9_3__9"])
    for size, c, left in [(17, 4, -1), (17, 4, 2), (16, 16, 0), (9, 1, 0), (9, 3, 1), (5, 8, -1), (49, 4, 2), (1, 1, 0)]:
        assert np.array_equal(np.packbits(O.chunk_mask(size, c, left).astype(np.uint8)), g["chunk_%d_%d_%d" % (size, c, left)]), (size, c, left)
    T = 320
    valid = ~O.pad_mask(np.arange(0, T + 1), T)[:, None, :]
    sub = O.subsample_mask(valid)
    assert list(sub.shape) == list(g["sub_mask_T320_shape"])
    assert np.array_equal(np.packbits(sub.astype(np.uint8)), g["sub_mask_T320_packed"])
    assert np.array_equal(sub.sum(-1)[:, 0], g["sub_valid_count_T320"])
    assert np.array_equal(np.array([O.subsampled_len(t) for t in range(7, T + 1)]), g["tprime_T7_320"])
    # valid T' = ceil((L-6)/4) for L >= 7
    for L in range(7, T + 1):
        assert g["sub_valid_count_T320"][L] == -(-(L - 6) // 4)
    pad = ~O.pad_mask([13, 8], 13)[:, None, :]
    assert np.array_equal(O.attn_mask(pad, 13, True, False, -1, -1, -1), g["attn_dyn_full"].astype(bool))
    assert np.array_equal(O.attn_mask(pad, 13, True, False, 3, -1, 1), g["attn_dyn_c3_l1"].astype(bool))
    assert np.array_equal(O.attn_mask(pad, 13, False, False, 0, 4, -1), g["attn_static_c4"].astype(bool))
    assert np.array_equal(O.attn_mask(pad, 13, False, False, 0, 4, 0), g["attn_static_c4_l0"].astype(bool))
    assert np.array_equal(O.attn_mask(pad, 13, False, False, 0, -1, -1), g["attn_none"].astype(bool))


# ------------------------------------------------------------------ module level
def test_modules_against_reference():
    g, meta = load_golden("mods_d144")
    D, H, FF, K, B, T = (meta[k] for k in ("D", "H", "FF", "K", "B", "T"))
    lens = meta["lens"]
    pad = torch.from_numpy(~O.pad_mask(lens, T)).unsqueeze(1)
    chunk = torch.from_numpy(O.chunk_mask(T, 5, 1)).unsqueeze(0) & pad
    assert np.array_equal((chunk.sum(-1) == 0).numpy().astype(np.uint8), g["relmhsa_chunk_fullmasked_rows"])
    assert g["relmhsa_chunk_fullmasked_rows"].sum() > 0          # the fixture really has fully-masked rows (Q2)

    rel = O.rel_pos_table(D)
    ab = O.abs_pos_table(D)
    # host transcendental functions differ in the last bit between CPU models: tolerance, not bit equality
    assert np.abs(rel[:64].numpy() - g["rel_pe_0_64"]).max() < 1e-5
    assert np.abs(rel[4990:5000].numpy() - g["rel_pe_4990_5000"]).max() < 2e-3
    assert np.abs(ab[:64].float().numpy() - g["abs_pe_0_64"]).max() < 1e-3
    assert np.abs(ab[4990:5000].float().numpy() - g["abs_pe_4990_5000"]).max() < 3e-3
    assert np.abs(O.rel_pos_table(256)[1000:1004].numpy() - g["rel_pe256_1000_1004"]).max() < 5e-4

    x = torch.from_numpy(synth.normal(41, (B, T, D)))
    P = table(FFN_SHAPES(D, FF), 31)
    assert relerr(O.ffn(P, "", x, "swish"), g["ffn_swish"]) < TOL
    assert relerr(O.ffn(P, "", x, "relu"), g["ffn_relu"]) < TOL

    x = torch.from_numpy(synth.normal(42, (B, T, D)))
    P = table(attn_shapes(D, H), 32)
    pos_b = rel[0:B].unsqueeze(1)
    o, c = O.rel_mhsa(P, "", x, pad, pos_b, None, H)
    assert relerr(o, g["relmhsa_pad"]) < TOL and relerr(c, g["relmhsa_pad_cache"]) < TOL
    o, _ = O.rel_mhsa(P, "", x, chunk, pos_b, None, H)
    assert relerr(o, g["relmhsa_chunk"]) < TOL
    o, _ = O.rel_mhsa(P, "", x, None, pos_b, None, H)
    assert relerr(o, g["relmhsa_nomask"]) < TOL
    cache = torch.from_numpy(synth.normal(46, (1, H, 20, 2 * (D // H))))
    o, c = O.rel_mhsa(P, "", x[:1], None, rel[5:5 + 20 + T].unsqueeze(1), cache, H)
    assert relerr(o, g["relmhsa_stream"]) < TOL and relerr(c, g["relmhsa_stream_cache"]) < TOL

    P = table(attn_shapes(D, H, rel=False), 36)
    o, c = O.mhsa(P, "", x, pad, None, H)
    assert relerr(o, g["mhsa_pad"]) < TOL and relerr(c, g["mhsa_pad_cache"]) < TOL
    o, c = O.mhsa(P, "", x[:1], None, cache, H)
    assert relerr(o, g["mhsa_stream"]) < TOL and relerr(c, g["mhsa_stream_cache"]) < TOL

    x = torch.from_numpy(synth.normal(43, (B, T, D)))
    P = table(conv_shapes(D, K), 33)
    assert relerr(O.conv_module(P, "", x, pad), g["conv_pad"]) < TOL
    assert relerr(O.conv_module(P, "", x, None), g["conv_nomask"]) < TOL
    assert list(g["conv_cache_shape"]) == [0, 0, 0]

    xf = torch.from_numpy(synth.fbank(44, 3, 83))
    padf = torch.from_numpy(~O.pad_mask(meta["sub_lens"], 83)).unsqueeze(1)
    P = table(sub_shapes(D), 34)
    y, p, mk = O.subsampling(P, "", xf, padf, rel, 0, True)
    assert relerr(y, g["sub_out"]) < TOL
    assert np.abs(p.numpy() - g["sub_pos"]).max() < 1e-5
    assert np.array_equal(mk.numpy().astype(np.uint8), g["sub_mask"])
    _, p5, _ = O.subsampling(P, "", xf, padf, rel, 5, True)
    assert np.abs(p5.numpy() - g["sub_pos_off5"]).max() < 1e-5
    assert np.abs(rel[7:16].unsqueeze(1).numpy() - g["sub_position_encoding_7_9"]).max() < 1e-5

    x = torch.from_numpy(synth.normal(45, (B, T, D)))
    shapes = layer_shapes(D, H, FF, K)
    manifest = json.loads(bytes(g["layer_manifest"]).decode())
    assert {k: list(v) for k, v in shapes.items()} == manifest      # parameter names/shapes of SURVEY 8b
    P = table(shapes, 35)
    o, ac = O.encoder_layer(P, "", x, pad, pos_b, pad, None, H, True)
    assert relerr(o, g["layer_out"]) < TOL and relerr(ac, g["layer_attn_cache"]) < TOL


# ------------------------------------------------------------------ whole encoder
def run_cfg(name, **fw):
    g, meta = load_golden(name)
    cfg = meta["cfg"]
    P = table(encoder_shapes(cfg), meta["wseed"])
    x = torch.from_numpy(synth.fbank(meta["xseed"], meta["batch"], meta["frames"]))
    return g, meta, P, x


def test_encoder_cfg1():
    g, meta, P, x = run_cfg("enc_cfg1")
    assert {k: list(v) for k, v in encoder_shapes(meta["cfg"]).items()} == meta["state"]
    assert sum(int(np.prod(v)) for k, v in meta["state"].items() if "num_batches" not in k and "running" not in k) == 1591488
    keep = {}
    y, m = O.encoder_forward(P, O.Config(**meta["cfg"]), x, meta["lens"], collect=keep)
    assert np.array_equal(m.numpy().astype(np.uint8), g["mask"])
    assert m.sum(-1).flatten().tolist() == [49, 40]
    assert relerr(keep["embed_out"], g["embed_out"]) < TOL
    assert relerr(keep["layer_out_0"], g["layer_out_0"]) < TOL
    assert relerr(keep["layer_out_1"], g["layer_out_1"]) < TOL
    assert relerr(y, g["y"]) < TOL


def test_encoder_cfg1_chunk_masks():
    g, meta, P, x = run_cfg("enc_cfg1_chunk")
    y, m = O.encoder_forward(P, O.Config(**dict(meta["cfg"], use_dynamic_chunk_size=True)), x, meta["lens"], 4, 2)
    assert relerr(y, g["y_dyn4_left2"]) < TOL
    y, _ = O.encoder_forward(P, O.Config(**dict(meta["cfg"], use_dynamic_chunk_size=True)), x, meta["lens"], -1)
    assert relerr(y, g["y_dynfull"]) < TOL
    y, _ = O.encoder_forward(P, O.Config(**dict(meta["cfg"], static_chunk_size=3)), x, meta["lens"])
    assert relerr(y, g["y_static3"]) < TOL


def test_encoder_cfg1_norel():
    g, meta, P, x = run_cfg("enc_cfg1_norel")
    assert {k: list(v) for k, v in encoder_shapes(meta["cfg"]).items()} == meta["state"]
    keep = {}
    y, m = O.encoder_forward(P, O.Config(**meta["cfg"]), x, meta["lens"], collect=keep)
    assert relerr(keep["embed_out"], g["embed_out"]) < TOL
    assert relerr(keep["layer_out_0"], g["layer_out_0"]) < TOL
    assert relerr(y, g["y"]) < TOL


def test_encoder_cfg2_small():
    g, meta, P, x = run_cfg("enc_cfg2s")
    keep = {}
    y, m = O.encoder_forward(P, O.Config(**meta["cfg"]), x, meta["lens"], collect=keep)
    assert np.array_equal(m.numpy().astype(np.uint8), g["mask"])
    for k in ("embed_out", "layer_out_0", "layer_out_5", "layer_out_11"):
        assert relerr(keep[k], g[k]) < TOL, k
    assert relerr(y, g["y"]) < TOL


def test_encoder_streaming():
    g, meta = load_golden("enc_cfg1_stream")
    cfg = O.Config(**meta["cfg"])
    P = table(encoder_shapes(meta["cfg"]), meta["wseed"])
    x = torch.from_numpy(synth.fbank(meta["xseed"], 1, meta["frames"]))
    assert relerr(O.encoder_forward_chunk_by_chunk(P, cfg, x, 16, 4), g["y_left4"]) < TOL
    assert relerr(O.encoder_forward_chunk_by_chunk(P, cfg, x, 16, -1), g["y_unbounded"]) < TOL
    c0, a0 = O.encoder_forward_chunk(P, cfg, x[:, 0:67], 0, 32, None)
    c1, a1 = O.encoder_forward_chunk(P, cfg, x[:, 64:131], 16, 32, a0)
    c2, a2 = O.encoder_forward_chunk(P, cfg, x[:, 128:195], 32, 32, a1)
    for got, key in ((c0, "chunk0"), (c1, "chunk1"), (c2, "chunk2"), (a0, "cache0"), (a1, "cache1"), (a2, "cache2")):
        assert relerr(got, g[key]) < TOL, key
    assert list(a2.shape) == [2, 4, 32, 72]
    assert list(g["cnn_cache_shape"]) == [2, 0, 0, 0]


def test_flop_model_matches_survey():
    # 22.351 GFLOP/utt at config 2, 0.493 at config 1, 78.256 for config 4's encoder
    assert abs(O.encoder_flops_per_utt(1000, 80, 256, 2048, 15, 12) / 1e9 - 22.351) < 0.01
    assert abs(O.encoder_flops_per_utt(200, 80, 144, 576, 15, 2) / 1e9 - 0.493) < 0.001
    assert abs(O.encoder_flops_per_utt(1000, 80, 512, 2048, 15, 17) / 1e9 - 78.256) < 0.01


def test_ctc_head_oracle_matches_reference():
    """CTCDecoder.forward of the reference (dropout 0) vs the oracle's restatement: Linear + log-softmax + alpha recursion."""
    g, meta = load_golden("ctc_head")
    for c in meta["cases"]:
        n = c["name"]
        P = table(c["state"], c["wseed"])
        enc_out = torch.from_numpy(synth.normal(c["xseed"], (c["B"], c["T"], c["D"]), 1.0))
        loss, nll = O.ctc_head_loss(P, "", enc_out, g[n + "_enc_lens"], g[n + "_labels"], g[n + "_label_lens"])
        assert np.allclose(nll, g[n + "_nll"], rtol=2e-5, atol=1e-4), (n, nll, g[n + "_nll"])
        assert abs(loss - float(g[n + "_loss"][0])) <= 2e-5 * abs(float(g[n + "_loss"][0]))


def test_joint_oracle_matches_reference():
    """TransducerJoint.forward of the reference (tests/golden/joint.npz) vs the oracle's restatement, both entry forms."""
    g, meta = load_golden("joint")
    assert [c["name"] for c in meta["cases"]] == ["small", "vocab5002", "oddvocab", "step"]
    for c in meta["cases"]:
        P = table(c["state"], c["wseed"])
        enc, pred = joint_case_inputs(c)
        out = O.joint_forward(P, "", enc, pred)
        assert check_joint_case(g, c, out.numpy()) < 2e-6, c["name"]
        e = O.linear(enc, P["enc_ffn.weight"], P["enc_ffn.bias"]).unsqueeze(2)
        p = O.linear(pred, P["pred_ffn.weight"], P["pred_ffn.bias"]).unsqueeze(1)
        assert torch.equal(O.joint_forward(P, "", e, p, pre_project=False), out)
