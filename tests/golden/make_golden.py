#!/usr/bin/env python3
"""Generate the committed golden fixtures by RUNNING THE REFERENCE on CPU.

Run only in the build container (``/root/reference`` does not exist on the GPU
box and nothing under tests/ reads it at test time):

    python tests/golden/make_golden.py

The reference's torch-only modules are imported from ``/root/reference/src``
(read-only); their parameters are overwritten with the deterministic synthetic
values of ``tests/synth.py``; inputs come from the same helper.  Only OUTPUTS
(and a few intermediates) are stored, as float32 ``.npz`` files, so the
fixtures stay small.  Weights/inputs are regenerated from seeds at test time.

Nothing here is reference source text: the fixtures are data.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/ -> synth
sys.path.insert(0, "/root/reference/src")          # reference flat modules

import synth  # noqa: E402

import attention as ref_attention  # noqa: E402
import convolution as ref_convolution  # noqa: E402
import encoder as ref_encoder  # noqa: E402
import encoder_layer as ref_encoder_layer  # noqa: E402
import feedforward as ref_feedforward  # noqa: E402
import utils as ref_utils  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)

CFG1 = dict(input_dim=80, kernel_size=15, encoder_dim=144, dropout=0.1, attention_dropout=0.1,
            pos_enc_dropout=0.1, hidden_dim=576, num_heads=4, encoder_num_layers=2, max_len=5000,
            use_relative=True)
CFG2 = dict(input_dim=80, kernel_size=15, encoder_dim=256, dropout=0.1, attention_dropout=0.1,
            pos_enc_dropout=0.1, hidden_dim=2048, num_heads=4, encoder_num_layers=12, max_len=5000,
            use_relative=True)


def t2n(t):
    t = t.detach().cpu()
    if t.dtype == torch.bool:
        return t.numpy().astype(np.uint8)
    return t.contiguous().numpy()


def save(name, arrays, meta):
    arrays = dict(arrays)
    arrays["meta"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-24s %8.1f KB  %d arrays" % (name, os.path.getsize(path) / 1024.0, len(arrays)))


def build_encoder(cfg, wseed, **extra):
    kw = dict(cfg)
    kw.update(extra)
    enc = ref_encoder.ConformerEncoder(cmvn=None, **kw).eval()
    synth.load_synth_(enc, wseed)
    return enc


def run_encoder(enc, x, lens, keep_layers, **fw):
    keep = {}
    hooks = [enc.embed.register_forward_hook(
        lambda m, i, o: keep.update(embed_out=t2n(o[0]), pos_embed=t2n(o[1])))]
    for li in keep_layers:
        hooks.append(enc.encoders[li].register_forward_hook(
            lambda m, i, o, li=li: keep.update({"layer_out_%d" % li: t2n(o[0])})))
    with torch.no_grad():
        y, m = enc(torch.from_numpy(x), torch.tensor(lens, dtype=torch.int32), **fw)
    for h in hooks:
        h.remove()
    keep["y"] = t2n(y)
    keep["mask"] = t2n(m)
    return keep


def state_manifest(mod):
    return {k: list(v.shape) for k, v in mod.state_dict().items()}


# ----------------------------------------------------------------------------- encoders
def gen_encoders():
    # config 1 (BASELINE.json configs[0]): 2-layer d=144 h=4, B=2, T=200
    x = synth.fbank(21, 2, 200)
    lens = [200, 163]
    enc = build_encoder(CFG1, 11)
    out = run_encoder(enc, x, lens, [0, 1])
    save("enc_cfg1", out, dict(cfg=CFG1, wseed=11, xseed=21, batch=2, frames=200, lens=lens,
                               state=state_manifest(enc)))

    # dynamic-chunk training mask (decoding_chunk_size>0) and static chunk mask
    enc = build_encoder(CFG1, 11, use_dynamic_chunk_size=True)
    o1 = run_encoder(enc, x, lens, [], decoding_chunk_size=4, num_decoding_chunk_size=2)
    o2 = run_encoder(enc, x, lens, [], decoding_chunk_size=-1)
    enc = build_encoder(CFG1, 11, static_chunk_size=3)
    o3 = run_encoder(enc, x, lens, [])
    save("enc_cfg1_chunk", dict(y_dyn4_left2=o1["y"], y_dynfull=o2["y"], y_static3=o3["y"], mask=o1["mask"]),
         dict(cfg=CFG1, wseed=11, xseed=21, batch=2, frames=200, lens=lens))

    # absolute-position variant (use_relative=False): fp16-quantised PE table, plain MHSA
    cfg = dict(CFG1, use_relative=False)
    enc = build_encoder(cfg, 13)
    out = run_encoder(enc, x, lens, [0])
    save("enc_cfg1_norel", dict(y=out["y"], mask=out["mask"], embed_out=out["embed_out"],
                                layer_out_0=out["layer_out_0"]),
         dict(cfg=cfg, wseed=13, xseed=21, batch=2, frames=200, lens=lens, state=state_manifest(enc)))

    # config-2 architecture (12-layer d=256 h=4 ff=2048) on a small ragged batch
    x2 = synth.fbank(22, 3, 120)
    lens2 = [120, 97, 64]
    enc = build_encoder(CFG2, 12)
    out = run_encoder(enc, x2, lens2, [0, 5, 11])
    save("enc_cfg2s", out, dict(cfg=CFG2, wseed=12, xseed=22, batch=3, frames=120, lens=lens2))

    # streaming (config 5 semantics at batch 1): chunk=16, left chunks 4 and unbounded
    enc = build_encoder(CFG1, 11)
    xs = torch.from_numpy(synth.fbank(23, 1, 331))
    arrays = {}
    with torch.no_grad():
        ys, _ = enc.forward_chunk_by_chunk(xs, 16, 4)
        yu, _ = enc.forward_chunk_by_chunk(xs, 16, -1)
        arrays["y_left4"] = t2n(ys)
        arrays["y_unbounded"] = t2n(yu)
        # two explicit forward_chunk calls so the caches themselves are pinned
        empty = torch.zeros((0, 0, 0, 0))
        c0, a0, n0 = enc.forward_chunk(xs[:, 0:67], 0, 32, empty, empty)
        c1, a1, n1 = enc.forward_chunk(xs[:, 64:131], 16, 32, a0, n0)
        c2, a2, n2 = enc.forward_chunk(xs[:, 128:195], 32, 32, a1, n1)
        arrays.update(chunk0=t2n(c0), chunk1=t2n(c1), chunk2=t2n(c2),
                      cache0=t2n(a0), cache1=t2n(a1), cache2=t2n(a2))
        arrays["cnn_cache_shape"] = np.array(list(n2.shape), dtype=np.int64)
    save("enc_cfg1_stream", arrays, dict(cfg=CFG1, wseed=11, xseed=23, frames=331, chunk=16))


# ----------------------------------------------------------------------------- modules
def gen_modules():
    D, H, FF, K, B, T = 144, 4, 576, 15, 3, 37
    arrays = {}
    lens = [37, 30, 19]
    pad = ~ref_utils.make_pad_mask(torch.tensor(lens, dtype=torch.int32), T).unsqueeze(1)   # (B,1,T) True=valid
    chunk = ref_utils.subsequent_chunk_mask(T, 5, 1, torch.device("cpu")).unsqueeze(0) & pad    # (B,T,T)
    empty_mask = torch.ones((0, 0, 0), dtype=torch.bool)
    with torch.no_grad():
        # FFN (swish / relu)
        x = torch.from_numpy(synth.normal(41, (B, T, D)))
        m = synth.load_synth_(ref_feedforward.PositionwiseFeedForwardModule(D, 0.1, FF).eval(), 31)
        arrays["ffn_swish"] = t2n(m(x))
        m = synth.load_synth_(ref_feedforward.PositionwiseFeedForwardModule(D, 0.1, FF, activation="relu").eval(), 31)
        arrays["ffn_relu"] = t2n(m(x))

        # positional tables
        rpe = ref_attention.RelativePositionalEncoding(D, 0.1).eval()
        ape = ref_attention.PositionalEncoding(D, 0.1).eval()
        arrays["rel_pe_0_64"] = t2n(rpe.pe[:64, 0])
        arrays["rel_pe_4990_5000"] = t2n(rpe.pe[4990:5000, 0])
        arrays["abs_pe_0_64"] = t2n(ape.pe[:64, 0].float())
        arrays["abs_pe_4990_5000"] = t2n(ape.pe[4990:5000, 0].float())
        rpe256 = ref_attention.RelativePositionalEncoding(256, 0.0)
        arrays["rel_pe256_1000_1004"] = t2n(rpe256.pe[1000:1004, 0])

        # relative MHSA: batch-path pos_embed (B,1,D), three mask kinds, then KV-cache path at B=1
        x = torch.from_numpy(synth.normal(42, (B, T, D)))
        m = synth.load_synth_(ref_attention.RelativeMultiHeadSelfAttentionModule(D, H, 0.1).eval(), 32)
        pos_b = rpe.pe[0:B]
        o, c = m(x, x, x, pad, pos_b)
        arrays["relmhsa_pad"], arrays["relmhsa_pad_cache"] = t2n(o), t2n(c)
        o, _ = m(x, x, x, chunk, pos_b)
        arrays["relmhsa_chunk"] = t2n(o)
        arrays["relmhsa_chunk_fullmasked_rows"] = t2n((chunk.sum(-1) == 0))
        o, _ = m(x, x, x, empty_mask, pos_b)
        arrays["relmhsa_nomask"] = t2n(o)
        cache = torch.from_numpy(synth.normal(46, (1, H, 20, 2 * (D // H))))
        pos_s = rpe.pe[5:5 + 20 + T]
        o, c = m(x[:1], x[:1], x[:1], empty_mask, pos_s, cache)
        arrays["relmhsa_stream"], arrays["relmhsa_stream_cache"] = t2n(o), t2n(c)

        # plain MHSA (use_relative=False variant)
        m = synth.load_synth_(ref_attention.MultiHeadSelfAttentionModule(D, H, 0.1).eval(), 36)
        o, c = m(x, x, x, pad)
        arrays["mhsa_pad"], arrays["mhsa_pad_cache"] = t2n(o), t2n(c)
        o, c = m(x[:1], x[:1], x[:1], empty_mask, None, cache)
        arrays["mhsa_stream"], arrays["mhsa_stream_cache"] = t2n(o), t2n(c)

        # convolution module (note Q1: third ctor arg lands in the `bias` slot)
        x = torch.from_numpy(synth.normal(43, (B, T, D)))
        m = synth.load_synth_(ref_convolution.ConvolutionModule(D, K, FF).eval(), 33)
        o, c = m(x, pad)
        arrays["conv_pad"] = t2n(o)
        arrays["conv_cache_shape"] = np.array(list(c.shape), dtype=np.int64)
        o, _ = m(x, empty_mask)
        arrays["conv_nomask"] = t2n(o)

        # subsampling front-end
        xf = torch.from_numpy(synth.fbank(44, 3, 83))
        lf = [83, 60, 7]
        padf = ~ref_utils.make_pad_mask(torch.tensor(lf, dtype=torch.int32), 83).unsqueeze(1)
        m = synth.load_synth_(ref_convolution.ConvolutionSubSampling(80, D, ref_attention.RelativePositionalEncoding(D, 0.1)).eval(), 34)
        o, p, mk = m(xf, padf)
        arrays["sub_out"], arrays["sub_pos"], arrays["sub_mask"] = t2n(o), t2n(p), t2n(mk)
        o, p, mk = m(xf, padf, 5)
        arrays["sub_pos_off5"] = t2n(p)
        arrays["sub_position_encoding_7_9"] = t2n(m.position_encoding(7, 9))

        # one full encoder layer
        x = torch.from_numpy(synth.normal(45, (B, T, D)))
        m = synth.load_synth_(ref_encoder_layer.ConformerEncoderLayer(D, K, 0.1, 0.1, FF, H, True).eval(), 35)
        o, mk, ac, cc = m(x, pad, pos_b, pad)
        arrays["layer_out"], arrays["layer_attn_cache"] = t2n(o), t2n(ac)
        arrays["layer_manifest"] = np.frombuffer(json.dumps(state_manifest(m), sort_keys=True).encode(), dtype=np.uint8)
    save("mods_d144", arrays, dict(D=D, H=H, FF=FF, K=K, B=B, T=T, lens=lens, sub_lens=lf,
                                   seeds=dict(ffn=(31, 41), relmhsa=(32, 42), cache=46, mhsa=36, conv=(33, 43),
                                              sub=(34, 44), layer=(35, 45))))


# ----------------------------------------------------------------------------- masks (bit-exact path)
def gen_masks():
    arrays = {}
    dev = torch.device("cpu")
    lens = torch.tensor([5, 0, 9, 3], dtype=torch.int32)
    arrays["pad_5_0_9_3__9"] = t2n(ref_utils.make_pad_mask(lens, 9))
    for size, c, left in [(17, 4, -1), (17, 4, 2), (16, 16, 0), (9, 1, 0), (9, 3, 1), (5, 8, -1), (49, 4, 2), (1, 1, 0)]:
        arrays["chunk_%d_%d_%d" % (size, c, left)] = np.packbits(t2n(ref_utils.subsequent_chunk_mask(size, c, left, dev)))
    # subsampled length rule: for every raw length L in 0..320 on a T=320 utterance, the number of valid
    # frames that survive mask[:, :, 2::2][:, :, 2::2]; and T' for every T in 7..320 from the conv shapes
    T = 320
    pm = ~ref_utils.make_pad_mask(torch.arange(0, T + 1, dtype=torch.int32), T).unsqueeze(1)
    sub = pm[:, :, 2::2][:, :, 2::2]
    arrays["sub_valid_count_T320"] = sub.sum(-1).squeeze(1).numpy().astype(np.int64)
    arrays["sub_mask_T320_packed"] = np.packbits(t2n(sub))
    arrays["sub_mask_T320_shape"] = np.array(list(sub.shape), dtype=np.int64)
    tprime = []
    conv = torch.nn.Sequential(torch.nn.Conv2d(1, 1, 3, 2), torch.nn.Conv2d(1, 1, 3, 2))
    with torch.no_grad():
        for t in range(7, T + 1):
            tprime.append(conv(torch.zeros(1, 1, t, 80)).shape[2])
    arrays["tprime_T7_320"] = np.array(tprime, dtype=np.int64)
    # make_attn_mask selector (deterministic branches only)
    x = torch.zeros(2, 13, 4)
    pad = ~ref_utils.make_pad_mask(torch.tensor([13, 8], dtype=torch.int32), 13).unsqueeze(1)
    arrays["attn_dyn_full"] = t2n(ref_utils.make_attn_mask(x, pad, True, False, -1, -1, -1))
    arrays["attn_dyn_c3_l1"] = t2n(ref_utils.make_attn_mask(x, pad, True, False, 3, -1, 1))
    arrays["attn_static_c4"] = t2n(ref_utils.make_attn_mask(x, pad, False, False, 0, 4, -1))
    arrays["attn_static_c4_l0"] = t2n(ref_utils.make_attn_mask(x, pad, False, False, 0, 4, 0))
    arrays["attn_none"] = t2n(ref_utils.make_attn_mask(x, pad, False, False, 0, -1, -1))
    save("masks", arrays, dict(note="bool arrays stored as uint8 or np.packbits"))


def gen_ctc():
    """CTCDecoder.forward (reference src/decoder.py:7-23) with dropout 0 (it applies dropout with training=True even in eval, quirk
    Q7): the scalar loss the reference returns, plus the per-utterance terms from the same torch op with reduction='none'."""
    import decoder as ref_decoder  # noqa: E402  (torch-only; imports attention / decoder_layer / utils of the reference)
    arrays, cases = {}, []
    for name, (V, D, B, T, Umax, seed) in dict(small=(73, 144, 3, 49, 9, 21), vocab5002=(5002, 256, 2, 60, 12, 22)).items():
        dec = ref_decoder.CTCDecoder(V, D, 0.0).eval()
        synth.load_synth_(dec, seed)
        rs = np.random.RandomState(seed)
        enc_out = torch.from_numpy(synth.normal(seed + 100, (B, T, D), 1.0))
        enc_lens = np.sort(rs.randint(T // 2, T + 1, size=B))[::-1].astype(np.int64).copy()
        enc_lens[0] = T
        label_lens = rs.randint(1, Umax + 1, size=B).astype(np.int64)
        label_lens[0] = Umax
        labels = rs.randint(1, V, size=(B, Umax)).astype(np.int64)
        labels[1, 1:3] = labels[1, 0]                       # repeated labels: the skip transition must be refused there
        for b in range(B):
            labels[b, label_lens[b]:] = 0
        with torch.no_grad():
            loss = dec(enc_out, torch.from_numpy(enc_lens), torch.from_numpy(labels), torch.from_numpy(label_lens))
            probs = dec.ctc_lo(enc_out).transpose(0, 1).log_softmax(2)
            per = torch.nn.CTCLoss(reduction="none")(probs, torch.from_numpy(labels), torch.from_numpy(enc_lens), torch.from_numpy(label_lens))
        arrays[name + "_loss"] = t2n(loss.reshape(1))
        arrays[name + "_nll"] = t2n(per)
        arrays[name + "_enc_lens"], arrays[name + "_labels"], arrays[name + "_label_lens"] = enc_lens, labels, label_lens
        cases.append(dict(name=name, V=V, D=D, B=B, T=T, Umax=Umax, wseed=seed, xseed=seed + 100,
                          state={k: list(v.shape) for k, v in dec.state_dict().items()}))
    save("ctc_head", arrays, dict(cases=cases))


def gen_joint():
    """TransducerJoint.forward (reference src/joint.py:4-38): full (B,T,U,V) logits for a small case, and for the vocabulary of
    BASELINE config 4 (5002, join_dim 512) a few hundred sampled logits plus per-(b,t,u) sums -- the full tensor would be 12 MB --
    and the (1,1,1,V) step shape of the greedy search (model.py:250)."""
    import joint as ref_joint  # noqa: E402
    arrays, cases = {}, []
    for name, (V, E, Pd, J, B, T, U, seed) in dict(small=(74, 144, 96, 64, 2, 7, 5, 31), vocab5002=(5002, 256, 256, 512, 2, 12, 6, 32),
                                                   oddvocab=(73, 144, 96, 64, 2, 5, 3, 34), step=(5002, 256, 256, 512, 1, 1, 1, 33)).items():
        jn = ref_joint.TransducerJoint(V, E, Pd, J).eval()
        synth.load_synth_(jn, seed)
        enc = torch.from_numpy(synth.normal(seed + 100, (B, T, E), 1.0))
        pred = torch.from_numpy(synth.normal(seed + 200, (B, U, Pd), 1.0))
        with torch.no_grad():
            out = jn(enc, pred)
            out_np = jn(jn.enc_ffn(enc), jn.pred_ffn(pred), pre_project=False)      # same numbers through the other entry
        assert tuple(out.shape) == (B, T, U, V) and torch.equal(out, out_np)
        if out.numel() <= 200000:
            arrays[name + "_out"] = t2n(out)
        else:
            rs = np.random.RandomState(seed)
            idx = rs.randint(0, out.numel(), size=512).astype(np.int64)
            arrays[name + "_idx"] = idx
            arrays[name + "_vals"] = t2n(out.reshape(-1)[torch.from_numpy(idx)])
            arrays[name + "_rowsum"] = t2n(out.double().sum(-1).float())
            arrays[name + "_rowabs"] = t2n(out.double().abs().sum(-1).float())
        cases.append(dict(name=name, V=V, E=E, P=Pd, J=J, B=B, T=T, U=U, wseed=seed, eseed=seed + 100, pseed=seed + 200,
                          state={k: list(v.shape) for k, v in jn.state_dict().items()}))
    save("joint", arrays, dict(cases=cases))


GREEDY_CASES = [   # name, V, E (encoder dim), P (predictor output), J, embed, hidden, layers, T', n_steps, seed
    ("small", 73, 144, 96, 64, 48, 80, 2, 23, 3, 51),
    ("cap1", 73, 144, 96, 64, 48, 80, 1, 17, 1, 52),
    ("config4", 5002, 512, 512, 512, 256, 256, 2, 31, 4, 53),
]


def greedy_modules(pred_cls, joint_cls, V, E, Pd, J, emb, hid, layers, seed):
    """predictor + joint with synthetic parameters (tests/synth.py: load_synth_, then greedy_joint_ so that blanks and symbols alternate)."""
    pr = pred_cls(V, emb, Pd, hid, 0.1, layers).eval()
    jn = joint_cls(V, E, Pd, J).eval()
    synth.load_synth_(pr, seed)
    synth.load_synth_(jn, seed + 1)
    synth.greedy_joint_(jn, V)
    return pr, jn


def gen_greedy():
    """Tokens of the greedy search (model.py:215-269) with the REFERENCE's RNNPredictor.forward_step (predictor.py:76-86) and
    TransducerJoint.forward (joint.py:20-38) doing every step; the loop around them is restated here line by line because model.py does not
    import in this container (torchaudio).  Also a continued search: the second half of an utterance started from the first half's
    (token, LSTM state), as greedy_search_streaming_app does (model.py:186-196)."""
    import joint as ref_joint  # noqa: E402
    import predictor as ref_predictor  # noqa: E402

    def search(pr, jn, enc, n_steps, token=None, cache=None, blank=0):
        padding = torch.zeros(1, 1)
        tok = torch.tensor([blank]).reshape(1, 1) if token is None else token
        cache = pr.init_state(tok) if cache is None else cache
        t, hyps, prev, per_frame, pred_out, new_cache = 0, [], True, 0, None, None
        while t < enc.size(1):
            if prev:
                pred_out, new_cache = pr.forward_step(tok, padding, cache)
            k = jn(enc[:, t:t + 1, :], pred_out).log_softmax(dim=-1).argmax(dim=-1).squeeze()
            if k != blank:
                hyps.append(int(k))
                prev = True
                per_frame += 1
                tok = k.reshape(1, 1)
                cache = new_cache
            if k == blank or per_frame >= n_steps:
                if k == blank:
                    prev = False
                t += 1
                per_frame = 0
        return hyps, tok, cache

    arrays, cases = {}, []
    for name, V, E, Pd, J, emb, hid, layers, T, n_steps, seed in GREEDY_CASES:
        pr, jn = greedy_modules(ref_predictor.RNNPredictor, ref_joint.TransducerJoint, V, E, Pd, J, emb, hid, layers, seed)
        lens = [T, T - 5, max(1, T // 2)]
        with torch.no_grad():
            for u, n in enumerate(lens):
                enc = torch.from_numpy(synth.normal(seed + 10 + u, (1, T, E), 1.0))
                hyps, _, _ = search(pr, jn, enc[:, :n], n_steps)
                arrays["%s_utt%d" % (name, u)] = np.asarray(hyps, dtype=np.int64)
                if u == 0:                                              # the same utterance in two calls with carried state
                    a, tok, cache = search(pr, jn, enc[:, :T // 2], n_steps)
                    b, _, _ = search(pr, jn, enc[:, T // 2:], n_steps, token=tok, cache=cache)
                    arrays["%s_utt0_first" % name] = np.asarray(a, dtype=np.int64)
                    arrays["%s_utt0_second" % name] = np.asarray(b, dtype=np.int64)
                    # (a + b differs from hyps only through the "previous output was non-blank" flag, which a new call resets)
        cases.append(dict(name=name, V=V, E=E, P=Pd, J=J, embed=emb, hidden=hid, layers=layers, T=T, n_steps=n_steps, seed=seed, lens=lens))
        print("  greedy %-8s tokens per utterance: %s" % (name, [len(arrays["%s_utt%d" % (name, u)]) for u in range(3)]))
    save("greedy", arrays, dict(cases=cases))


# ----------------------------------------------------------------------------- training (config 3): losses and gradients
GRAD_SAMPLES = 1024


def pack_grad(arrays, key, t, seed):
    """Small tensors in full; big ones as GRAD_SAMPLES sampled entries (indices from RandomState(seed)) plus float64 sum / |sum| / sum of
    squares -- enough to pin every gradient without storing 35 M floats."""
    a = t2n(t).astype(np.float32).reshape(-1)
    if a.size <= 4096:
        arrays[key] = a.reshape(tuple(t.shape))
        return
    idx = np.random.RandomState(seed).randint(0, a.size, size=GRAD_SAMPLES).astype(np.int64)
    arrays[key + "__idx"] = idx
    arrays[key + "__vals"] = a[idx]
    a64 = a.astype(np.float64)
    arrays[key + "__stats"] = np.array([a64.sum(), np.abs(a64).sum(), (a64 * a64).sum(), np.abs(a64).max()], dtype=np.float64)


def ctc_batch(rs, B, V, Umax, enc_valid):
    """labels int64 [B,Umax] in [1,V), lengths <= the number of valid encoder frames / 2 so that an alignment exists."""
    label_lens = np.array([max(1, min(Umax, int(rs.randint(max(1, Umax // 2), Umax + 1)), int(enc_valid[b]) // 2)) for b in range(B)], dtype=np.int64)
    label_lens[0] = min(Umax, int(enc_valid[0]) // 2)
    labels = rs.randint(1, V, size=(B, Umax)).astype(np.int64)
    labels[-1, 1] = labels[-1, 0]                            # one repeated label
    for b in range(B):
        labels[b, label_lens[b]:] = 0
    return labels, label_lens


def gen_train():
    """Train mode (module.train(), every dropout probability 0): CTC loss (decoder.py:18-23 on top of encoder.py:54-75), d loss / d
    parameter for every parameter, and the BatchNorm running statistics after the step (batch statistics over ALL B*T' positions,
    padded ones included -- quirk Q6)."""
    import decoder as ref_decoder  # noqa: E402
    zero_drop = dict(dropout=0.0, attention_dropout=0.0, pos_enc_dropout=0.0)
    cases = {
        "train_cfg1": (dict(CFG1, **zero_drop), 11, 21, 2, 200, [200, 163], 73, 9, 51, {}),
        "train_cfg1_chunk": (dict(CFG1, **zero_drop), 11, 21, 2, 200, [200, 163], 73, 9, 51,
                             dict(ctor=dict(use_dynamic_chunk_size=True), fw=dict(decoding_chunk_size=4, num_decoding_chunk_size=2))),
        "train_cfg2s": (dict(CFG2, **zero_drop), 12, 22, 3, 120, [120, 97, 64], 5002, 7, 52, {}),
    }
    for name, (cfg, wseed, xseed, B, T, lens, V, Umax, cseed, extra) in cases.items():
        enc = build_encoder(cfg, wseed, **extra.get("ctor", {})).train()
        dec = synth.load_synth_(ref_decoder.CTCDecoder(V, cfg["encoder_dim"], 0.0), cseed).train()
        x = torch.from_numpy(synth.fbank(xseed, B, T))
        y, m = enc(x, torch.tensor(lens, dtype=torch.int32), **extra.get("fw", {}))
        enc_lens = m.squeeze(1).sum(1)
        labels, label_lens = ctc_batch(np.random.RandomState(cseed), B, V, Umax, enc_lens.numpy())
        loss = dec(y, enc_lens, torch.from_numpy(labels), torch.from_numpy(label_lens))
        loss.backward()
        arrays = dict(loss=t2n(loss.reshape(1)), labels=labels, label_lens=label_lens, enc_lens=t2n(enc_lens))
        pack_grad(arrays, "y", y, 7)
        gi = 0
        for pre, mod in (("enc.", enc), ("ctc.", dec)):
            for k, p in mod.named_parameters():
                assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
                pack_grad(arrays, "grad:" + pre + k, p.grad, 1000 + gi)
                gi += 1
        for k, v in enc.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
                arrays["bn:" + k] = t2n(v)
        save(name, arrays, dict(cfg=cfg, wseed=wseed, xseed=xseed, batch=B, frames=T, lens=lens, V=V, Umax=Umax, cseed=cseed,
                                ctor=extra.get("ctor", {}), fw=extra.get("fw", {}), grad_samples=GRAD_SAMPLES))

    # ---- module level: out = m(x); (out * G).sum().backward() -> out, d/dx, d/dparam (D=144 modules of gen_modules) ----
    D, H, FF, K, B, T = 144, 4, 576, 15, 3, 37
    lens = [37, 30, 19]
    pad = ~ref_utils.make_pad_mask(torch.tensor(lens, dtype=torch.int32), T).unsqueeze(1)
    chunk_all = ref_utils.subsequent_chunk_mask(T, 5, 1, torch.device("cpu")).unsqueeze(0).expand(B, T, T)   # no fully-masked row
    chunk_pad = chunk_all & pad                                                                                # has fully-masked rows
    rpe = ref_attention.RelativePositionalEncoding(D, 0.0)
    pos_b = rpe.pe[0:B]
    arrays = {}

    def run(tag, mod, x, call, gseed, pseed0):
        mod.zero_grad()
        xr = x.clone().requires_grad_(True)
        out = call(mod, xr)
        G = torch.from_numpy(synth.normal(gseed, tuple(out.shape)))
        (out * G).sum().backward()
        pack_grad(arrays, tag + ":out", out, pseed0)
        if xr.grad is not None:
            pack_grad(arrays, tag + ":dx", xr.grad, pseed0 + 1)
        for i, (k, p) in enumerate(mod.named_parameters()):
            if p.grad is not None:
                arrays[tag + ":finite:" + k] = np.array([int(torch.isfinite(p.grad).all())], dtype=np.int64)
                pack_grad(arrays, tag + ":grad:" + k, torch.nan_to_num(p.grad, nan=0.0), pseed0 + 2 + i)
        return out

    x = torch.from_numpy(synth.normal(41, (B, T, D)))
    m = synth.load_synth_(ref_feedforward.PositionwiseFeedForwardModule(D, 0.0, FF).train(), 31)
    run("ffn", m, x, lambda mod, xr: mod(xr), 61, 2000)

    x = torch.from_numpy(synth.normal(42, (B, T, D)))
    m = synth.load_synth_(ref_attention.RelativeMultiHeadSelfAttentionModule(D, H, 0.0).train(), 32)
    run("relmhsa_pad", m, x, lambda mod, xr: mod(xr, xr, xr, pad, pos_b)[0], 62, 2100)
    run("relmhsa_chunk", m, x, lambda mod, xr: mod(xr, xr, xr, chunk_all, pos_b)[0], 63, 2200)
    run("relmhsa_chunkpad", m, x, lambda mod, xr: mod(xr, xr, xr, chunk_pad, pos_b)[0], 64, 2300)   # reference: NaN gradients (recorded)
    m = synth.load_synth_(ref_attention.MultiHeadSelfAttentionModule(D, H, 0.0).train(), 36)
    run("mhsa_pad", m, x, lambda mod, xr: mod(xr, xr, xr, pad)[0], 65, 2400)

    x = torch.from_numpy(synth.normal(43, (B, T, D)))
    m = synth.load_synth_(ref_convolution.ConvolutionModule(D, K, FF).train(), 33)
    run("conv_pad", m, x, lambda mod, xr: mod(xr, pad)[0], 66, 2500)
    arrays["conv_pad:running_mean"], arrays["conv_pad:running_var"] = t2n(m.norm.running_mean), t2n(m.norm.running_var)
    arrays["conv_pad:num_batches_tracked"] = t2n(m.norm.num_batches_tracked).reshape(1)

    xf = torch.from_numpy(synth.fbank(44, 3, 83))
    padf = ~ref_utils.make_pad_mask(torch.tensor([83, 60, 7], dtype=torch.int32), 83).unsqueeze(1)
    m = synth.load_synth_(ref_convolution.ConvolutionSubSampling(80, D, ref_attention.RelativePositionalEncoding(D, 0.0)).train(), 34)
    run("sub", m, xf, lambda mod, xr: mod(xr, padf)[0], 67, 2600)

    x = torch.from_numpy(synth.normal(45, (B, T, D)))
    m = synth.load_synth_(ref_encoder_layer.ConformerEncoderLayer(D, K, 0.0, 0.0, FF, H, True).train(), 35)
    run("layer", m, x, lambda mod, xr: mod(xr, pad, pos_b, pad)[0], 68, 2700)
    m = synth.load_synth_(ref_encoder_layer.ConformerEncoderLayer(D, K, 0.0, 0.0, FF, H, False).train(), 37)
    run("layer_norel", m, x, lambda mod, xr: mod(xr, pad, None, pad)[0], 69, 2800)
    save("train_mods_d144", arrays, dict(D=D, H=H, FF=FF, K=K, B=B, T=T, lens=lens, sub_lens=[83, 60, 7], grad_samples=GRAD_SAMPLES,
                                         seeds=dict(ffn=(31, 41, 61), relmhsa=(32, 42, 62, 63, 64), mhsa=(36, 42, 65), conv=(33, 43, 66),
                                                    sub=(34, 44, 67), layer=(35, 45, 68), layer_norel=(37, 45, 69))))


def gen_labels():
    """The host-side label helpers of the reference's utils.py (pad_list, add_blank, add_sos_eos, reverse_sequence, make_subsequent_mask,
    load_vocabs) on integer inputs drawn from numpy seeds: the drop-in utils.py must keep providing them (the file-swap route imports
    `from utils import *` in model.py / decoder.py and `load_vocabs` in executor.py)."""
    import tempfile
    rs = np.random.RandomState(77)
    lens = [7, 4, 1, 5]
    targets = np.full((4, 7), -1, dtype=np.int64)
    for b, n in enumerate(lens):
        targets[b, :n] = rs.randint(2, 50, size=n)
    tt = torch.from_numpy(targets)
    arrays = {"targets": targets, "lens": np.array(lens, dtype=np.int64)}
    arrays["add_blank"] = t2n(ref_utils.add_blank(tt, 0, -1))
    a, b = ref_utils.add_sos_eos(tt, 51, 52, -1)
    arrays["sos_in"], arrays["eos_out"] = t2n(a), t2n(b)
    arrays["reverse"] = t2n(ref_utils.reverse_sequence(tt, torch.tensor(lens), -1))
    arrays["subsequent_6"] = t2n(ref_utils.make_subsequent_mask(6, torch.device("cpu")))
    seqs = [torch.from_numpy(rs.standard_normal(n).astype(np.float32)) for n in (5, 2, 3)]
    arrays["pad_list"] = t2n(ref_utils.pad_list(seqs, -2.5))
    words = ["<blank>", "<unk>", "a", "bc", "<sos/eos>"]
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for i, w in enumerate(words):
            f.write("%s %d\n" % (w, i))
    vocab, n = ref_utils.load_vocabs(f.name)
    os.unlink(f.name)
    save("labels", arrays, {"vocab": vocab, "vocab_size": n, "words": words, "pad_seed": 77})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "labels":       # new fixture only: the others keep the torch RNG stream they were committed with
        gen_labels()
        sys.exit(0)
    gen_masks()
    gen_modules()
    gen_encoders()
    gen_ctc()
    gen_joint()
    gen_train()
    gen_greedy()          # the generators above keep the torch RNG stream they were committed with
    gen_labels()          # numpy seeds only
