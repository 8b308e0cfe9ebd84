"""RNN-T greedy search: the reference's basic_greedy_search (model.py:215-269).

tests/golden/greedy.npz holds token sequences produced with the REFERENCE's RNNPredictor.forward_step and TransducerJoint.forward doing
every step (the loop around them restated in the generator: model.py does not import without torchaudio).  Checked against them:
  * the oracle's restatement of loop + predictor step + joint (CPU, `not gpu`);
  * the drop-in RNNPredictor (predictor.py) driven by the same loop (CPU);
  * greedy.BatchedGreedySearch -- all utterances of a case as ONE batch, control state in tensors, no per-step host decisions -- eagerly
    on the CPU here, and on the GPU with the steps captured in a HIP graph (`-m gpu`), including ragged lengths, the per-frame cap
    (n_steps 1, 3, 4) and a search continued from a carried (token, LSTM state).
Token sequences are integers: the comparison is exact."""
import numpy as np
import pytest
import torch

import synth
from conftest import load_golden


def _case_modules(c):
    import joint
    import predictor
    pr = predictor.RNNPredictor(c["V"], c["embed"], c["P"], c["hidden"], 0.1, c["layers"]).eval()
    jn = joint.TransducerJoint(c["V"], c["E"], c["P"], c["J"]).eval()
    synth.load_synth_(pr, c["seed"])
    synth.load_synth_(jn, c["seed"] + 1)
    synth.greedy_joint_(jn, c["V"])
    return pr, jn


def _enc(c, u):
    return torch.from_numpy(synth.normal(c["seed"] + 10 + u, (1, c["T"], c["E"]), 1.0))


def _cases():
    g, meta = load_golden("greedy")
    return g, meta["cases"]


def test_oracle_greedy_search_matches_reference_tokens():
    from oracle import conformer_oracle as O
    g, cases = _cases()
    for c in cases:
        pr, jn = _case_modules(c)
        P = {"p." + k: v.detach() for k, v in pr.state_dict().items()}
        P.update({"j." + k: v.detach() for k, v in jn.state_dict().items()})
        for u, n in enumerate(c["lens"]):
            enc = _enc(c, u)[0]
            hyps, _ = O.rnnt_greedy_search(P, "p.", "j.", enc, n, n_steps=c["n_steps"])
            assert hyps == g["%s_utt%d" % (c["name"], u)].tolist(), (c["name"], u)
        enc = _enc(c, 0)[0]
        half = c["T"] // 2
        a, (tok, st) = O.rnnt_greedy_search(P, "p.", "j.", enc[:half], half, n_steps=c["n_steps"])
        b, _ = O.rnnt_greedy_search(P, "p.", "j.", enc[half:], c["T"] - half, n_steps=c["n_steps"], token=tok, state=st)
        assert a == g[c["name"] + "_utt0_first"].tolist() and b == g[c["name"] + "_utt0_second"].tolist()


def test_predictor_module_state_dict_and_step_semantics():
    """Parameter names / shapes of the reference's RNNPredictor, init_state, and forward_step's padding (1 = keep the old state)."""
    import predictor
    pr = predictor.RNNPredictor(73, 48, 96, 80, 0.1, 2).eval()
    names = sorted(pr.state_dict())
    assert names == sorted(["embed.weight", "projection.weight", "projection.bias"] +
                           ["rnn.%s_%s_l%d" % (a, b, l) for a in ("weight", "bias") for b in ("ih", "hh") for l in (0, 1)])
    synth.load_synth_(pr, 7)
    tok = torch.tensor([[5], [9]])
    h0, c0 = pr.init_state(tok)
    assert h0.shape == (2, 2, 80) and float(h0.abs().max()) == 0.0
    with torch.no_grad():
        out, (h1, c1) = pr.forward_step(tok, torch.zeros(2, 1), (h0, c0))
        out2, (h2, c2) = pr.forward_step(tok, torch.tensor([[0.0], [1.0]]), (h1, c1))
        full = pr(torch.tensor([[5, 5], [9, 9]]))
    assert out.shape == (2, 1, 96) and torch.allclose(full[:, 0], out[:, 0], atol=1e-6) and torch.allclose(full[:, 1], out2[:, 0], atol=1e-6)
    assert torch.equal(h2[:, 1], h1[:, 1]) and torch.equal(c2[:, 1], c1[:, 1]) and not torch.equal(h2[:, 0], h1[:, 0])


def _batched(c, g, device, use_graph, fused=False):
    import greedy
    pr, jn = _case_modules(c)
    pr, jn = pr.to(device), jn.to(device)
    enc = torch.cat([_enc(c, u) for u in range(3)], 0).to(device)
    gs = greedy.BatchedGreedySearch(pr, jn, blank=0, n_steps=c["n_steps"], steps_per_replay=8, use_graph=use_graph, fused=fused)
    hyps, _ = gs.search(enc, c["lens"])
    for u in range(3):
        assert hyps[u] == g["%s_utt%d" % (c["name"], u)].tolist(), (c["name"], u, len(hyps[u]))
    # again through the same object (captured graph reused), then a continued search: first halves, then the rest from the carried state
    hyps2, _ = gs.search(enc, c["lens"])
    assert hyps2 == hyps
    half = c["T"] // 2
    first, (tok, st) = gs_half(greedy, pr, jn, c, enc[:1, :half], [half], use_graph, fused=fused)
    second, _ = gs_half(greedy, pr, jn, c, enc[:1, half:], [c["T"] - half], use_graph, tok, st, fused=fused)
    assert first[0] == g[c["name"] + "_utt0_first"].tolist() and second[0] == g[c["name"] + "_utt0_second"].tolist()


def gs_half(greedy, pr, jn, c, enc, lens, use_graph, tok=None, st=None, fused=False):
    gs = greedy.BatchedGreedySearch(pr, jn, blank=0, n_steps=c["n_steps"], steps_per_replay=8, use_graph=use_graph, fused=fused)
    return gs.search(enc.contiguous(), lens, token=tok, state=st)


def test_batched_search_on_cpu_matches_reference_tokens():
    g, cases = _cases()
    for c in cases:
        if c["V"] > 1000:
            continue                                   # the 5002-class case runs on the GPU
        _batched(c, g, torch.device("cpu"), False)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("use_graph", [True, False])
def test_batched_search_on_gpu_matches_reference_tokens(use_graph, fused):
    """fused: one step = six launches of csrc/greedy.hip (f32 MFMA products, LSTM cell / tanh / argmax epilogues, control kernel) instead of
    the torch-operation form; both must reproduce the reference-module tokens."""
    g, cases = _cases()
    for c in cases:
        _batched(c, g, torch.device("cuda"), use_graph, fused)


@pytest.mark.gpu
def test_fused_search_40_streams_matches_torch_form():
    """More than one 16-stream tile (B = 40, ragged lengths, some streams empty): the fused step against the torch-operation form."""
    import greedy
    g, cases = _cases()
    c = [x for x in cases if x["name"] == "small"][0]
    pr, jn = _case_modules(c)
    pr, jn = pr.cuda(), jn.cuda()
    B, T = 40, c["T"]
    enc = torch.cat([torch.from_numpy(synth.normal(700 + b, (1, T, c["E"]), 1.0)) for b in range(B)]).cuda()
    lens = [(7 * b) % (T + 1) for b in range(B)]
    res = []
    for fused in (True, False):
        gs = greedy.BatchedGreedySearch(pr, jn, blank=0, n_steps=c["n_steps"], steps_per_replay=16, use_graph=True, fused=fused)
        res.append(gs.search(enc, lens)[0])
    assert res[0] == res[1] and sum(len(h) for h in res[0]) > 50 and res[0][0] == []
