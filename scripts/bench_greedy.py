"""RNN-T greedy search at the BASELINE config-4 head (V = 5002, join 512, 2 x 256 LSTM), T' = 249 frames: the reference-style host loop
(one utterance, two host synchronisations per step: src/model.py:215-269 driven with this repository's modules) against
greedy.BatchedGreedySearch (B streams, control state on the device, steps captured in a HIP graph).  Usage: [B]"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "conformer-pytorch-lightning_amd"))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import greedy  # noqa: E402
import joint  # noqa: E402
import predictor  # noqa: E402
import synth  # noqa: E402


@torch.no_grad()
def host_loop(pr, jn, enc, n_steps, blank=0):
    padding = torch.zeros(1, 1, device=enc.device)
    tok = torch.tensor([[blank]], device=enc.device)
    cache = pr.init_state(tok)
    e = jn.enc_ffn(enc)
    t, hyps, prev, per_frame, pred, new_cache = 0, [], True, 0, None, None
    while t < enc.size(1):
        if prev:
            out, new_cache = pr.forward_step(tok, padding, cache)
            pred = jn.pred_ffn(out)
        k = jn.ffn_out(torch.tanh(e[:, t:t + 1] + pred)).argmax(dim=-1).squeeze()
        if k != blank:                                   # host sync
            hyps.append(k.item())                        # host sync
            prev, per_frame, tok, cache = True, per_frame + 1, k.reshape(1, 1), new_cache
        if k == blank or per_frame >= n_steps:
            if k == blank:
                prev = False
            t += 1
            per_frame = 0
    return hyps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = "cuda"
    V, E, Pd, J, emb, hid, layers, T, n_steps = 5002, 512, 512, 512, 256, 256, 2, 249, 4
    pr = predictor.RNNPredictor(V, emb, Pd, hid, 0.1, layers).eval()
    jn = joint.TransducerJoint(V, E, Pd, J).eval()
    synth.load_synth_(pr, 53)
    synth.load_synth_(jn, 54)
    synth.greedy_joint_(jn, V)
    pr, jn = pr.to(dev), jn.to(dev)
    enc = torch.cat([torch.from_numpy(synth.normal(500 + b, (1, T, E), 1.0)) for b in range(B)]).to(dev)
    lens = [T] * B
    ref = host_loop(pr, jn, enc[:1], n_steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = host_loop(pr, jn, enc[:1], n_steps)
    torch.cuda.synchronize()
    t_host = time.perf_counter() - t0
    steps = T + len(ref)
    print("host loop, 1 utterance: %d frames, %d symbols, %d steps: %.1f ms = %.0f us per step" % (T, len(ref), steps, t_host * 1e3, t_host / steps * 1e6))
    for graph, fused in ((False, False), (True, False), (True, True)):
        gs = greedy.BatchedGreedySearch(pr, jn, n_steps=n_steps, steps_per_replay=32, use_graph=graph, fused=fused)
        hyps, _ = gs.search(enc, lens)
        assert hyps[0] == ref, "stream 0 differs from the host loop"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            hyps, _ = gs.search(enc, lens)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        nsteps = max(T + len(h) for h in hyps)
        print("batched, %2d streams, %s: %.1f ms per batch = %.2f ms per utterance (%d steps, %.0f us per step); x%.1f the host loop's utterances/s" % (
            B, ("HIP graph of 32 steps" if graph else "eager launches   ") + (", fused HIP step" if fused else ", torch operations"), dt * 1e3, dt * 1e3 / B, nsteps, dt / nsteps * 1e6, t_host / (dt / B)))


if __name__ == "__main__":
    main()
