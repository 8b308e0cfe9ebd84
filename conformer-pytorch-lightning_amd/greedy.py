"""RNN-T greedy search for B streams at once, resident on the device -- the reference's `Transducer.basic_greedy_search`
(src/model.py:215-269) without its host loop.

The reference decodes ONE utterance with a Python `while t < T'` loop: per iteration one LSTM step, one joint evaluation, an argmax and
TWO host synchronisations (`joint_out_max != self.blank`, `.item()`), i.e. it is bound by launch + sync latency ("host-bound").
Here the per-stream control state -- frame index t, the symbols emitted on the current frame, the predictor's input token and LSTM state,
the hypothesis buffer and its length -- lives in device tensors and one decoding step is a fixed sequence of device operations for all B
streams, with the reference's branches turned into selects:

    pred = projection(LSTM(embed(token), state))          the reference recomputes it only after a non-blank; it is a pure function
                                                            of (token, state), which change only then -- recomputing gives the same value
    k    = argmax(ffn_out(tanh(enc_ffn(enc[t]) + pred_ffn(pred))))      (log_softmax dropped: monotone)
    non-blank:  append k, token <- k, state <- the LSTM's new state, per-frame count + 1
    blank, or per-frame count == n_steps:  t + 1, per-frame count <- 0      (model.py:262-267, including the quirk that a frame left because of
                                                                            the cap keeps "previous output was non-blank")
    a stream with t == T'_b is finished and no longer changes.

`steps_per_replay` such steps are captured in ONE HIP graph; the host replays it and looks at a single "all finished" flag per replay -- one
synchronisation per `steps_per_replay` steps instead of two per step, and B streams share every weight read.  `enc_ffn` is applied to all
frames once (model.py:250 does it per step).  Everything runs in float32 torch operations on the reference's own parameters (the joint
and the predictor are tiny here: one row per stream); the result is the reference's token sequence per stream -- tests compare with the
oracle's restatement of the loop and with tokens produced by running the reference's predictor / joint modules.
"""
import torch
import torch.nn.functional as F


class BatchedGreedySearch:

    def __init__(self, predictor, joint, blank=0, n_steps=64, steps_per_replay=32, use_graph=True):
        self.predictor, self.joint = predictor, joint
        self.blank, self.n_steps, self.steps_per_replay, self.use_graph = int(blank), int(n_steps), int(steps_per_replay), bool(use_graph)
        self._graph = None
        self._key = None

    # -- one LSTM step on (B, E) inputs with nn.LSTM's parameters (eval mode: no inter-layer dropout); torch.nn.LSTM restated for T = 1
    def _lstm_step(self, x, h, c):
        rnn = self.predictor.rnn
        hs, cs = [], []
        for l in range(rnn.num_layers):
            w_ih, w_hh = getattr(rnn, "weight_ih_l%d" % l), getattr(rnn, "weight_hh_l%d" % l)
            b_ih, b_hh = (getattr(rnn, "bias_ih_l%d" % l), getattr(rnn, "bias_hh_l%d" % l)) if rnn.bias else (None, None)
            gates = F.linear(x, w_ih, b_ih) + F.linear(h[l], w_hh, b_hh)
            i, f, g, o = gates.chunk(4, dim=-1)
            c1 = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(g)
            x = torch.sigmoid(o) * torch.tanh(c1)
            hs.append(x)
            cs.append(c1)
        return x, torch.stack(hs), torch.stack(cs)

    def _step(self, S):
        ar = S["ar"]
        e = S["enc_proj"][ar, torch.minimum(S["t"], S["tmax"])]                       # (B, J): the frame each stream is on
        y, h1, c1 = self._lstm_step(self.predictor.embed(S["token"]), S["h"], S["c"])
        pred = self.predictor.projection(y)
        z = self.joint.ffn_out(torch.tanh(e + self.joint.pred_ffn(pred)))
        k = z.argmax(dim=-1)
        live = ~S["done"]
        nb = (k != self.blank) & live
        pos = torch.minimum(S["count"], S["cap"])
        S["hyps"][ar, pos] = torch.where(nb, k, S["hyps"][ar, pos])
        S["count"].add_(nb.to(torch.int64))
        S["token"].copy_(torch.where(nb, k, S["token"]))
        S["h"].copy_(torch.where(nb[None, :, None], h1, S["h"]))
        S["c"].copy_(torch.where(nb[None, :, None], c1, S["c"]))
        S["frame_count"].add_(nb.to(torch.int64))
        adv = ((k == self.blank) | (S["frame_count"] >= self.n_steps)) & live
        S["t"].add_(adv.to(torch.int64))
        S["frame_count"].mul_((~adv).to(torch.int64))
        S["done"].copy_(S["t"] >= S["lens"])

    def _state(self, B, T, dev):
        L, H = self.predictor.num_layers, self.predictor.hidden_size
        J = self.joint.enc_ffn.out_features
        cap = T * self.n_steps                                                          # the most a stream can emit
        z64 = lambda *s: torch.zeros(s, dtype=torch.int64, device=dev)
        return dict(ar=torch.arange(B, device=dev), enc_proj=torch.zeros((B, T, J), device=dev), t=z64(B), tmax=torch.full((B,), T - 1, dtype=torch.int64, device=dev),
                    lens=z64(B), token=z64(B), h=torch.zeros((L, B, H), device=dev), c=torch.zeros((L, B, H), device=dev), count=z64(B), frame_count=z64(B),
                    hyps=z64(B, cap + 1), cap=torch.full((B,), cap, dtype=torch.int64, device=dev), done=torch.zeros((B,), dtype=torch.bool, device=dev),
                    all_done=torch.zeros((), dtype=torch.bool, device=dev))

    @torch.no_grad()
    def search(self, enc_out, enc_lens, token=None, state=None):
        """enc_out (B, T', E) float32 encoder output, enc_lens (B,) valid frames per stream.  token (B,) / state (h, c): the predictor's input
        and LSTM state to start from (a continued stream, model.py:186-192); default: blank and zeros.  Returns (list of B token lists,
        (token, (h, c)) to continue with)."""
        if self.predictor.training or self.joint.training:
            raise RuntimeError("greedy search is an eval-mode operation (the predictor's dropout would be live)")
        B, T, _ = enc_out.shape
        dev = enc_out.device
        key = (B, T, str(dev))
        if self._key != key:
            self._S, self._graph, self._key = self._state(B, T, dev), None, key
        S = self._S
        S["enc_proj"].copy_(self.joint.enc_ffn(enc_out.float()))
        S["lens"].copy_(torch.as_tensor(enc_lens, device=dev).to(torch.int64).clamp(0, T))
        for k in ("t", "count", "frame_count", "hyps"):
            S[k].zero_()
        S["token"].fill_(self.blank) if token is None else S["token"].copy_(token.reshape(B))
        if state is None:
            S["h"].zero_(); S["c"].zero_()
        else:
            S["h"].copy_(state[0]); S["c"].copy_(state[1])
        S["done"].copy_(S["t"] >= S["lens"])

        def run_chunk():
            for _ in range(self.steps_per_replay):
                self._step(S)
            S["all_done"].copy_(S["done"].all())

        if self.use_graph and dev.type == "cuda" and self._graph is None:
            snap = {k: S[k].clone() for k in ("t", "count", "frame_count", "hyps", "token", "h", "c", "done")}
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                run_chunk()                                                             # warm-up (allocator, lazy init) outside the capture
            torch.cuda.current_stream(dev).wait_stream(side)
            for k, v in snap.items():
                S[k].copy_(v)                                                           # ... and undone
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                run_chunk()
            for k, v in snap.items():
                S[k].copy_(v)                                                           # (capture does not execute, but keep the state explicit)
        limit = (T * (self.n_steps + 1)) // self.steps_per_replay + 2                   # every step emits or advances: at most T (n_steps + 1) of them
        for _ in range(limit):
            if self._graph is not None:
                self._graph.replay()
            else:
                run_chunk()
            if bool(S["all_done"]):                                                     # the one host synchronisation per replay
                break
        else:
            raise RuntimeError("greedy search did not finish within its step bound")
        counts = S["count"].tolist()
        hyps = S["hyps"].cpu()
        return [hyps[b, :counts[b]].tolist() for b in range(B)], (S["token"].clone(), (S["h"].clone(), S["c"].clone()))
