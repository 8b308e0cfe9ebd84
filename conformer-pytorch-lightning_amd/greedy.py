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

    def __init__(self, predictor, joint, blank=0, n_steps=64, steps_per_replay=32, use_graph=True, fused=None):
        """fused: one step = six HIP launches on f32 weights (csrc/greedy.hip, include/cfm.h cfm_greedy_step) instead of ~45 torch operations;
        default: on an MI355X when the sizes fit (B <= 64 streams, dimensions multiples of 16), the torch-operation form otherwise (and on CPU)."""
        self.predictor, self.joint = predictor, joint
        self.blank, self.n_steps, self.steps_per_replay, self.use_graph = int(blank), int(n_steps), int(steps_per_replay), bool(use_graph)
        self.fused = fused
        self._graph = None
        self._key = None
        self._key_f = None
        self._fw = None

    # -- fused step -----------------------------------------------------------------------------------------------------------------------
    def _fused_ok(self, B, dev):
        pr, jn = self.predictor, self.joint
        dims = (pr.embed_size, pr.hidden_size, pr.projection.out_features, jn.pred_ffn.out_features)
        return (dev.type == "cuda" and B <= 64 and pr.num_layers <= 4 and all(d % 16 == 0 for d in dims) and pr.rnn.bias and
                jn.pred_ffn.in_features == pr.projection.out_features)

    def _fused_weights(self, dev):
        """f32 packs for cfm_greedy_step, rebuilt when a source parameter changes: [W_ih | W_hh] per layer with rows ordered [unit][gate],
        b_ih + b_hh likewise, the vocabulary projection padded to a multiple of 16 rows (bias -inf there: never the argmax)."""
        pr, jn = self.predictor, self.joint
        srcs = list(pr.parameters()) + list(jn.parameters())
        key = tuple((t.data_ptr(), t._version) for t in srcs)
        if self._fw is not None and self._fw[0] == key:
            return self._fw[1]
        H = pr.hidden_size
        perm = (torch.arange(H, device=dev)[:, None] + H * torch.arange(4, device=dev)[None, :]).reshape(-1)     # row u*4 + gate <- gate*H + u
        W = {"embed": pr.embed.weight.detach().float().contiguous(), "lstm_w": [], "lstm_b": []}
        for l in range(pr.num_layers):
            w = torch.cat([getattr(pr.rnn, "weight_ih_l%d" % l).detach().float(), getattr(pr.rnn, "weight_hh_l%d" % l).detach().float()], 1)
            b = getattr(pr.rnn, "bias_ih_l%d" % l).detach().float() + getattr(pr.rnn, "bias_hh_l%d" % l).detach().float()
            W["lstm_w"].append(w[perm].contiguous())
            W["lstm_b"].append(b[perm].contiguous())
        V = jn.ffn_out.out_features
        Vp = (V + 15) // 16 * 16
        ow = torch.zeros((Vp, jn.ffn_out.in_features), device=dev)
        ow[:V] = jn.ffn_out.weight.detach().float()
        ob = torch.full((Vp,), float("-inf"), device=dev)
        ob[:V] = jn.ffn_out.bias.detach().float()
        W.update(proj_w=pr.projection.weight.detach().float().contiguous(), proj_b=pr.projection.bias.detach().float().contiguous(),
                 pf_w=jn.pred_ffn.weight.detach().float().contiguous(), pf_b=jn.pred_ffn.bias.detach().float().contiguous(), out_w=ow, out_b=ob, Vp=Vp)
        self._fw = (key, W)
        return W

    def _fused_desc(self, S, W, B, T):
        import ctypes
        import cfm
        pr, jn = self.predictor, self.joint
        L, H = pr.num_layers, pr.hidden_size
        dev = S["t"].device
        for k, shape, dt in (("h_new", (L, B, H), torch.float32), ("c_new", (L, B, H), torch.float32), ("pred", (B, pr.projection.out_features), torch.float32),
                             ("act", (B, jn.pred_ffn.out_features), torch.float32), ("pmax", (W["Vp"] // 16, B), torch.float32),
                             ("pidx", (W["Vp"] // 16, B), torch.int32), ("done8", (B,), torch.uint8), ("n_done", (1,), torch.int32)):
            if k not in S:
                S[k] = torch.zeros(shape, dtype=dt, device=dev)
        d = cfm.GreedyDesc()
        d.embed = W["embed"].data_ptr()
        for l in range(L):
            d.lstm_w[l], d.lstm_b[l] = W["lstm_w"][l].data_ptr(), W["lstm_b"][l].data_ptr()
        for k in ("proj_w", "proj_b", "pf_w", "pf_b", "out_w", "out_b"):
            setattr(d, k, W[k].data_ptr())
        d.enc_proj = S["enc_proj"].data_ptr()
        for k in ("token", "t", "count", "frame_count", "hyps", "lens", "h", "c", "h_new", "c_new", "pred", "act", "pmax", "pidx", "n_done"):
            setattr(d, k, S[k].data_ptr())
        d.done = S["done8"].data_ptr()
        d.hyp_cap, d.hyp_ld = int(S["cap"][0]), S["hyps"].shape[1]
        d.B, d.T, d.L, d.E, d.H, d.P, d.J, d.Vp = B, T, L, pr.embed_size, H, pr.projection.out_features, jn.pred_ffn.out_features, W["Vp"]
        d.blank, d.n_steps = self.blank, self.n_steps
        return d

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
            self._S, self._graph, self._key, self._key_f = self._state(B, T, dev), None, key, None
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
        fused = self._fused_ok(B, dev) if self.fused is None else bool(self.fused)
        if fused:
            import ctypes
            import cfm
            if not self._fused_ok(B, dev):
                raise RuntimeError("the fused greedy step needs a GPU, B <= 64 streams and dimensions that are multiples of 16")
            W = self._fused_weights(dev)
            if self._key_f != (key, id(W)):
                self._desc, self._graph, self._key_f = self._fused_desc(S, W, B, T), None, (key, id(W))
            S["done8"].copy_(S["done"].to(torch.uint8))
            S["n_done"].copy_(S["done"].sum().to(torch.int32).reshape(1))
            desc, lib = self._desc, cfm.lib()

        def run_chunk():
            if fused:
                for _ in range(self.steps_per_replay):
                    cfm.check(lib.cfm_greedy_step(ctypes.byref(desc), cfm.stream()), "cfm_greedy_step")
                S["all_done"].copy_((S["n_done"] >= B).reshape(()))
                return
            for _ in range(self.steps_per_replay):
                self._step(S)
            S["all_done"].copy_(S["done"].all())

        if self.use_graph and dev.type == "cuda" and self._graph is None:
            snap = {k: S[k].clone() for k in ("t", "count", "frame_count", "hyps", "token", "h", "c", "done") + (("done8", "n_done") if fused else ())}
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
