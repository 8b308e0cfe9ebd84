"""RNN-T prediction network -- drop-in for `RNNPredictor` of the reference's src/predictor.py:14-86.

Embedding -> Dropout -> nn.LSTM (batch_first) -> Linear, with the reference's constructor arguments, parameter names (`embed`, `rnn`,
`projection`: its state_dicts load unchanged) and the three entry points the transducer uses: `forward(inputs, states=None)` for the
teacher-forced (B, U) label matrix of the loss (model.py:100), `init_state(inputs)` and `forward_step(inputs, padding, cache)` for the
decoder's one-symbol step (model.py:245-248), whose `padding` (1 = keep the old state for that item) freezes a stream's LSTM state.

The LSTM itself stays a stock torch module (MIOpen on ROCm): SURVEY.md 2 row 10 scopes it out as a kernel -- two layers of 256 units on one
symbol are four 1024 x 256 matrix-vector products.  What this repository adds on top is `greedy.BatchedGreedySearch`: the reference's
host-driven, batch-1 `while t < T'` loop (model.py:215-269) as a batched, device-resident one.
"""
import torch
import torch.nn as nn


class RNNPredictor(nn.Module):

    def __init__(self, vocab_size, embed_size, output_size, hidden_size, embed_dropout, num_layers, bias=True, dropout=0.1):
        super().__init__()
        self.num_layers, self.hidden_size, self.embed_size = num_layers, hidden_size, embed_size
        self.embed = nn.Embedding(vocab_size, embed_size)
        self.dropout = nn.Dropout(embed_dropout)
        self.rnn = nn.LSTM(input_size=embed_size, hidden_size=hidden_size, num_layers=num_layers, bias=bias, batch_first=True, dropout=dropout)
        self.projection = nn.Linear(hidden_size, output_size)

    def init_state(self, inputs):
        """[h0, c0], zeros of shape (num_layers, B, hidden) on the inputs' device (predictor.py:40-54)."""
        shape = (self.num_layers, inputs.size(0), self.hidden_size)
        return [torch.zeros(shape, device=inputs.device), torch.zeros(shape, device=inputs.device)]

    def _embed(self, inputs):
        return self.dropout(self.embed(inputs))

    def forward(self, inputs, states=None):
        x = self._embed(inputs)
        if states is None:
            h0, c0 = self.init_state(inputs)
            states = (h0.to(x.dtype), c0.to(x.dtype))
        y, _ = self.rnn(x, states)
        return self.projection(y)

    def forward_step(self, inputs, padding, cache):
        """One symbol per item: inputs (B, 1) token ids, cache (h, c) -> (projection output (B, 1, P), (h', c')); where padding (B, 1) is 1
        the item keeps its old state (predictor.py:76-86: new * (1 - padding) + old * padding)."""
        h, c = cache
        y, (h1, c1) = self.rnn(self._embed(inputs), (h, c))
        keep = padding.unsqueeze(0)
        return self.projection(y), (keep * h + (1 - keep) * h1, keep * c + (1 - keep) * c1)
