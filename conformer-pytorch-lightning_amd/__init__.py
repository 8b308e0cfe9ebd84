"""conformer-pytorch-lightning_amd -- MI355X-native conformer encoder hot path.

The directory name is not a Python identifier (it mirrors the reference repository's name), and the reference resolves
its modules as FLAT names from a directory on ``sys.path`` (``from attention import ...``, encoder.py:3-6).  So this
directory is used the same way: put it on ``sys.path`` and import ``attention``, ``convolution``, ``feedforward``,
``encoder_layer``, ``encoder``, ``utils``, ``cmvn`` and the binding package ``cfm``.  Loading this ``__init__`` (e.g. via
importlib with an alias) does exactly that.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import cfm  # noqa: E402,F401
from encoder import ConformerEncoder  # noqa: E402,F401
from encoder_layer import ConformerEncoderLayer  # noqa: E402,F401
from attention import (MultiHeadSelfAttentionModule, PositionalEncoding, RelativeMultiHeadSelfAttentionModule,  # noqa: E402,F401
                       RelativePositionalEncoding)
from convolution import ConvolutionModule, ConvolutionSubSampling  # noqa: E402,F401
from feedforward import PositionwiseFeedForwardModule  # noqa: E402,F401
