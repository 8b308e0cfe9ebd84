"""CPU: the host-side label helpers the drop-in utils.py re-provides (with this package shadowing the reference's src/ on sys.path,
model.py / decoder.py do `from utils import *` and executor.py `from utils import load_vocabs`).  Expected values: tests/golden/labels.npz,
written by tests/golden/make_golden.py from the reference's own functions (src/utils.py:31-81,163-190).  Integer work: bit-exact."""
import json
import os
import sys

import numpy as np
import torch

from conftest import PKG, ROOT

if PKG not in sys.path:
    sys.path.insert(0, PKG)


def test_label_helpers_equal_the_reference(tmp_path):
    import utils as U
    g = np.load(os.path.join(ROOT, "tests", "golden", "labels.npz"))
    meta = json.loads(bytes(g["meta"]).decode())
    tt = torch.from_numpy(g["targets"])
    lens = torch.from_numpy(g["lens"])
    out = U.add_blank(tt, 0, -1)
    assert out.dtype == torch.long and np.array_equal(out.numpy(), g["add_blank"])
    a, b = U.add_sos_eos(tt, 51, 52, -1)
    assert a.dtype == torch.long and np.array_equal(a.numpy(), g["sos_in"]) and np.array_equal(b.numpy(), g["eos_out"])
    r = U.reverse_sequence(tt, lens, -1)
    assert r.dtype == torch.int32 and np.array_equal(r.numpy(), g["reverse"])
    m = U.make_subsequent_mask(6, torch.device("cpu"))
    assert m.dtype == torch.bool and np.array_equal(m.numpy().astype(np.uint8), g["subsequent_6"])
    rs = np.random.RandomState(meta["pad_seed"])
    for n in (7, 4, 1, 5):                                     # the generator's draws, in its order
        rs.randint(2, 50, size=n)
    seqs = [torch.from_numpy(rs.standard_normal(n).astype(np.float32)) for n in (5, 2, 3)]
    p = U.pad_list(seqs, -2.5)
    assert p.dtype == torch.float32 and np.array_equal(p.numpy(), g["pad_list"])
    path = tmp_path / "vocab.txt"
    path.write_text("".join("%s %d\n" % (w, i) for i, w in enumerate(meta["words"])))
    table, n = U.load_vocabs(str(path))
    assert table == meta["vocab"] and n == meta["vocab_size"]


def test_drop_in_utils_exports_every_public_name_of_the_reference():
    """The names the reference's `from utils import *` call sites use (model.py, decoder.py, encoder.py, executor.py)."""
    import utils as U
    for name in ("load_cmvn", "pad_list", "load_vocabs", "add_blank", "make_pad_mask", "subsequent_chunk_mask", "make_attn_mask",
                 "make_subsequent_mask", "add_sos_eos", "reverse_sequence"):
        assert callable(getattr(U, name)), name
