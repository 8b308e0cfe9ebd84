"""pytest wiring: markers, import paths, shared fixture loaders."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "conformer-pytorch-lightning_amd")
for p in (ROOT, os.path.join(ROOT, "tests"), PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    arrays = {k: z[k] for k in z.files if k != "meta"}
    meta = json.loads(bytes(z["meta"]).decode())
    return arrays, meta


def joint_case_inputs(c):
    """Encoder / predictor outputs of a tests/golden/joint.npz case (tests/golden/make_golden.py gen_joint), as torch f32 tensors."""
    import synth
    import torch
    enc = torch.from_numpy(synth.normal(c["eseed"], (c["B"], c["T"], c["E"]), 1.0))
    pred = torch.from_numpy(synth.normal(c["pseed"], (c["B"], c["U"], c["P"]), 1.0))
    return enc, pred


def check_joint_case(g, c, out):
    """max|d| / max|ref| of a (B,T,U,V) joint output against what the fixture holds for the case: the full tensor, or 512 sampled
    logits plus the per-(b,t,u) sums over the vocabulary."""
    n = c["name"]
    out = np.asarray(out, dtype=np.float64)
    assert out.shape == (c["B"], c["T"], c["U"], c["V"]), out.shape
    if n + "_out" in g:
        ref = g[n + "_out"].astype(np.float64)
        return float(np.abs(out - ref).max() / np.abs(ref).max())
    scale = float(np.abs(g[n + "_vals"]).max())
    e = float(np.abs(out.reshape(-1)[g[n + "_idx"]] - g[n + "_vals"]).max() / scale)
    es = float(np.abs(out.sum(-1) - g[n + "_rowsum"]).max() / float(g[n + "_rowabs"].max()))
    return max(e, es)


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ---- gradient fixtures (tests/golden/make_golden.py pack_grad): full tensors, or sampled entries + float64 statistics ----
def grad_ref_max(g, key):
    """max|ref| of a packed tensor."""
    if key in g:
        return float(np.abs(g[key]).max()) if g[key].size else 0.0
    return float(g[key + "__stats"][3])


def grad_err(g, key, got, floor=0.0):
    """max|got - ref| / max(max|ref|, floor) over what the fixture holds for `key`: every entry of a small tensor, or the sampled
    entries of a big one together with its sum and sum of squares (scaled by the sum of |ref| / of ref^2)."""
    a = np.asarray(got, dtype=np.float64).reshape(-1)
    scale = max(grad_ref_max(g, key), floor, 1e-30)
    if key in g:
        ref = g[key].astype(np.float64).reshape(-1)
        assert a.size == ref.size, (key, a.size, ref.size)
        return float(np.abs(a - ref).max() / scale)
    idx, vals, st = g[key + "__idx"], g[key + "__vals"].astype(np.float64), g[key + "__stats"]
    e = float(np.abs(a[idx] - vals).max() / scale)
    e_sum = abs(a.sum() - st[0]) / max(st[1], floor * a.size, 1e-30)
    e_sq = abs((a * a).sum() - st[2]) / max(st[2], (floor ** 2) * a.size, 1e-30)
    return max(e, float(e_sum), float(e_sq) / 2)       # d(x^2)/x^2 = 2 dx/x
