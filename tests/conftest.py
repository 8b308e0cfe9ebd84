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


@pytest.fixture(scope="session")
def golden():
    return load_golden
