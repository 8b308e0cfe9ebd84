"""Which Python lines launch the torch-side glue kernels (copy / add / fill ...) of a training step?

Runs the bench's config-3 job under a TorchDispatchMode that records, for every aten op on device tensors, the innermost frames of this
repo (forward, custom-Function backward and the trainer alike) and prints call counts per optimizer step.
Usage (GPU box): python scripts/trace_glue.py [steps]
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SKIP = ("aten.view", "aten.detach", "aten.as_strided", "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.t.",
        "aten.transpose", "aten.permute", "aten._unsafe_view", "aten.alias", "aten.empty", "aten.reshape", "aten.unbind", "aten.split", "aten.narrow",
        "aten.stride", "aten.size", "aten.is_", "aten.sym_", "aten._local_scalar_dense", "aten.lift_fresh", "aten.new_empty", "aten.empty_like",
        "aten.empty_strided", "aten.unflatten", "aten.flatten")


class Tally(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            frames = [f for f in traceback.extract_stack()[:-1] if "/torch/" not in f.filename and "trace_glue" not in f.filename
                      and "python3" not in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(frames[-3:])) or "(autograd engine)"
            self.counts[(name, where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    enc, dec, tr, mbs, frames = bench.build_train_job(dev, 0, None)
    for i in range(2):
        tr.step([mbs[0], mbs[1]])
    torch.cuda.synchronize()
    tally = Tally()
    with tally:
        for i in range(steps):
            tr.step([mbs[(2 * i) % len(mbs)], mbs[(2 * i + 1) % len(mbs)]])
        torch.cuda.synchronize()
    print("aten ops per optimizer step (2 micro-batches), by innermost repo frames:")
    for (name, where), n in tally.counts.most_common(80):
        print(f"  {n / steps:7.1f}  {name:28s} {where}")


if __name__ == "__main__":
    main()
