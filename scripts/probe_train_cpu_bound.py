"""Is the config-3 training step bound by the host (Python + launch calls) or by the device?  Times the enqueue side of K steps (no sync inside)
against the synchronised wall time.  Usage (GPU box): python scripts/probe_train_cpu_bound.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda:0")
    enc, dec, tr, mbs, frames = bench.build_train_job(dev, 0, None)
    for i in range(3):
        tr.step([mbs[0], mbs[1]])
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step([mbs[(2 * i) % len(mbs)], mbs[(2 * i + 1) % len(mbs)]])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("host enqueue %.2f ms per step, device drained %.2f ms later in total, wall %.2f ms per step" % ((t1 - t0) / steps * 1e3, (t2 - t1) * 1e3, (t2 - t0) / steps * 1e3),
              flush=True)


if __name__ == "__main__":
    main()
