"""Does a short timed region read slower than a long one because the GPU is still ramping?  Times consecutive windows of 20 graph replays of the
config-2 forward, each window bracketed by device syncs like bench.py's timed region, from a cold start.  Usage (GPU box): python scripts/probe_clock_ramp.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    import cfm
    cfm.set_precision("bf16")
    enc = bench.build_encoder(dev)
    x = torch.from_numpy(np.random.RandomState(1234).standard_normal((32, 1000, 80)).astype(np.float32)).to(dev)
    lens = torch.full((32,), 1000, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    with torch.no_grad(), torch.cuda.stream(st):
        for _ in range(3):
            enc(x, lens)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            enc(x, lens)
        st.synchronize()
        time.sleep(1.0)                                     # idle, as after setup work on the host
        out = []
        for w in range(25):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize(dev)
            out.append((time.perf_counter() - t0) / 20 * 1e3)
        print("ms per step in consecutive windows of 20 replays:", " ".join("%.3f" % v for v in out))


if __name__ == "__main__":
    main()
