"""Experiment: config-2 forward steps replayed from S captured graphs on S streams (steps i, i+1, ... in flight together) against one stream.
Each step is still one batch of 32 x (80 x 1000); only the launch ramps and drains of neighbouring steps can overlap.
Measured (MI355X, round 2): 1 stream 1.264 ms per step, 2 streams 1.262 ms, 3 streams 1.223 ms -- nothing to gain: the chained kernels hold
154 KB of LDS on 249 of the 256 CUs, so a neighbouring step's kernels cannot start under them.  Not used by bench.py.
Usage (GPU box): python scripts/exp_two_streams.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    import cfm
    cfm.set_precision("bf16")
    enc = bench.build_encoder(dev)
    B, T = 32, 1000
    for S in (1, 2, 3, 1, 2):
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        xs = [torch.from_numpy(np.random.RandomState(1234 + i).standard_normal((B, T, 80)).astype(np.float32)).to(dev) for i in range(S)]
        lens = torch.full((B,), T, dtype=torch.int32, device=dev)
        graphs, outs = [], []
        with torch.no_grad():
            for s, x in zip(streams, xs):
                with torch.cuda.stream(s):
                    for _ in range(3):
                        enc(x, lens)
                    s.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        y, m = enc(x, lens)
                    g.replay()
                    s.synchronize()
                    graphs.append(g)
                    outs.append(y)
        ref = outs[0].clone()
        for i in range(20):
            with torch.cuda.stream(streams[i % S]):
                graphs[i % S].replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % S]):
                graphs[i % S].replay()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        same = bool(torch.equal(ref, outs[0]))
        print("streams %d: %.4f ms per step = %.2f M frames/s; stream-0 output unchanged: %s" % (S, dt * 1e3, B * T / dt / 1e6, same), flush=True)
        del graphs, outs


if __name__ == "__main__":
    main()
