"""Whole-utterance forward latency for SMALL batches (config-2 model, bf16, T = 1000): the row chains against the split feed-forward path
(ConformerEncoder.split_small_batches, csrc/ffnsplit.hip).  Usage (GPU box): python scripts/bench_small_batch.py"""
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
    for B in ((1, 2, 4, 8) if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):     # e.g. CFM_FFSPLIT_MAX_ROWS=100000 ... 16 32: the split path at full size
        x = torch.from_numpy(np.random.RandomState(B).standard_normal((B, 1000, 80)).astype(np.float32)).to(dev)
        lens = torch.full((B,), 1000, dtype=torch.int32, device=dev)
        res, outs = [], []
        for split in (False, True):
            enc.split_small_batches = split
            st = torch.cuda.Stream(device=dev)
            with torch.no_grad(), torch.cuda.stream(st):
                for _ in range(3):
                    y, _ = enc(x, lens)
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    y, _ = enc(x, lens)
                for _ in range(60):
                    g.replay()
                st.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    g.replay()
                st.synchronize()
                res.append((time.perf_counter() - t0) / 200 * 1e3)
                outs.append(y.clone())
        d = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
        print("batch %d x 1000 frames (%4d rows): row chains %.3f ms, split feed-forward %.3f ms per forward; max|d|/max|y| between the two %.1e" % (
            B, B * 249, res[0], res[1], d), flush=True)
    enc.split_small_batches = False


if __name__ == "__main__":
    main()
