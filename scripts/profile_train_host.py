"""cProfile of the host side of the config-3 training step (where do the ~14 ms of enqueue time go?).  Usage (GPU box): python scripts/profile_train_host.py [steps]"""
import cProfile
import os
import pstats
import sys

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
    pr = cProfile.Profile()
    pr.enable()
    for i in range(steps):
        tr.step([mbs[(2 * i) % len(mbs)], mbs[(2 * i + 1) % len(mbs)]])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
