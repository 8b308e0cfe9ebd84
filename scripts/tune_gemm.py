#!/usr/bin/env python3
"""Time every GEMM shape of the config-2 encoder with every tile id (GPU box only).  Prints a table; used to set the
tile heuristics in csrc/gemm.hip.  Not part of the test suite."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch  # noqa: E402
import cfm  # noqa: E402

SHAPES = [("ffn1  silu", 7968, 2048, 256, cfm.ACT_SILU, False), ("ffn2  +res", 7968, 256, 2048, 0, True),
          ("qkv", 7968, 768, 256, 0, False), ("out   +res", 7968, 256, 256, 0, True), ("pw1   glu", 7968, 512, 256, cfm.ACT_GLU, False),
          ("front lin", 7968, 256, 4864, 0, False)]


def time_it(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 4, 5, 6]
    print("%-12s %6s %6s %6s | " % ("shape", "M", "N", "K") + " ".join("tile%-2d us (TF/s)   " % t for t in tiles))
    for name, M, N, K, act, res in SHAPES:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        bias = torch.randn(N, device="cuda")
        cols = N // 2 if act == cfm.ACT_GLU else N
        r = torch.randn(M, cols, device="cuda") if res else None
        out = torch.empty(M, cols, device="cuda", dtype=torch.float32 if res else torch.bfloat16)
        cells = []
        for t in tiles:
            try:
                us = time_it(lambda: cfm.gemm(a, w, bias=bias, act=act, residual=r, alpha=0.5, out=out, tile=t))
                cells.append("%7.1f (%6.1f)    " % (us, 2.0 * M * N * K / us / 1e6))
            except RuntimeError as ex:
                cells.append("   n/a             ")
        print("%-12s %6d %6d %6d | " % (name, M, N, K) + " ".join(cells), flush=True)
    # conv2 implicit GEMM
    B, T1, F1, C = 32, 499, 39, 256
    T2, F2 = 249, 19
    img = torch.randn(B, T1, F1, C, device="cuda").bfloat16()
    w = (torch.randn(C, 9 * C, device="cuda") * (9 * C) ** -0.5).bfloat16()
    bias = torch.randn(C, device="cuda")
    cells = []
    for t in tiles:
        us = time_it(lambda: cfm.gemm(img, w, bias=bias, act=cfm.ACT_RELU, conv=(C, T1, F1, T2, F2, B * T2 * F2), out_dtype=torch.bfloat16, tile=t), 10)
        cells.append("%7.1f (%6.1f)    " % (us, 2.0 * B * T2 * F2 * C * 9 * C / us / 1e6))
    print("%-12s %6d %6d %6d | " % ("conv2", B * T2 * F2, C, 9 * C) + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
