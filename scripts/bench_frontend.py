"""Front-end at BASELINE config 2 (32 x 1000 x 80, C = 256): the fused kernel (csrc/frontend.hip) against conv1 + implicit-GEMM conv2.
Usage: python scripts/bench_frontend.py [B T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "conformer-pytorch-lightning_amd"))
import cfm  # noqa: E402


def timeit(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    F, C = 80, 256
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.randn((B, T, F), device=dev)
    w1 = torch.randn((9, C), device=dev) * 0.3
    b1 = torch.randn((C,), device=dev) * 0.1
    w2 = (torch.randn((C, 9 * C), device=dev) / 48.0).bfloat16()
    b2 = torch.randn((C,), device=dev) * 0.1
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    M = B * T2 * F2
    flops = 2.0 * M * C * 9 * C

    def two():
        h1 = cfm.conv1_relu(x, w1, b1, torch.bfloat16, mma=True)
        return cfm.gemm(h1, w2, bias=b2, act=cfm.ACT_RELU, conv=(C, T1, F1, T2, F2, M), out_dtype=torch.bfloat16)

    def fused():
        return cfm.conv12_relu(x, w1, b1, w2, b2)

    a, b = two(), fused()
    print("rows %d  bit-identical: %s" % (M, torch.equal(a.view(torch.int16), b.view(torch.int16))))
    t2, tf = timeit(two), timeit(fused)
    print("conv1 + conv2 (two kernels + tail) %8.1f us   %6.0f TFLOP/s" % (t2, flops / t2 * 1e-6))
    print("fused (frontend.hip)               %8.1f us   %6.0f TFLOP/s (conv2 FLOPs only)" % (tf, flops / tf * 1e-6))


if __name__ == "__main__":
    main()
