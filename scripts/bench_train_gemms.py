"""The dense products of one conformer block's training step at a BASELINE config-3 micro-batch (M = B*T' rows, d = 256, ff = 2048), each
timed alone: forward (cfm_gemm), input gradient (cfm_gemm on the transposed pack), weight gradient (cfm_gemm_tn, accumulating).
Usage: python scripts/bench_train_gemms.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "conformer-pytorch-lightning_amd"))
import cfm  # noqa: E402


def timeit(fn, n=200, warm=20):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2380
    D, FF = 256, 2048
    dev = "cuda"
    bf = torch.bfloat16
    torch.manual_seed(0)
    print("M = %d rows; us per call (back-to-back launches, so launch gaps are included), TFLOP/s" % M)
    for name, N, K in (("ffn W1   [2048 <- 256]", FF, D), ("ffn W2   [256 <- 2048]", D, FF), ("qkv      [768 <- 256]", 3 * D, D),
                       ("out/pw2  [256 <- 256]", D, D), ("pw1      [512 <- 256]", 2 * D, D)):
        x = torch.randn((M, K), device=dev).to(bf)
        w = (torch.randn((N, K), device=dev) / K ** 0.5).to(bf)
        wt = w.t().contiguous()
        dy = torch.randn((M, N), device=dev).to(bf)
        bias = torch.randn((N,), device=dev)
        y = torch.empty((M, N), dtype=bf, device=dev)
        dx = torch.empty((M, K), dtype=torch.float32, device=dev)
        dw = torch.zeros((N, K), device=dev)
        db = torch.zeros((N,), device=dev)
        fl = 2.0 * M * N * K
        tf = timeit(lambda: cfm.gemm(x, w, bias=bias, out=y))
        tb = timeit(lambda: cfm.gemm(dy, wt, out=dx))
        tw = timeit(lambda: cfm.gemm_tn(dy, x, out=dw, colsum=db, want_colsum=True, accumulate=True))
        print("%-24s forward %6.1f us (%5.0f)   dgrad %6.1f us (%5.0f)   wgrad %6.1f us (%5.0f)" % (name, tf, fl / tf * 1e-6, tb, fl / tb * 1e-6, tw, fl / tw * 1e-6))
    # the cost of an empty-ish launch for scale
    z = torch.zeros((1024,), device=dev)
    print("torch add_ on 1k floats (launch floor): %.1f us" % timeit(lambda: z.add_(1.0)))


if __name__ == "__main__":
    main()
