"""The front-end's second convolution's weight gradient (cfm_gemm_tn with the implicit im2col operand) at a config-3 micro-batch: device time per
tile / split choice.  Usage (GPU box): python scripts/bench_conv2_wgrad.py [B] [T]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "conformer-pytorch-lightning_amd"))
import cfm  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    C, F = 256, 80
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    M = B * T2 * F2
    dev, bf = "cuda", torch.bfloat16
    lib = cfm.lib()
    img = torch.randn((B, T1, F1, C), device=dev).to(bf)
    dy = torch.randn((M, C), device=dev).to(bf)
    out = torch.zeros((C, 9 * C), device=dev)
    cs = torch.zeros((C,), device=dev)
    print("conv2 weight gradient: B = %d, T = %d -> M = %d rows, N = %d, K = %d (%.1f GFLOP)" % (B, T, M, C, 9 * C, 2.0 * M * C * 9 * C / 1e9))
    for tile, splits in ((0, 0), (128, 4), (128, 7), (128, 11), (128, 14), (64, 1), (64, 2), (64, 3), (64, 4), (64, 6)):
        d = cfm.GemmTnDesc()
        d.tile = tile
        d.A, d.B, d.C, d.colsum = dy.data_ptr(), img.data_ptr(), out.data_ptr(), cs.data_ptr()
        d.lda, d.ldc, d.M, d.N, d.K = C, 9 * C, M, C, 9 * C
        d.a_dtype = d.b_dtype = d.mma_dtype = cfm.BF16
        d.conv_C, d.conv_T1, d.conv_F1, d.conv_T2, d.conv_F2 = C, T1, F1, T2, F2
        d.accumulate, d.splits, d.alpha = 1, splits, 1.0
        st = cfm.stream()
        for _ in range(5):
            cfm.check(lib.cfm_gemm_tn(ctypes.byref(d), st), "cfm_gemm_tn")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.cfm_gemm_tn(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print("  tile %3s splits %2s  %7.1f us  %6.1f TFLOP/s" % (tile or "auto", splits or "auto", us, 2.0 * M * C * 9 * C / us / 1e6))


if __name__ == "__main__":
    main()
