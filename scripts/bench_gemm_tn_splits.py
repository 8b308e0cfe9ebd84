"""cfm_gemm_tn at a config-3 micro-batch (M rows, d = 256, ff = 2048): device time per call against the number of M splits, per weight shape.
The descriptor is built once and the C entry point is called in a tight loop, so the host is not the bottleneck.  Usage: [M]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "conformer-pytorch-lightning_amd"))
import cfm  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2380
    dev, bf = "cuda", torch.bfloat16
    lib = cfm.lib()
    print("M = %d; us per call (device-bound loop of 300 calls)" % M)
    for name, N, K in (("ffn W1 2048x256", 2048, 256), ("ffn W2 256x2048", 256, 2048), ("qkv 768x256", 768, 256), ("pw1 512x256", 512, 256), ("out 256x256", 256, 256)):
        a = torch.randn((M, N), device=dev).to(bf)
        b = torch.randn((M, K), device=dev).to(bf)
        c = torch.zeros((N, K), device=dev)
        cs = torch.zeros((N,), device=dev)
        row = []
        for tile, splits in ((64, 1), (64, 2), (64, 3), (64, 4), (64, 6), (64, 8), (128, 1), (128, 2), (128, 4)):
            d = cfm.GemmTnDesc()
            d.tile = tile
            d.A, d.B, d.C, d.colsum = a.data_ptr(), b.data_ptr(), c.data_ptr(), cs.data_ptr()
            d.lda, d.ldb, d.ldc, d.M, d.N, d.K = N, K, K, M, N, K
            d.a_dtype = d.b_dtype = d.mma_dtype = cfm.BF16
            d.accumulate, d.splits, d.alpha = 1, splits, 1.0
            st = cfm.stream()
            for _ in range(20):
                cfm.check(lib.cfm_gemm_tn(ctypes.byref(d), st), "cfm_gemm_tn")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                lib.cfm_gemm_tn(ctypes.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            row.append("%d/s%d %5.1f" % (tile, splits, e0.elapsed_time(e1) / 300 * 1e3))
        print("%-18s %s" % (name, "   ".join(row)))


if __name__ == "__main__":
    main()
