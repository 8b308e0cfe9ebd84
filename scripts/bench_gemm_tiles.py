"""cfm_gemm at the shapes of a config-3 training micro-batch, device time per tile choice (descriptor built once, tight loop of C calls).
Usage: python scripts/bench_gemm_tiles.py [M]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "conformer-pytorch-lightning_amd"))
import cfm  # noqa: E402

TILES = {-1: "auto-train", 0: "auto", 1: "128x128", 2: "64x128", 3: "64x64", 4: "128x64", 5: "32x64", 6: "32x128", 9: "32x64k2", 10: "64x64k4", 11: "64x64k2"}


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2380
    shapes = (("N=2048 K=256 (ffn W1 fwd, dz dgrad)", 2048, 256), ("N=256 K=2048 (ffn W2 fwd, dxn dgrad)", 256, 2048), ("N=768 K=256 (qkv)", 768, 256),
              ("N=256 K=256 (out, pw2)", 256, 256), ("N=512 K=256 (pw1)", 512, 256), ("N=256 K=768 (dqkv dgrad)", 256, 768))
    if len(sys.argv) > 3:                               # python scripts/bench_gemm_tiles.py M N K: one shape (e.g. 7968 256 4864 = the front-end Linear of config 2)
        shapes = (("N=%s K=%s" % (sys.argv[2], sys.argv[3]), int(sys.argv[2]), int(sys.argv[3])),)
    dev, bf = "cuda", torch.bfloat16
    lib = cfm.lib()
    print("M = %d; us per call" % M)
    for name, N, K in shapes:
        a = torch.randn((M, K), device=dev).to(bf)
        w = torch.randn((N, K), device=dev).to(bf)
        c = torch.empty((M, N), dtype=bf, device=dev)
        bias = torch.zeros((N,), device=dev)
        row = []
        for tile in (-1, 1, 2, 3, 4, 5, 6, 9, 10, 11):
            d = cfm.GemmDesc()
            d.A, d.W, d.C, d.bias = a.data_ptr(), w.data_ptr(), c.data_ptr(), bias.data_ptr()
            d.lda, d.ldc, d.M, d.N, d.K = K, N, M, N, K
            d.a_dtype = d.w_dtype = d.c_dtype = cfm.BF16
            d.alpha, d.tile = 1.0, tile
            st = cfm.stream()
            if lib.cfm_gemm(ctypes.byref(d), st) != 0:
                row.append("%s  n/a" % TILES[tile])
                continue
            for _ in range(20):
                lib.cfm_gemm(ctypes.byref(d), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                lib.cfm_gemm(ctypes.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            row.append("%s %5.1f" % (TILES[tile], e0.elapsed_time(e1) / 300 * 1e3))
        print("%-40s %s" % (name, "   ".join(row)))


if __name__ == "__main__":
    main()
