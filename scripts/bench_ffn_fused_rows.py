"""The fused feed-forward launch (csrc/ffn.hip) and the macaron row chain (csrc/rowchain.hip) against the three launches of the training forward
(LayerNorm + W1/SiLU + W2/residual) at a training window's row counts.  Usage (GPU box): python scripts/bench_ffn_fused_rows.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "conformer-pytorch-lightning_amd"))
import cfm  # noqa: E402
from cfm import packing  # noqa: E402


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    D, FF = 256, 2048
    dev, bf = "cuda", torch.bfloat16
    prec = cfm.Precision("bf16")
    w1, w2 = torch.randn((FF, D), device=dev) * D ** -0.5, torch.randn((D, FF), device=dev) * FF ** -0.5
    b1, b2 = torch.zeros(FF, device=dev), torch.zeros(D, device=dev)
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    w1f, w2f = packing.pack_ffn_fragments(w1, w2, bf)
    w1m, w2m = w1.to(bf).contiguous(), w2.to(bf).contiguous()
    for M in [int(a) for a in sys.argv[1:]] or [2400, 3400, 4400, 7968]:
        x = torch.randn((M, D), device=dev)
        out = torch.empty_like(x)
        z = torch.empty((M, FF), dtype=bf, device=dev)
        t_fused = timeit(lambda: cfm.ffn_fused(x, w1f, w2f, b1, b2, FF, ln=(g, b), alpha=0.5, add_x=True, out_f32=out))

        def three():
            xn = cfm.layernorm(x, g, b, out1_dtype=bf)[0]
            h = cfm.gemm(xn, w1m, bias=b1, act=cfm.ACT_SILU, out_dtype=bf, pre_out=z, tile=cfm.TILE_AUTO_TRAIN)
            cfm.gemm(h, w2m, bias=b2, residual=x, alpha=0.5, tile=cfm.TILE_AUTO_TRAIN)
        t3 = timeit(three)
        print("M = %5d: fused feed-forward launch %6.1f us; LayerNorm + W1 (SiLU, pre-activation kept) + W2 (residual) %6.1f us" % (M, t_fused, t3), flush=True)


if __name__ == "__main__":
    main()
