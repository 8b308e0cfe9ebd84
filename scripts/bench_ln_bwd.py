"""cfm_layernorm_bwd_fused at a training window's row count: what do the gamma / beta atomics, the second (dropout) output and the residual cost?
Usage (GPU box): python scripts/bench_ln_bwd.py [M]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "conformer-pytorch-lightning_amd"))
import cfm  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 3400
    D = 256
    dev = "cuda"
    x, dy, dres = (torch.randn((M, D), device=dev) for _ in range(3))
    gamma = torch.randn(D, device=dev)
    dx = torch.empty_like(x)
    dx2 = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ws = torch.empty(cfm.lib().cfm_layernorm_bwd_ws(M, D), device=dev)
    lib = cfm.lib()
    for name, acc, use2, p, res in (("atomics + dropout output + residual (the block's call)", 1, 1, 0.1, 1), ("partials + reduce launch, same outputs", 0, 1, 0.1, 1),
                                    ("atomics, second output without dropout", 1, 1, 0.0, 1), ("atomics, no second output", 1, 0, 0.0, 1),
                                    ("partials + reduce, no second output", 0, 0, 0.0, 1), ("atomics, nothing else (no residual)", 1, 0, 0.0, 0)):
        d = cfm.LnBwdDesc()
        d.x, d.dy, d.gamma, d.dres, d.dx, d.dgamma, d.dbeta, d.ws = x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), (dres.data_ptr() if res else None), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr()
        d.M, d.D, d.dy_dtype, d.accumulate, d.eps = M, D, cfm.F32, acc, 1e-5
        if use2:
            d.dx2, d.dx2_dtype, d.alpha2, d.p1, d.seed1 = dx2.data_ptr(), cfm.BF16, 0.5, p, 1234
        st = cfm.stream()
        for _ in range(20):
            lib.cfm_layernorm_bwd_fused(ctypes.byref(d), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            lib.cfm_layernorm_bwd_fused(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        print("M = %d  %-62s %6.1f us per call" % (M, name, e0.elapsed_time(e1) / 300 * 1e3))


if __name__ == "__main__":
    main()
