#!/usr/bin/env python3
"""Device-time probes of the GEMM kernel (GPU box only): what does the epilogue / K depth / output width cost?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch  # noqa: E402
import cfm  # noqa: E402


def dev_us(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    cfm.prof_reset()
    cfm.prof_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    cfm.prof_enable(False)
    t = cfm.prof_table()
    cfm.prof_reset()
    return sum(e["ms"] for e in t.values()) / iters * 1e3


def run(name, M, N, K, act=0, res=False, out_dt=torch.bfloat16, tile=0, a_dt=torch.bfloat16):
    a = torch.randn(M, K, device="cuda").to(a_dt)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    cols = N // 2 if act == cfm.ACT_GLU else N
    r = torch.randn(M, cols, device="cuda") if res else None
    out = torch.empty(M, cols, device="cuda", dtype=torch.float32 if res else out_dt)
    us = dev_us(lambda: cfm.gemm(a, w, bias=bias, act=act, residual=r, alpha=0.5, out=out, tile=tile))
    print("%-44s M=%5d N=%5d K=%5d tile=%d  %8.2f us  %7.1f TF/s  out %6.1f MB" % (
        name, M, N, K, tile, us, 2.0 * M * N * K / us / 1e6, out.numel() * out.element_size() / 1e6), flush=True)


if __name__ == "__main__":
    M = 7968
    print("--- empty-ish kernels: launch + epilogue floor")
    run("K=64 N=256 bf16 out", M, 256, 64)
    run("K=64 N=2048 bf16 out", M, 2048, 64)
    run("K=64 N=2048 f32 out", M, 2048, 64, out_dt=torch.float32)
    print("--- FFN1 variants")
    for t in (1, 2, 3, 4):
        run("ffn1 silu", M, 2048, 256, cfm.ACT_SILU, tile=t)
    run("ffn1 no act", M, 2048, 256, 0, tile=1)
    run("ffn1 silu f32 out", M, 2048, 256, cfm.ACT_SILU, out_dt=torch.float32, tile=1)
    run("ffn1 K=128", M, 2048, 128, cfm.ACT_SILU, tile=1)
    run("ffn1 K=512", M, 2048, 512, cfm.ACT_SILU, tile=1)
    run("ffn1 K=1024", M, 2048, 1024, cfm.ACT_SILU, tile=1)
    print("--- FFN2 variants")
    for t in (1, 2, 3, 4, 5, 6):
        run("ffn2 +res", M, 256, 2048, 0, True, tile=t)
    run("ffn2 no res bf16 out", M, 256, 2048, 0, False, tile=2)
    print("--- small ones")
    for t in (2, 3, 5):
        run("out-proj +res", M, 256, 256, 0, True, tile=t)
        run("qkv", M, 768, 256, 0, False, tile=t)
        run("pw1 glu", M, 512, 256, cfm.ACT_GLU, False, tile=t)
    run("pos gemm (a f32)", 32, 256, 256, a_dt=torch.float32)
    print("--- square reference points")
    run("4096^3", 4096, 4096, 4096, tile=1)
    run("8192x8192x1024", 8192, 8192, 1024, tile=1)
