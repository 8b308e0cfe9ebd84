"""Experiment: how the logits' row stride / store width changes the vocabulary GEMM of the joint (M=163344, K=512, N~5002)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch, cfm
M, K = 16 * 249 * 41, 512
a = torch.randn(M, K, device="cuda").bfloat16()
from cfm import packing
def run(N, ldc, odt, tile=0, bias=True):
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda") if bias else None
    buf = torch.empty((M, ldc), dtype=odt, device="cuda")
    for _ in range(2):
        cfm.gemm(a, w, bias=b, out=buf[:, :N], tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        cfm.gemm(a, w, bias=b, out=buf[:, :N], tile=tile)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("N=%d ldc=%d %s tile=%d: %.1f us  %.1f TFLOP/s" % (N, ldc, str(odt).split(".")[-1], tile, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
for odt in (torch.float32, torch.bfloat16):
    run(5002, 5002, odt)
    run(5004, 5004, odt)
    run(5002, 5024, odt)   # 32-float aligned rows, pair path avoided? (ldc%4==0 -> vec4 except last pair)
    run(5120, 5120, odt)
    run(5120, 5120, odt, tile=4)
    run(5002, 5002, odt, tile=1)
    run(5002, 5002, odt, tile=7)
    run(5120, 5120, odt, tile=7)
