import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch, cfm
M = 7968
def go(N, K, act, n=5, tile=1):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(n):
        cfm.gemm(a, w, bias=bias, act=act, out=out, tile=tile)
    torch.cuda.synchronize()
go(2048, 64, 0)      # dispatch group 1: floor case
go(2048, 256, cfm.ACT_SILU)   # ffn1
go(256, 2048, 0, tile=2)      # ffn2-like (no residual)
