"""How does the fused FFN's time scale with the number of workgroups?  (per-CU vs aggregate L2 limit)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch, cfm
from cfm import packing
from probe_gemm import dev_us
D, FF = 256, 2048
w1 = torch.randn(FF, D, device="cuda") * D ** -0.5; w2 = torch.randn(D, FF, device="cuda") * FF ** -0.5
b1 = torch.randn(FF, device="cuda"); b2 = torch.randn(D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
w1f, w2f = packing.pack_ffn_fragments(w1, w2, torch.bfloat16)
for M in (32, 256, 512, 1024, 2048, 3984, 7968, 15936, 31872):
    x = torch.randn(M, D, device="cuda"); out = torch.empty_like(x)
    us = dev_us(lambda: cfm.ffn_fused(x, w1f, w2f, b1, b2, FF, ln=(g, b), alpha=0.5, add_x=True, ln1=(g, b), out_f32=out))
    wgs = (M + 31) // 32
    print("M=%6d  workgroups=%5d  %8.2f us   %7.1f TFLOP/s   weight stream %6.2f TB/s aggregate, %6.1f GB/s per active CU" % (
        M, wgs, us, 4.0 * M * D * FF / us / 1e6, wgs * 2.0 / us, 2.0e3 / us * min(1.0, 256.0 / wgs) if wgs > 256 else 2.0e3 / us), flush=True)
