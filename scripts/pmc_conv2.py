"""Launch the config-2 front-end conv2 implicit GEMM a few times (for rocprofv3 --pmc runs on the GPU box).
   python scripts/pmc_conv2.py [tile]       tile 0 = auto (split launch), 1 = 128x128, 8 = 256x256 LDS-DMA"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch, cfm
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, T1, F1, C = 32, 499, 39, 256
T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
img = torch.randn(B, T1, F1, C, device="cuda").bfloat16()
w = (torch.randn(C, 9 * C, device="cuda") * (9 * C) ** -0.5).bfloat16()
b = torch.randn(C, device="cuda")
out = torch.empty(B * T2 * F2, C, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    cfm.gemm(img, w, bias=b, act=cfm.ACT_RELU, conv=(C, T1, F1, T2, F2, B * T2 * F2), out=out, tile=tile)
torch.cuda.synchronize()
