import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch, cfm
from cfm import packing
M, D, FF = 7968, 256, 2048
x = torch.randn(M, D, device="cuda"); w1 = torch.randn(FF, D, device="cuda") * D ** -0.5; w2 = torch.randn(D, FF, device="cuda") * FF ** -0.5
b1 = torch.randn(FF, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
w1f, w2f = packing.pack_ffn_fragments(w1, w2, torch.bfloat16)
y0, y1 = torch.empty_like(x), torch.empty_like(x)
for _ in range(6):
    cfm.ffn_partial(x, (g, b), w1f, w2f, b1, FF, y0, y1)
torch.cuda.synchronize()
