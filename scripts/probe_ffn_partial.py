import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch, cfm
from cfm import packing
from probe_gemm import dev_us
M, D, FF = 7968, 256, 2048
x = torch.randn(M, D, device="cuda"); w1 = torch.randn(FF, D, device="cuda") * D ** -0.5; w2 = torch.randn(D, FF, device="cuda") * FF ** -0.5
b1 = torch.randn(FF, device="cuda"); b2 = torch.randn(D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
w1f, w2f = packing.pack_ffn_fragments(w1, w2, torch.bfloat16)
y0, y1, xo, out = (torch.empty_like(x) for _ in range(4))
print("ffn_fused (32-row tiles, full)      %7.2f us" % dev_us(lambda: cfm.ffn_fused(x, w1f, w2f, b1, b2, FF, ln=(g, b), alpha=0.5, add_x=True, ln1=(g, b), out_f32=out)))
print("ffn_partial plain input             %7.2f us" % dev_us(lambda: cfm.ffn_partial(x, (g, b), w1f, w2f, b1, FF, y0, y1)))
z0, z1 = torch.empty_like(x), torch.empty_like(x)
print("ffn_partial reduce input            %7.2f us" % dev_us(lambda: cfm.ffn_partial(x, (g, b), w1f, w2f, b1, FF, z0, z1, pending=(y0, y1, b2, 0.5, (g, b)), x_out=xo)))
a16 = torch.randn(M, D, device="cuda").bfloat16(); wh = packing.pack_frag_major(torch.randn(D, D, device="cuda") * D ** -0.5, torch.bfloat16)
print("ffn_partial head input              %7.2f us" % dev_us(lambda: cfm.ffn_partial(x, (g, b), w1f, w2f, b1, FF, z0, z1, head=(a16, wh, b2, None), x_out=xo)))
wq = packing.pack_frag_major(torch.randn(3 * D, D, device="cuda") * D ** -0.5, torch.bfloat16); bq = torch.randn(3 * D, device="cuda")
qkv = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
print("rowchain reduce + LN + QKV tail     %7.2f us" % dev_us(lambda: cfm.rowchain(M, D, cfm.BF16, x=x, pending=(y0, y1, b2, 0.5, None), ln=(g, b), out_f32=out, tail=(wq, bq, 3 * D, False, qkv))))
print("rowchain reduce rows only           %7.2f us" % dev_us(lambda: cfm.rowchain(M, D, cfm.BF16, x=x, pending=(y0, y1, b2, 0.5, (g, b)), out_f32=out)))
