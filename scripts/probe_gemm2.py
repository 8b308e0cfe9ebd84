import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch, cfm
from probe_gemm import run, dev_us
print("--- N sweep at K=64, tile 1 (128x128): workgroups per CU 1,1,2,4,8")
for N in (256, 512, 1024, 2048, 4096):
    run("K=64 sweep", 7968, N, 64, tile=1)
print("--- same output bytes, short rows: M=63744 N=256")
run("K=64 tall", 63744, 256, 64, tile=1)
run("K=64 tall", 63744, 256, 64, tile=2)
print("--- no bias")
M, N, K = 7968, 2048, 64
a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
print("no bias: %.2f us" % dev_us(lambda: cfm.gemm(a, w, out=out, tile=1)))
print("--- pure copy kernels for scale: torch copy 32.6MB / 65MB")
src = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); dst = torch.empty_like(src)
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
print("torch copy bf16 32.6MB: %.2f us" % t(lambda: dst.copy_(src)))
print("torch fill bf16 32.6MB: %.2f us" % t(lambda: dst.zero_()))
x = torch.randn(7968, 256, device="cuda"); g = torch.ones(256, device="cuda"); b = torch.zeros(256, device="cuda")
o = torch.empty(7968, 256, device="cuda", dtype=torch.bfloat16)
print("layernorm dev: %.2f us" % dev_us(lambda: cfm.layernorm(x, g, b, want1=False, out2=o)))
print("cast 32.6MB dev: %.2f us" % dev_us(lambda: cfm.cast(src, torch.bfloat16)))
