"""Experiment: the front-end's second convolution (config 2: image [32,499,39,256] bf16 -> [32*249*19, 256]) on the 128x128 kernel,
the 256x256 LDS-DMA kernel, and a batch split that gives the last partial round of 256x256 tiles to the 128x128 kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch, cfm
B, T1, F1, C, N = 32, 499, 39, 256, 256
T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
img = torch.randn(B, T1, F1, C, device="cuda").bfloat16()
w = (torch.randn(N, 9 * C, device="cuda") * (9 * C) ** -0.5).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty((B * T2 * F2, N), dtype=torch.bfloat16, device="cuda")
def conv(b0, b1, tile):
    cfm.gemm(img[b0:b1], w, bias=bias, act=cfm.ACT_RELU, conv=(C, T1, F1, T2, F2, (b1 - b0) * T2 * F2), out=out[b0 * T2 * F2:b1 * T2 * F2], tile=tile)
def timeit(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-48s %.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
timeit(lambda: conv(0, B, 0), "128x128 (auto)")
timeit(lambda: conv(0, B, 8), "256x256, all 592 tiles")
ref = out.clone(); conv(0, B, 0); assert torch.equal(ref, out)
for nb in (27, 28, 26, 24):
    timeit(lambda: (conv(0, nb, 8), conv(nb, B, 1)), "256x256 on %d utterances + 128x128 on %d" % (nb, B - nb))
    timeit(lambda: (conv(0, nb, 8), conv(nb, B, 2)), "256x256 on %d utterances + 64x128 on %d" % (nb, B - nb))
timeit(lambda: conv(0, 27, 8), "256x256 on 27 utterances alone")
timeit(lambda: conv(27, B, 1), "128x128 on 5 utterances alone")
