"""CTC head at the config-3 shape (informational): B=32, T'=249, D=256, V=5002, Umax=40, bf16 projection."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfm, decoder
cfm.set_precision("bf16")
B, T, D, V, U = 32, 249, 256, 5002, 40
dec = decoder.CTCDecoder(V, D, 0.0).eval().to("cuda")
x = torch.randn(B, T, D, device="cuda")
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
labels = torch.randint(1, V, (B, U), dtype=torch.int32, device="cuda")
llens = torch.full((B,), U, dtype=torch.int32, device="cuda")
with torch.no_grad():
    for _ in range(3):
        dec(x, lens, labels, llens)
    cfm.prof_reset(); cfm.prof_enable(True)
    for _ in range(20):
        loss = dec(x, lens, labels, llens)
    torch.cuda.synchronize(); cfm.prof_enable(False)
    for k, e in sorted(cfm.prof_table().items(), key=lambda kv: -kv[1]["ms"]):
        print("%-28s %8.2f us/launch  %7.1f TFLOP/s  %7.1f GB/s" % (k, e["ms"] / e["calls"] * 1e3, e["flops"] / (e["ms"] * 1e-3) / 1e12 if e["flops"] else 0.0,
                                                                     e["bytes"] / (e["ms"] * 1e-3) / 1e9))
    print("loss", float(loss))
