"""CTC loss + gradient kernels at a config-3 micro-batch shape, the backward recursion beside the forward one (one launch) or inside ctc_grad.
Usage (GPU box): python scripts/bench_ctc_train.py [B T' U]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch  # noqa: E402

import cfm  # noqa: E402

B, T, U = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (10, 250, 60)
V, Vp = 5002, 5008
logits = torch.randn((B, T, Vp), device="cuda")
el = torch.full((B,), T, dtype=torch.int32, device="cuda")
lb = torch.randint(1, V, (B, U), dtype=torch.int32, device="cuda")
ll = torch.full((B,), U, dtype=torch.int32, device="cuda")
for now in (True, False, True, False):
    for _ in range(3):
        nll, st = cfm.ctc_nll_train(logits, V, el, lb, ll, beta_now=now)
        cfm.ctc_grad(logits, V, el, lb, ll, st)
    cfm.prof_reset()
    cfm.prof_enable(True)
    for _ in range(20):
        nll, st = cfm.ctc_nll_train(logits, V, el, lb, ll, beta_now=now)
        cfm.ctc_grad(logits, V, el, lb, ll, st)
    torch.cuda.synchronize()
    cfm.prof_enable(False)
    tab = cfm.prof_table()
    print("B %d T' %d U %d beta beside alpha: %s   " % (B, T, U, now) + "  ".join("%s %.1f us" % (k, e["ms"] / e["calls"] * 1e3) for k, e in sorted(tab.items())),
          "  total %.1f us" % sum(e["ms"] / e["calls"] * 1e3 for e in tab.values()))
