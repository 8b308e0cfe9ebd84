"""Race screen (GPU box): the config-2 forward, eager and graph-replayed, must return bitwise identical output every time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfm, bench
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for mode in ("bf16", "fp16", "fp32"):
    cfm.set_precision(mode)
    enc = bench.build_encoder(dev)
    x = torch.from_numpy(np.random.RandomState(9).standard_normal((32, 1000, 80)).astype(np.float32)).to(dev)
    lens = torch.from_numpy(np.sort(np.random.RandomState(3).randint(600, 1001, size=32))[::-1].copy().astype(np.int32)).to(dev)
    with torch.no_grad():
        ref, mref = enc(x, lens)
        bad = 0
        for i in range(N if mode != "fp32" else N // 4):
            y, m = enc(x, lens)
            bad += int(not (torch.equal(y, ref) and torch.equal(m, mref)))
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            y, m = enc(x, lens); stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                y, m = enc(x, lens)
            for i in range(N if mode != "fp32" else N // 4):
                g.replay()
                stream.synchronize()
                bad += int(not torch.equal(y, ref))
    print("%s: %d mismatching runs out of %d (eager + graph), output finite: %s" % (mode, bad, 2 * (N if mode != "fp32" else N // 4), bool(torch.isfinite(ref).all())), flush=True)
    assert bad == 0
print("soak ok")
