"""BASELINE config 4 at its real size (informational): 17-layer d=512 h=8 ff=2048 encoder forward on B=16 x (80 x 1000), then the transducer
joint (U+1 = 41, join 512, V = 5002).  Prints ms per encoder forward (graph replay), ms per joint call, the per-kernel table of the encoder
and the dominant kernel's fraction of the dense bf16 MFMA peak."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfm, encoder as enc_mod, joint as joint_mod
from oracle import conformer_oracle as O

prec = os.environ.get("CFM_PRECISION", "bf16")
cfm.set_precision(prec)
dev = torch.device("cuda", 0)
cfg = dict(input_dim=80, kernel_size=15, encoder_dim=512, dropout=0.1, attention_dropout=0.1, pos_enc_dropout=0.1, hidden_dim=2048, num_heads=8,
           encoder_num_layers=17, max_len=5000, use_relative=True)
torch.manual_seed(0)
enc = enc_mod.ConformerEncoder(cmvn=None, **cfg).eval().to(dev)
jn = joint_mod.TransducerJoint(5002, 512, 256, 512).eval().to(dev)
B, T, U = 16, 1000, 41
x = torch.from_numpy(np.random.RandomState(1234).standard_normal((B, T, 80)).astype(np.float32)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
pred = torch.randn(B, U, 256, device=dev)
stream = torch.cuda.Stream()
with torch.no_grad(), torch.cuda.stream(stream):
    for _ in range(3):
        y, m = enc(x, lens)
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        y, m = enc(x, lens)
    for _ in range(60):                                   # ~0.25 s: the clocks settle (bench.py's ramp note)
        g.replay()
    stream.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    stream.synchronize(); ms = (time.perf_counter() - t0) / 100 * 1e3
    fl = O.encoder_flops_per_utt(T, 80, 512, 2048, 15, 17) * B
    print("config 4 encoder (17 x d=512, B=16, T=1000, %s): %.3f ms per forward = %.2f M frames/s, %.1f TFLOP/s = %.3f of the dense MFMA peak"
          % (prec, ms, B * T / ms / 1e3, fl / ms / 1e9, fl / ms / 1e9 / 2500.0))
    for _ in range(2):
        out = jn(y, pred)
    stream.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out = jn(y, pred)
    stream.synchronize(); mj = (time.perf_counter() - t0) / 10 * 1e3
    print("config 4 joint (B=16, T'=249, U+1=41, join 512, V=5002 -> %.0f M logits): %.3f ms per call" % (out.numel() / 1e6, mj))
    del out
    cfm.prof_reset(); cfm.prof_enable(True)
    for _ in range(10):
        enc(x, lens)
    stream.synchronize(); cfm.prof_enable(False)
    tab = cfm.prof_table()
    tot = sum(e["ms"] for e in tab.values())
    print("encoder kernels (10 eager forwards): device time %.3f ms per forward" % (tot / 10))
    for k, e in sorted(tab.items(), key=lambda kv: -kv[1]["ms"]):
        tf = e["flops"] / (e["ms"] * 1e-3) / 1e12 if e["flops"] else 0.0
        print("  %-30s calls/fwd %4d  avg %8.2f us  share %5.1f%%  %8.1f TFLOP/s (%.3f of peak)" % (k, e["calls"] // 10, e["ms"] / e["calls"] * 1e3, 100 * e["ms"] / tot, tf, tf / 2500.0))
