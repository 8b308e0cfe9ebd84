"""Race screen (GPU box) for the workgroup-pair route of config 4 (d = 512): the same forward must return bitwise identical output every time (the pairs of
a row tile exchange nothing inside a launch; what one writes and the other reads across launches is ordered by the stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import numpy as np, torch
import cfm, encoder as enc_mod
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = dict(input_dim=80, kernel_size=15, encoder_dim=512, dropout=0.1, attention_dropout=0.1, pos_enc_dropout=0.1, hidden_dim=2048, num_heads=8,
           encoder_num_layers=17, max_len=5000, use_relative=True)
for mode in ("bf16", "fp16"):
    cfm.set_precision(mode)
    torch.manual_seed(0)
    enc = enc_mod.ConformerEncoder(cmvn=None, **cfg).eval().to(dev)
    x = torch.from_numpy(np.random.RandomState(9).standard_normal((16, 1000, 80)).astype(np.float32)).to(dev)
    lens = torch.from_numpy(np.sort(np.random.RandomState(3).randint(500, 1001, size=16))[::-1].copy().astype(np.int32)).to(dev)
    with torch.no_grad():
        ref, mref = enc(x, lens)
        bad = sum(int(not torch.equal(enc(x, lens)[0], ref)) for _ in range(N))
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            enc(x, lens); st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                y, m = enc(x, lens)
            for _ in range(N):
                g.replay(); st.synchronize()
                bad += int(not torch.equal(y, ref))
    print("%s: %d mismatching runs out of %d (eager + graph), ragged batch of 16, output finite: %s" % (mode, bad, 2 * N, bool(torch.isfinite(ref).all())))
    assert bad == 0
print("soak ok")
