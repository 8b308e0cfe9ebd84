"""Transducer joint at the BASELINE config-4 shape (informational): B=16, T'=249, U+1=41, encoder/predictor 512, join 512, V=5002.
   python scripts/bench_joint.py [bf16|fp16|fp32] [f32|16]      (second argument: logits dtype)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch
import cfm, joint
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
o16 = len(sys.argv) > 2 and sys.argv[2] == "16"
cfm.set_precision(mode)
B, T, U, E, P, J, V = 16, 249, 41, 512, 512, 512, 5002
jn = joint.TransducerJoint(V, E, P, J).eval().to("cuda")
if o16:
    jn.out_dtype = cfm.get_precision().w_dtype
enc, pred = torch.randn(B, T, E, device="cuda"), torch.randn(B, U, P, device="cuda")
with torch.no_grad():
    for _ in range(3):
        out = jn(enc, pred)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        out = jn(enc, pred)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * B * T * U * J * V
    print("joint %s, logits %s: %.3f ms per call (%d x %d x %d x %d logits, %.2f GB); ffn_out alone = %.1f GFLOP -> %.1f TFLOP/s whole call"
          % (mode, out.dtype, ms, B, T, U, V, out.numel() * out.element_size() / 1e9, flops / 1e9, flops / (ms * 1e-3) / 1e12))
    cfm.prof_reset(); cfm.prof_enable(True)
    for _ in range(5):
        out = jn(enc, pred)
    torch.cuda.synchronize(); cfm.prof_enable(False)
    for k, e in sorted(cfm.prof_table().items(), key=lambda kv: -kv[1]["ms"]):
        print("%-28s %9.2f us/launch  %7.1f TFLOP/s  %7.1f GB/s" % (k, e["ms"] / e["calls"] * 1e3, e["flops"] / (e["ms"] * 1e-3) / 1e12 if e["flops"] else 0.0,
                                                                      e["bytes"] / (e["ms"] * 1e-3) / 1e9))
