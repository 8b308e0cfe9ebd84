"""Experiment: per-workgroup fixed cost vs per-K-tile cost of cfm_gemm's 128x128 tile (M=163344, N=5120, bf16 out)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd"))
import torch, cfm
def run(M, N, K, odt=torch.bfloat16, tile=1):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    buf = torch.empty((M, N), dtype=odt, device="cuda")
    for _ in range(2):
        cfm.gemm(a, w, out=buf, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        cfm.gemm(a, w, out=buf, tile=tile)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    rounds = tiles / 512.0
    print("tile=%d " % tile + "M=%d N=%d K=%d %s: %.1f us  %.1f TFLOP/s  | %d tiles, %.1f rounds of 512 -> %.2f us per workgroup, %d K tiles" %
          (M, N, K, str(odt).split(".")[-1], ms * 1e3, 2.0 * M * N * K / ms / 1e9, tiles, rounds, ms * 1e3 / rounds, K // 64), flush=True)
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    run(16 * 249 * 41 // 2, 5120, K)
    run(16 * 249 * 41 // 2, 5120, K, tile=7)
    run(16 * 249 * 41 // 2, 5120, K, tile=8)
for K in (512,):
    run(16 * 249 * 41, 5002, K, torch.float32, tile=7)
    run(16 * 249 * 41, 5002, K, torch.float32, tile=8)
    run(16 * 249 * 41, 5002, K, torch.bfloat16, tile=8)
# one round only: 512 tiles (latency of a lone workgroup pair per CU)

