"""Config 5 (informational): 64 streams in lockstep, chunk = 16 output frames (67-frame windows), left context capped at 4 chunks
(Tc = 64), the config-2 model, bf16.  Prints ms per streaming step in steady state and stream-frames per second."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "conformer-pytorch-lightning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfm, bench

B, chunk, left = int(os.environ.get("STREAMS", "64")), 16, 4
cfm.set_precision("bf16")
dev = torch.device("cuda", 0)
enc = bench.build_encoder(dev)
window, hop, need = (chunk - 1) * 4 + 7, 4 * chunk, chunk * left
x = torch.from_numpy(np.random.RandomState(5).standard_normal((B, window + 40 * hop, 80)).astype(np.float32)).to(dev)
empty = torch.zeros((0, 0, 0, 0), device=dev)
cache, offset, times = empty, 0, []
with torch.no_grad():
    for step in range(40):
        win = x[:, step * hop: step * hop + window].contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y, cache, _ = enc.forward_chunk(win, offset, need, cache, empty)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        offset += y.size(1)
steady = sorted(times[10:])
ms = 1e3 * steady[len(steady) // 2]
print("streams %d  chunk %d  cache %s: %.3f ms per step (median of steady state, eager launches) = %.0f input frames/s over all streams (%.1f x real time per stream at 10 ms frames)"
      % (B, chunk, tuple(cache.shape), ms, B * hop / (ms * 1e-3), hop * 10.0 / ms))

# ---- the same steps through encoder.StreamingSession: one captured HIP graph per steady-state step
import encoder as enc_mod
sess = enc_mod.StreamingSession(enc, chunk, left)
times = []
with torch.no_grad():
    for step in range(40):
        win = x[:, step * hop: step * hop + window].contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = sess.step(win)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
steady = sorted(times[12:])
ms = 1e3 * steady[len(steady) // 2]
print("streams %d  graph session: %.3f ms per step (median of replayed steps) = %.0f input frames/s over all streams (%.1f x real time per stream)"
      % (B, ms, B * hop / (ms * 1e-3), hop * 10.0 / ms))

# ---- round 2: encoder.StreamingBatch -- per-stream offsets on the device, K/V ring buffers (no cat + trim), the whole step one graph
sb = enc_mod.StreamingBatch(enc, B, chunk, left)
times = []
with torch.no_grad():
    for step in range(40):
        win = x[:, step * hop: step * hop + window].contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = sb.step(win)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
steady = sorted(times[12:])
ms = 1e3 * steady[len(steady) // 2]
print("streams %d  StreamingBatch (ring KV, per-stream offsets, one graph): %.3f ms per step = %.0f input frames/s over all streams (%.1f x real time per stream)"
      % (B, ms, B * hop / (ms * 1e-3), hop * 10.0 / ms))

# ---- per-kernel table of one StreamingBatch step (eager launches, HIP events attached to every dispatch)
if os.environ.get("KERNEL_TABLE", "1") != "0":
    sb2 = enc_mod.StreamingBatch(enc, B, chunk, left, graph=False)
    with torch.no_grad():
        for step in range(8):
            sb2.step(x[:, step * hop: step * hop + window].contiguous())
        torch.cuda.synchronize()
        cfm.prof_reset(); cfm.prof_enable(True)
        n = 20
        for step in range(8, 8 + n):
            sb2.step(x[:, step * hop: step * hop + window].contiguous())
        torch.cuda.synchronize(); cfm.prof_enable(False)
    tab = cfm.prof_table()
    tot = sum(e["ms"] for e in tab.values())
    print("StreamingBatch step, kernels (eager, %d steps): device time %.3f ms per step" % (n, tot / n))
    for k, e in sorted(tab.items(), key=lambda kv: -kv[1]["ms"]):
        print("  %-34s calls/step %5.1f  avg %7.2f us  share %5.1f%%" % (k, e["calls"] / n, e["ms"] / e["calls"] * 1e3, 100 * e["ms"] / tot))
