"""How fast can 319 MB be written on this box?  (reference point for the front-end's first convolution)"""
import torch, time
n = 32 * 499 * 39 * 256
buf = torch.empty(n, dtype=torch.bfloat16, device="cuda")
src = torch.randn(n // 2, device="cuda")
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
us = t(lambda: buf.zero_())
print("zero_ (memset) of %.0f MB: %.1f us = %.2f TB/s" % (n * 2 / 1e6, us, n * 2 / us / 1e6))
us = t(lambda: buf.fill_(1.5))
print("fill_ of %.0f MB: %.1f us = %.2f TB/s" % (n * 2 / 1e6, us, n * 2 / us / 1e6))
b32 = buf.view(torch.float32)
us = t(lambda: torch.clamp_min(src, 0.0, out=b32))
print("relu f32 %.0f MB read + %.0f MB write: %.1f us = %.2f TB/s total" % (n * 2 / 1e6, n * 2 / 1e6, us, n * 4 / us / 1e6))
