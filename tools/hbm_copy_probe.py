"""Probe: what a plain device copy / read-only reduction reaches on this box (the ceiling the InstanceNorm passes run against)."""
import torch
dev = torch.device("cuda:0")
n = 2 * 32 * 128 ** 3
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
t = timeit(lambda: y.copy_(x)); print(f"copy 537 MB -> 537 MB: {t:.3f} ms = {2 * n * 4 / t / 1e9:.2f} TB/s")
t = timeit(lambda: torch.add(x, 1.0, out=y)); print(f"add scalar (read + write): {t:.3f} ms = {2 * n * 4 / t / 1e9:.2f} TB/s")
t = timeit(lambda: x.sum()); print(f"sum (read only): {t:.3f} ms = {n * 4 / t / 1e9:.2f} TB/s")
t = timeit(lambda: torch.add(x, y, out=y)); print(f"add (2 reads + write): {t:.3f} ms = {3 * n * 4 / t / 1e9:.2f} TB/s")
