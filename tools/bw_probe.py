"""HBM bandwidth this box gives plain streaming kernels (torch): the yardstick for the plane-bound chain kernels.  GPU box."""
import torch

n = 1 << 30                      # 4 GB of fp32
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 4 / 1e9
t = timed(lambda: x.sum())
print(f"read  (sum)      {gb / t / 1e3:.2f} TB/s")
t = timed(lambda: y.copy_(x))
print(f"copy  (r + w)    {2 * gb / t / 1e3:.2f} TB/s")
t = timed(lambda: y.fill_(1.0))
print(f"write (fill)     {gb / t / 1e3:.2f} TB/s")
t = timed(lambda: torch.add(x, y, out=y))
print(f"add   (2 r + w)  {3 * gb / t / 1e3:.2f} TB/s")
