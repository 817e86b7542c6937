"""Probe: pure-write vs read+write HBM bandwidth on MI355X (torch elementwise kernels)."""
import torch
dev = torch.device('cuda:0')
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for mb in (25, 50, 100, 400):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    t = timeit(lambda: x.fill_(1.0)); print(f"fill   {mb:4d} MB  {t*1e6:7.1f} us  write {mb*1.048576/1e3/t/1e3:6.2f} TB/s")
    t = timeit(lambda: y.copy_(x));   print(f"copy   {mb:4d} MB  {t*1e6:7.1f} us  r+w   {2*mb*1.048576/1e3/t/1e3:6.2f} TB/s")
    t = timeit(lambda: x.sum());      print(f"sum    {mb:4d} MB  {t*1e6:7.1f} us  read  {mb*1.048576/1e3/t/1e3:6.2f} TB/s")
