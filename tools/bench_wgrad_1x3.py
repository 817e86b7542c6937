import os, sys, torch
sys.path.insert(0, '/root/repo')
import mdil_ss_amd
from mdil_ss_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for C, (H, W), ds in ((128, (64, 128), (2, 16)), (64, (128, 256), (1,))):
    x = torch.randn(6, H, W, C, device=dev); g = torch.randn(6, H, W, C, device=dev)
    w = torch.randn(C, C, 1, 3, device=dev); b = torch.randn(C, device=dev)
    for d in ds:
        G = ops.make_geom(6, H, W, H, W, ops._taps_1x3(d), C, H, W, C)
        t = timeit(lambda: ops.wgrad(G, C, C, x, None, g, (0, 1, 2), C * 3, 3, w, b))
        print(f"wgrad{C} 1x3 d{d} (+reduce): {t:.1f} us  (MDIL_NO_WGRADX={os.environ.get('MDIL_NO_WGRADX')})")
