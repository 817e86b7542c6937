import sys, torch
sys.path.insert(0, '.')
import mdil_ss_amd
from mdil_ss_amd import ops
sys.path.insert(0, 'tools')
from bench_kernels import timeit
dev = torch.device('cuda:0')
for C, (H, W) in ((128, (64, 128)), (64, (128, 256))):
    w3 = torch.randn(C, C, 3, 1, device=dev) * 0.05
    b = torch.randn(C, device=dev)
    wp = ops.pack_conv(w3, 'fwd')
    for N in (1, 2, 3, 6, 12, 24):
        x = torch.randn(N, H, W, C, device=dev).relu_()
        out = torch.empty_like(x)
        g3 = ops.make_geom(N, H, W, H, W, ops._taps_3x1(2), C, H, W, C)
        t = timeit(lambda: ops.tapconv(g3, C, C, x, None, wp, out, bias=b, relu=True), 30)
        fl = 2.0 * N * H * W * 3 * C * C
        print(f"C{C} N={N:2d} WGs={N*H*W//(64 if C==128 else 128):5d}  {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TF  us/img {t*1e6/N:6.2f}")
