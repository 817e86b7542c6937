"""Probe (tuning build with -DTC_TIMING=1): wall-clock stamps per workgroup of one tapconv launch:
entry, first stage staged, main loop done, stores acknowledged."""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
import mdil_ss_amd
from mdil_ss_amd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.load()
lib.mdil_debug_set_stamps.argtypes = [ctypes.c_void_p]
for C, (H, W), BM in ((128, (64, 128), 64), (64, (128, 256), 128)):
    N = 6
    x = torch.randn(N, H, W, C, device=dev).relu_(); out = torch.empty_like(x)
    w3 = torch.randn(C, C, 3, 1, device=dev) * 0.05; b = torch.randn(C, device=dev)
    wp = ops.pack_conv(w3, 'fwd')
    g3 = ops.make_geom(N, H, W, H, W, ops._taps_3x1(2), C, H, W, C)
    nwg = N * H * W // BM
    st = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
    for _ in range(3): ops.tapconv(g3, C, C, x, None, wp, out, bias=b, relu=True)
    torch.cuda.synchronize()
    lib.mdil_debug_set_stamps(st.data_ptr())
    ops.tapconv(g3, C, C, x, None, wp, out, bias=b, relu=True)
    torch.cuda.synchronize()
    lib.mdil_debug_set_stamps(None)
    s = st.view(nwg, 4).cpu().double()
    t0 = s[:, 0].min()
    s = (s - t0) / 100.0            # wall_clock64 ticks at 100 MHz -> microseconds
    q = lambda v: [round(float(torch.quantile(v, p)), 2) for p in (0.0, 0.1, 0.5, 0.9, 1.0)]
    print(f"C={C}: {nwg} WGs  (us since first WG entry; min/p10/median/p90/max)")
    print("  entry            ", q(s[:, 0]))
    print("  stage0 staged    ", q(s[:, 1]), " duration", q(s[:, 1] - s[:, 0]))
    print("  main loop done   ", q(s[:, 2]), " duration", q(s[:, 2] - s[:, 1]))
    print("  stores acked     ", q(s[:, 3]), " duration", q(s[:, 3] - s[:, 2]))
