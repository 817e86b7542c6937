"""Probe (tuning build with -DW4_TIMING=1, loaded through MDIL_HIP_LIB): wall-clock stamps per WAVE of
one w4conv launch: entry, weights resident, then per tile (MFMA loop done, stores acknowledged)."""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
import mdil_ss_amd
from mdil_ss_amd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.load()
lib.mdil_debug_set_w4conv_stamps.argtypes = [ctypes.c_void_p]
for C, (H, W), d, adapt in ((128, (64, 128), 2, False), (128, (64, 128), 2, True), (64, (128, 256), 1, False)):
    N = 6
    x = torch.randn(N, H, W, C, device=dev).relu_(); out = torch.empty_like(x)
    x2 = torch.randn(N, H, W, C, device=dev)
    w3 = torch.randn(C, C, 1, 3, device=dev) * 0.05; b = torch.randn(C, device=dev)
    pw = torch.randn(C, C, 1, 1, device=dev) * 0.05
    if adapt:
        wp = ops.pack_pair(w3, pw, 'fwd')
        g3 = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d) + [(0, 0, 1)], C, H, W, C)
        run = lambda: ops.tapconv(g3, C, C, x, x2, wp, out, bias=b)
    else:
        wp = ops.pack_conv(w3, 'fwd')
        g3 = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d), C, H, W, C)
        run = lambda: ops.tapconv(g3, C, C, x, None, wp, out, bias=b, relu=True)
    nwg = 256
    st = torch.zeros(nwg * 8 * 32, dtype=torch.int64, device=dev)
    for _ in range(3): run()
    torch.cuda.synchronize()
    lib.mdil_debug_set_w4conv_stamps(st.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.mdil_debug_set_w4conv_stamps(None)
    s = st.view(nwg, 8, 32)[:, :, :16].cpu().double()
    t0 = s[:, :, 0][s[:, :, 0] > 0].min()
    s = torch.where(s > 0, (s - t0) / 100.0, torch.full_like(s, float('nan')))   # 100 MHz -> us
    q = lambda v: [round(float(torch.nanquantile(v.flatten(), p)), 2) for p in (0.0, 0.1, 0.5, 0.9, 1.0)]
    print(f"C={C} adapter={adapt}: us since first wave entry; min/p10/median/p90/max")
    print("  entry              ", q(s[:, :, 0]))
    print("  weights resident   ", q(s[:, :, 1]))
    if not torch.isnan(s[:, :, 10]).all():           # fill breakdown (round 6): per-wave stamps inside the prologue
        print("  fill: raw taps arrived since entry      ", q(s[:, :, 10] - s[:, :, 0]))
        print("  fill: transform + LDS writes            ", q(s[:, :, 11] - s[:, :, 10]))
        print("  fill: set-up + first operand requests   ", q(s[:, :, 12] - s[:, :, 11]))
        print("  fill: barrier wait (own stamp 12 -> 1)  ", q(s[:, :, 1] - s[:, :, 12]))
        print("  fill: wave entry -> weights resident    ", q(s[:, :, 1] - s[:, :, 0]))
    for grp, name in ((slice(0, 4), "waves 0-3"), (slice(4, 8), "waves 4-7")):
        for k in range(6):
            a, b_ = s[:, grp, 2 + 2 * k], s[:, grp, 3 + 2 * k]
            if torch.isnan(a).all():
                continue
            prev = s[:, grp, 1] if k == 0 else s[:, grp, 1 + 2 * k]
            print(f"  {name} tile {k}: mfma done", q(a), " loop dur", q(a - prev), " epilogue dur", q(b_ - a))
    print("  kernel end         ", round(float(torch.nan_to_num(s[:, :, :14], nan=0.0).max()), 2))
    raw = st.view(nwg, 8, 32).cpu().double()
    for k in range(6):
        i0, i1 = (1 if k == 0 else 1 + 2 * k), 2 + 2 * k
        ok = raw[:, :, i1] > 0
        if not ok.any():
            continue
        cyc = (raw[:, :, 16 + i1] - raw[:, :, 16 + i0])[ok]
        us = ((raw[:, :, i1] - raw[:, :, i0]) / 100.0)[ok]
        print(f"  tile {k} loop: cycles", q(cyc), " clock GHz", q(cyc / (us * 1e3)))
