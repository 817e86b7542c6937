import sys, torch
sys.path.insert(0, '.')
import mdil_ss_amd
from mdil_ss_amd import ops
dev = torch.device('cuda:0'); C = 128
for (N, H, W, axis, d) in ((1, 8, 16, 'w', 2), (6, 64, 128, 'w', 16)):
    for epi in ('bias', 'bias_relu', 'gate', 'res', 'res_resgate'):
        g = torch.Generator().manual_seed(7)
        x = torch.randn(N, H, W, C, generator=g).to(dev)
        w = (torch.randn(C, C, 1, 3, generator=g) * 0.05).to(dev)
        b = torch.randn(C, generator=g).to(dev)
        r = torch.randn(N, H, W, C, generator=g).to(dev); r2 = torch.randn(N, H, W, C, generator=g).to(dev)
        geom = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d), C, H, W, C)
        wp = ops.pack_conv(w, 'fwd')
        y = torch.full((N, H, W, C), float('nan'), device=dev)
        kw = {'bias': dict(bias=b), 'bias_relu': dict(bias=b, relu=True), 'gate': dict(gate=r), 'res': dict(res=r),
              'res_resgate': dict(res=r, res_gate=r2)}[epi]
        ops.tapconv(geom, C, C, x, None, wp, y, **kw)
        z = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, None, padding=(0, d), dilation=(1, d)).permute(0, 2, 3, 1)
        if 'bias' in kw: z = z + b
        if 'res' in kw: z = z + (torch.where(r2 > 0, r, torch.zeros_like(r)) if 'res_gate' in kw else r)
        if kw.get('relu'): z = z.relu()
        if 'gate' in kw: z = torch.where(r > 0, z, torch.zeros_like(z))
        bad = (y - z).abs() > 1e-3
        print(f"{(N,H,W,axis,d)} {epi:12s} bad {int(bad.sum())}/{bad.numel()}  nan {int(torch.isnan(y).sum())}", end='')
        if bad.any():
            per_c = bad.reshape(-1, C).sum(0); per_w = bad.reshape(N * H, W, C).sum((0, 2))
            print("  bad per 16-channel group:", per_c.reshape(8, 16).sum(1).tolist(), " per w (first 16):", per_w[:16].tolist(), end='')
            i = torch.nonzero(bad)[0].tolist(); print("  first", i, float(y[tuple(i)]), float(z[tuple(i)]), end='')
        print()
        ops.invalidate_packs()
