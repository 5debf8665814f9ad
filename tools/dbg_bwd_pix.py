import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
lib = _lib.lib(); ops = default_ops()
wl = hotpath.HotPathWorkload("cfg2", mode="dropin")
wl.run_eager()
for l in (5, 4, 3, 2):
    n, c, h, w = hotpath.level_shapes(8, 384, 512)[l]
    off = wl.o["offset%d" % l]
    go = torch.randn(n, c, h, w, device="cuda")
    res = {}
    for pix in (0, 1):
        _lib.set_tuning(dc_bwdpix=pix)
        got = ops.DeformableConvolution_backward(go, wl.t["c2_%d" % l], off, wl.t["w_%d" % l], kernel=(3, 3), pad=(1, 1), req=("write", "write", "null", "null"))
        torch.cuda.synchronize()
        res[pix] = [g.clone() for g in got[:2]]
        if pix == 1:
            ws = list(ops._ws.values())
            fl = ws[0].view(torch.int32)[: n * ((h + 3) // 4) * ((w + 7) // 8)].cpu().numpy()
            print("L%d workspaces %d flags mean %.3f" % (l, len(ws), fl.mean()), fl[:32].tolist())
    for i, nm in enumerate(("gx", "goffset")):
        a, b = res[0][i], res[1][i]
        print("   %s max|old-new| / max|old| = %.3e" % (nm, (a - b).abs().max().item() / a.abs().max().item()))
