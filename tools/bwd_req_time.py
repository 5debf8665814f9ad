import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
lib = _lib.lib(); ops = default_ops()
wl = hotpath.HotPathWorkload("cfg2", mode="dropin"); wl.run_eager()
for l in (3, 2):
    n, c, h, w = hotpath.level_shapes(wl.N, wl.H, wl.W)[l]
    go = torch.randn(n, c, h, w, device="cuda")
    for req in (("write","write","null","null"), ("write","null","null","null"), ("null","write","null","null")):
        fn = lambda: ops.DeformableConvolution_backward(go, wl.t["c2_%d" % l], wl.o["offset%d" % l], wl.t["w_%d" % l], kernel=(3, 3), pad=(1, 1), req=req)
        for _ in range(3): fn()
        torch.cuda.synchronize(); lib.profile_reset(); lib.profile_enable(1)
        for _ in range(10): fn()
        lib.profile_enable(0); torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(8192); lib.profile_dump(buf, 8192); lib.profile_reset()
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            if "shared" in name: print("L%d" % l, req[:2], "%.1f us" % (float(ms) / int(cnt) * 1e3), flush=True)
