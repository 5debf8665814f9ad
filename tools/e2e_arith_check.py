import sys
sys.path.insert(0, "/root/repo")
import torch, bench
from maskflownet_amd import _lib
for mode in (-1, 1, 0):
    _lib.set_arithmetic(all=mode)
    r = bench.end_to_end(8, 384, 512, "cuda:0", torch)
    r.pop("_flow", None)
    print(mode, r["value"], r["ms_per_forward"])
_lib.set_arithmetic(all=-1)
lib = _lib.lib()
lib.profile_reset(); lib.profile_enable(1)
from maskflownet_amd import network
net = network.MaskFlownetS(network.random_params(seed=1), 8, 384, 512, device="cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
net.set_input(torch.rand(8, 3, 384, 512, generator=g) - 0.5, torch.rand(8, 3, 384, 512, generator=g) - 0.5)
net.forward_eager() if hasattr(net, "forward_eager") else net.run_eager() if hasattr(net, "run_eager") else None
lib.profile_enable(0); torch.cuda.synchronize()
import ctypes
buf = ctypes.create_string_buffer(65536); lib.profile_dump(buf, 65536)
print(buf.value.decode()[:3000])
