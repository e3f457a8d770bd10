"""Stand-alone timing of the AR-RFF fuse forward / backward at the bench size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arfe_amd import hip
from arfe_amd.hip import ptr, stream, check
from tools.kernel_bench import timeit
dev = torch.device('cuda:0')
R, C = 1024, 256
x = torch.randn(R, 3 * C, 7, 7, device=dev)
lw = torch.randn(R, C, 7, 7, device=dev).relu(); lh = torch.randn(R, C, 7, 7, device=dev).relu()
z = torch.empty(R, C, 7, 7, device=dev); dz = torch.randn_like(z)
dx = torch.empty_like(x); dlw = torch.empty_like(z); dlh = torch.empty_like(z)
L = hip.lib(); rs = 3 * C * 49
f = timeit(lambda: check(L.arfe_rff_fuse_fwd(R, C, ptr(x), rs, ptr(lw), ptr(lh), ptr(z), stream()), 'f'), iters=20)
b = timeit(lambda: check(L.arfe_rff_fuse_bwd(R, C, ptr(x), rs, ptr(lw), ptr(lh), ptr(dz), ptr(dx), rs, ptr(dlw), ptr(dlh), stream()), 'b'), iters=20)
mb = R * C * 49 * 4 / 1e6
print(f"fuse fwd {f*1e3:7.1f} us ({4*mb:.0f} MB)   fuse bwd {b*1e3:7.1f} us ({7*mb:.0f} MB -> {7*mb/5e3*1e3:.0f} us @5 TB/s)")
