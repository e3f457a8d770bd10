"""In-kernel clock of the 3x3 conv kernel (development aid): needs a build with -DARFE_PROBES -DARFE_CLOCK_PROBE
(tools/probe/build_variant.sh clk gemm_ops.hip -DARFE_PROBES -DARFE_CLOCK_PROBE; ARFE_LIB=arfe_amd/libarfe_var_clk.so).
The kernel prints d(s_memtime) / d(s_memrealtime) x 100 MHz around its main loop from a few workgroups; this script launches it
back to back for about two seconds (the chip settles its clock under load) and the caller reads the last lines."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arfe_amd import hip, functional as Fn
from arfe_amd.hip import ptr, stream, check
dev = torch.device('cuda:0')
L = hip.lib()
R, C = 1024, 256
x = torch.randn(R, C, 7, 7, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
b = torch.zeros(C, device=dev)
pk = Fn.conv3x3_pack_op(w)
y = torch.empty(R, C, 7, 7, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
for _ in range(n):
    check(L.arfe_conv3x3_fwd(R, C, ptr(x), C * 49, ptr(pk), ptr(b), 1, ptr(y), C * 49, stream()), 'conv')
torch.cuda.synchronize()
