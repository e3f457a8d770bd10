"""Time the 3x3 conv weight-gradient GEMM at the bench size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arfe_amd import hip, functional as Fn
from tools.kernel_bench import timeit
dev = torch.device('cuda:0')
R, C = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 256
x = torch.randn(R, C, 7, 7, device=dev); dy = torch.randn(R, C, 7, 7, device=dev); w = torch.randn(C, C, 3, 3, device=dev)
ws = torch.empty(int(hip.lib().arfe_conv3x3_workspace_bytes(C)), dtype=torch.uint8, device=dev)
ms = timeit(lambda: Fn.conv3x3_bwd_weight_op(x.data_ptr(), C * 49, R, C, w, dy, ws), iters=20)
print(f"conv3x3 wgrad R={R}: {ms*1e3:8.1f} us  {2.0*C*C*9*R*49/ms/1e9:7.1f} TFLOP/s algorithmic")
