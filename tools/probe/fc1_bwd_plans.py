"""FC1-sized linear backward (data gradient, weight gradient) under the current ARFE_GEMM_PLAN"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arfe_amd import functional as Fn
from tools.kernel_bench import timeit, dev
R, i, o = 1024, 12544, 1024
x = torch.randn(R, i, device=dev).relu(); w = torch.randn(o, i, device=dev) * 0.02; dy = torch.randn(R, o, device=dev)
p = os.environ.get('ARFE_GEMM_PLAN', 'auto')
ms = timeit(lambda: Fn.linear_bwd_data_op(x, w, dy, mask_dx_by_x=True), iters=10)
print(f"{p:>5s} dgrad {ms*1e3:8.1f} us", flush=True)
ms = timeit(lambda: Fn.linear_bwd_weight_op(x, w, dy), iters=10)
print(f"{p:>5s} wgrad {ms*1e3:8.1f} us", flush=True)
