"""large GEMM shapes under the current ARFE_GEMM_PLAN (mid 128x128 vs large 256x256 tile)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.kernel_bench import timeit, dev, L, ptr, stream, check
for (R, i, o) in ((4200, 256, 4200), (8400, 256, 4200), (4096, 4096, 4096), (50176, 2304, 256), (8192, 1024, 1024)):
    x = torch.randn(R, i, device=dev); w = torch.randn(o, i, device=dev) * 0.02; b = torch.zeros(o, device=dev)
    y = torch.empty(R, o, device=dev)
    ms = timeit(lambda: check(L.arfe_linear_fwd(R, i, o, ptr(x), ptr(w), ptr(b), 0, ptr(y), 0, 0, stream()), 'lin'), iters=10)
    print(f"{os.environ.get('ARFE_GEMM_PLAN', 'auto'):>5s}  R={R:5d} in={i:5d} out={o:4d}: {ms*1e3:8.1f} us  {2.0*R*i*o/ms/1e9:6.1f} TF alg", flush=True)
