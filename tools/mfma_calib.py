"""Calibration (development aid): what a library bf16 GEMM reaches on this chip for the MFMA work of one conv3x3 GEMM
launch in split-precision form (M=256, N=R*49, K=3*2304), to separate 'clock / matrix pipe' limits from staging overhead."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import timeit, conv
dev = torch.device('cuda:0')
for (M, N, K) in ((256, 50176, 6912), (256, 50176, 2304), (4096, 4096, 4096), (8192, 8192, 8192), (1024, 1024, 37632)):
    a = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, M, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: torch.matmul(a, b), iters=20)
    print(f"hipBLASLt bf16 [{N}x{K}]x[{K}x{M}]: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s")
conv(1024)
