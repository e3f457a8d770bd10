"""isolated timings of the small / medium contractions of the path through their C-ABI calls"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.kernel_bench import timeit, dev, L, ptr, stream, check

def conv1x1(B, cin, cout, hw):
    x = torch.randn(B, cin, hw, device=dev); w = torch.randn(cout, cin, device=dev) * 0.05; b = torch.zeros(cout, device=dev)
    y = torch.empty(B, cout, hw, device=dev)
    ms = timeit(lambda: check(L.arfe_conv1x1_fwd(B, cin, cout, hw, ptr(x), ptr(w), ptr(b), ptr(y), 0, 0, stream()), 'c1'), iters=20)
    print(f"conv1x1 fwd B={B} cin={cin} cout={cout} hw={hw}: {ms*1e3:7.1f} us  {2.0*B*cin*cout*hw/ms/1e9:6.1f} TF alg", flush=True)

def linear(R, i, o, relu=0):
    x = torch.randn(R, i, device=dev); w = torch.randn(o, i, device=dev) * 0.02; b = torch.zeros(o, device=dev)
    y = torch.empty(R, o, device=dev)
    ms = timeit(lambda: check(L.arfe_linear_fwd(R, i, o, ptr(x), ptr(w), ptr(b), relu, ptr(y), 0, 0, stream()), 'lin'), iters=20)
    print(f"linear fwd R={R} in={i} out={o} relu={relu}: {ms*1e3:7.1f} us  {2.0*R*i*o/ms/1e9:6.1f} TF alg", flush=True)

conv1x1(2, 256, 256, 4200)
conv1x1(2, 256, 768, 4200)
conv1x1(2, 768, 256, 4200)
conv1x1(2, 256, 256, 16800)
conv1x1(1, 256, 256, 128 * 33)
linear(1024, 1024, 1024)
linear(1024, 1024, 1024, 1)
linear(1024, 1024, 81)
linear(1024, 1024, 320)
linear(8400, 256, 256)
linear(1024, 12544, 1024)
