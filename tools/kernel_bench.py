"""Micro-timings of single C-ABI calls on the GPU (development aid; not part of the product path)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arfe_amd import hip, functional as Fn
from arfe_amd.hip import ptr, stream, check

dev = torch.device('cuda:0')
L = hip.lib()


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def conv(R, C=256):
    x = torch.randn(R, C, 7, 7, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    b = torch.zeros(C, device=dev)
    pk = Fn.conv3x3_pack_op(w)
    y = torch.empty(R, C, 7, 7, device=dev)
    ms = timeit(lambda: check(L.arfe_conv3x3_fwd(R, C, ptr(x), C * 49, ptr(pk), ptr(b), 1, ptr(y), C * 49, stream()), 'conv'))
    fl = 2.0 * C * C * 9 * R * 49
    print(f"conv3x3 fwd R={R}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  tiles={2*((R*49+127)//128)}")


def linear(R, i, o):
    x = torch.randn(R, i, device=dev); w = torch.randn(o, i, device=dev) * 0.02; b = torch.zeros(o, device=dev)
    y = torch.empty(R, o, device=dev)
    ms = timeit(lambda: check(L.arfe_linear_fwd(R, i, o, ptr(x), ptr(w), ptr(b), 0, ptr(y), 0, 0, stream()), 'lin'))
    fl = 2.0 * R * i * o
    print(f"linear fwd R={R} in={i} out={o}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    for R in (700, 940, 980, 1024, 1100, 1960, 2048):
        conv(R)
    linear(4096, 4096, 4096)
    linear(1024, 12544, 1024)
    linear(1024, 1024, 1024)
    linear(3072, 4096, 4096)
