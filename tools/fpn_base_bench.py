"""Timing of the base FPN at the BASELINE size: ResNet-50 stage outputs of a 1333x800 image
(B=2; 256x200x336, 512x100x168, 1024x50x84, 2048x25x42) -> 5 x 256-channel levels, forward + backward (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import arfe_amd
from arfe_amd import functional as Fn
dev = torch.device('cuda:0')
B = 2
chans, sizes = (256, 512, 1024, 2048), ((200, 336), (100, 168), (50, 84), (25, 42))
neck = arfe_amd.build_neck(dict(type='FPN', in_channels=list(chans), out_channels=256, num_outs=5)).to(dev)
neck.init_weights()
xs = [torch.randn(B, c, h, w, device=dev, requires_grad=True) for c, (h, w) in zip(chans, sizes)]
outs = neck(xs)
cots = [torch.randn_like(o) for o in outs]


def step():
    for p in neck.parameters():
        p.grad = None
    for x in xs:
        x.grad = None
    o = neck(xs)
    torch.autograd.backward(o, cots)


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flop_fwd = sum(2.0 * B * 256 * c * h * w for c, (h, w) in zip(chans, sizes)) + sum(2.0 * B * 256 * 256 * 9 * h * w for h, w in sizes)
print(f"base FPN fwd+bwd B={B}: {ms:.3f} ms/step = {B / ms * 1e3:.1f} imgs/s; algorithmic {3 * flop_fwd / 1e9:.0f} GFLOP/step "
      f"-> {3 * flop_fwd / ms / 1e9:.0f} TFLOP/s algorithmic ({9 * flop_fwd / ms / 1e9:.0f} executed on the bf16 pipe)")
with torch.no_grad():
    e0.record()
    for _ in range(n):
        neck(xs)
    e1.record()
    torch.cuda.synchronize()
print(f"forward only: {e0.elapsed_time(e1) / n:.3f} ms")
