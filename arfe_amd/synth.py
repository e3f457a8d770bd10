"""Synthetic inputs of the benchmark workload (BASELINE configs[2]: 1333x800 image, 5-level 256-channel pyramid,
512 RoIs per image).  Plain seeded CPU generators; the CPU checker keeps identical copies of its own
(tests/test_host_logic.py compares the two)."""
import numpy as np
import torch

PYRAMID_1333x800 = ((200, 336), (100, 168), (50, 84), (25, 42), (13, 21))
STRIDES = (4, 8, 16, 32, 64)


def synth_pyramid(batch, channels=256, sizes=PYRAMID_1333x800, seed=1234):
    gen = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, channels, h, w, generator=gen) for h, w in sizes]


def synth_rois(batch, per_img=512, img_w=1344, img_h=800, seed=1234):
    """x1,y1 ~ U(0,0.8*(W,H)); w,h = exp(U(ln16, ln512)) clipped to the image; batch idx = repeat."""
    gen = torch.Generator().manual_seed(seed + 7)
    n = batch * per_img
    x1 = torch.rand(n, generator=gen) * 0.8 * img_w
    y1 = torch.rand(n, generator=gen) * 0.8 * img_h
    lo, hi = float(np.log(16.0)), float(np.log(512.0))
    w = torch.exp(lo + torch.rand(n, generator=gen) * (hi - lo))
    h = torch.exp(lo + torch.rand(n, generator=gen) * (hi - lo))
    x2 = torch.minimum(x1 + w, torch.tensor(float(img_w - 1)))
    y2 = torch.minimum(y1 + h, torch.tensor(float(img_h - 1)))
    b = torch.arange(batch, dtype=torch.float32).repeat_interleave(per_img)
    return torch.stack([b, x1, y1, x2, y2], dim=1)
