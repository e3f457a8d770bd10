"""wall time of the first steps of the bench workload, one synchronised step at a time (warm-up behaviour)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from arfe_amd import synth as O
dev = torch.device('cuda:0')
neck, roi_head = bench.build_modules(dev)
xs = [x.to(dev).requires_grad_(True) for x in O.synth_pyramid(2, 256)]
rois = O.synth_rois(2, 512).to(dev)
cc, cr = torch.randn(1024, 81, device=dev), torch.randn(1024, 320, device=dev)
ts = []
for i in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for p in list(neck.parameters()) + list(roi_head.parameters()): p.grad = None
    for x in xs: x.grad = None
    feats = neck(xs)
    res = roi_head._bbox_forward(feats, rois)
    torch.autograd.backward([res['cls_score'], res['bbox_pred']], [cc, cr])
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(' '.join(f'{t:.1f}' for t in ts))
print('reserved MB', torch.cuda.memory_reserved() / 1e6)
