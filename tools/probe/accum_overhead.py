"""step time with parameter gradients accumulated into pre-existing (bucket-view-like) tensors against freshly installed
gradients: what the N > 1 reducer's in-place accumulation costs per step on one GPU (no communication involved)"""
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
params = list(roi_head.bbox_head.parameters()) + list(neck.parameters())
flat = torch.zeros(sum(p.numel() for p in params), device=dev)
views, off = [], 0
for p in params:
    views.append(flat[off:off + p.numel()].view_as(p)); off += p.numel()
def run(mode, n=30):
    def step():
        if mode == 'views':
            flat.zero_()
            for p, v in zip(params, views): p.grad = v
        else:
            for p in params: p.grad = None
        for x in xs: x.grad = None
        res = roi_head._bbox_forward(neck(xs), rois)
        torch.autograd.backward([res['cls_score'], res['bbox_pred']], [cc, cr])
    for _ in range(8): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for m in ('none', 'views', 'none', 'views'):
    print(m, f"{run(m):.3f} ms/step", flush=True)
# the reducer's own bookkeeping as at N > 1 (buckets, hooks, gather, fc1 sink), without a process group
from arfe_amd.dist import GradAllReducer
red = GradAllReducer([list(roi_head.bbox_head.parameters()), list(neck.parameters())],
                     early=[roi_head.bbox_head.shared_fcs[0].weight], world=2)
def rstep():
    red.zero_grad()
    for x in xs: x.grad = None
    res = roi_head._bbox_forward(neck(xs), rois)
    torch.autograd.backward([res['cls_score'], res['bbox_pred']], [cc, cr])
    red.synchronize()
for _ in range(8): rstep()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): rstep()
torch.cuda.synchronize(); print(f"reducer dry run {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
