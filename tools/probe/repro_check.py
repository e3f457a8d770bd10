"""run-to-run reproducibility of one step at the bench size: which gradients are bit-identical between two runs"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from arfe_amd import synth as O
dev = torch.device('cuda:0')
neck, roi_head = bench.build_modules(dev)
torch.manual_seed(1)
for p in list(neck.parameters()) + list(roi_head.parameters()):
    p.data.add_(0.01 * torch.randn_like(p))
xs = [x.to(dev).requires_grad_(True) for x in O.synth_pyramid(2, 256)]
rois = O.synth_rois(2, 512).to(dev)
cc, cr = torch.randn(1024, 81, device=dev), torch.randn(1024, 320, device=dev)
named = [('head.' + n, p) for n, p in roi_head.bbox_head.named_parameters()] + [('neck.' + n, p) for n, p in neck.named_parameters()]
def step():
    for _, p in named: p.grad = None
    for x in xs: x.grad = None
    res = roi_head._bbox_forward(neck(xs), rois)
    torch.autograd.backward([res['cls_score'], res['bbox_pred']], [cc, cr])
    torch.cuda.synchronize()
    return [res['cls_score'].detach().clone(), res['bbox_pred'].detach().clone()] + [x.grad.clone() for x in xs] + [p.grad.clone() for _, p in named]
a = step(); b = step(); c = step()
names = ['cls', 'reg'] + [f'dx{i}' for i in range(5)] + [n for n, _ in named]
diff = [(n, float((u - v).abs().max() / u.abs().max().clamp_min(1e-30))) for n, u, v, w in zip(names, a, b, c) if not (torch.equal(u, v) and torch.equal(u, w))]
print(f"{len(names) - len(diff)} of {len(names)} tensors bit-identical over three runs; differing:")
for n, e in diff: print(f"  {n}: {e:.2e}")
