"""Implicit host synchronisations inside one bench step (development aid): runs the step of bench.py under
torch.cuda.set_sync_debug_mode('warn') and prints every warning once.  A synchronising call in the step (an .item(), a
nonzero, a host-side shape decision) lets the GPU run dry while the host catches up; the step is meant to have none."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from oracle import arfe_oracle as O
dev = torch.device('cuda:0')
neck, roi_head = bench.build_modules(dev)
xs = [x.to(dev).requires_grad_(True) for x in O.synth_pyramid(2, 256, seed=1234)]
rois = O.synth_rois(2, 512, seed=1234).to(dev)
g = torch.Generator().manual_seed(99)
cot_cls, cot_reg = torch.randn(1024, 81, generator=g).to(dev), torch.randn(1024, 320, generator=g).to(dev)


def step():
    for p in list(neck.parameters()) + list(roi_head.parameters()):
        p.grad = None
    feats = neck(xs)
    res = roi_head._bbox_forward(feats, rois)
    torch.autograd.backward([res['cls_score'], res['bbox_pred']], [cot_cls, cot_reg])


for _ in range(3):
    step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    step()
    step()
torch.cuda.set_sync_debug_mode('default')
torch.cuda.synchronize()
seen = set()
for x in w:
    key = (str(x.message)[:120], x.filename, x.lineno)
    if key not in seen:
        seen.add(key)
        print(f'{x.filename}:{x.lineno}: {x.message}')
print(f'{len(w)} synchronisation warnings in 2 steps ({len(seen)} distinct)')
