"""2 ranks over gloo on one GPU: the bench step with the early fc1 sink; all ranks see the same data, so the averaged gradient of
every parameter must equal a single-process run's (launch: torch.distributed.run --nproc-per-node 2, ARFE_DIST_BACKEND=gloo)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import bench
from arfe_amd import synth as O
from arfe_amd.dist import GradAllReducer
dist.init_process_group(backend='gloo')
rank = dist.get_rank()
dev = torch.device('cuda:0')
torch.manual_seed(0)
neck, roi_head = bench.build_modules(dev)
torch.manual_seed(1)
for p in list(neck.parameters()) + list(roi_head.parameters()):          # non-trivial weights (conv_out is zero after init)
    p.data.add_(0.01 * torch.randn_like(p))
xs = [x.to(dev).requires_grad_(True) for x in O.synth_pyramid(1, 256, sizes=((40, 64), (20, 32), (10, 16), (5, 8), (3, 4)))]
rois = torch.tensor([[0, 8., 8., 120., 90.], [0, 30., 20., 200., 150.], [0, 5., 100., 250., 140.], [0, 100., 10., 140., 150.]] * 16, device=dev)
cc, cr = torch.ones(64, 81, device=dev), torch.ones(64, 320, device=dev)
head_params = list(roi_head.bbox_head.parameters())
fc1 = roi_head.bbox_head.shared_fcs[0].weight
# single-process reference first (no reducer, no hooks, no sink): the step's plain gradients
res = roi_head._bbox_forward(neck(xs), rois)
torch.autograd.backward([res['cls_score'], res['bbox_pred']], [cc, cr])
torch.cuda.synchronize()
ref = {n: p.grad.detach().clone() for n, p in roi_head.bbox_head.named_parameters()}
for p in head_params + list(neck.parameters()): p.grad = None
red = GradAllReducer([head_params, list(neck.parameters())], early=[fc1])
def step():
    red.zero_grad()
    for x in xs: x.grad = None
    res = roi_head._bbox_forward(neck(xs), rois)
    torch.autograd.backward([res['cls_score'], res['bbox_pred']], [cc, cr])
    red.synchronize()
step(); step()
torch.cuda.synchronize()
got = {n: p.grad.detach().clone() for n, p in list(roi_head.bbox_head.named_parameters())}
worst = 0.0
for n, p in roi_head.bbox_head.named_parameters():
    e = float((got[n] - ref[n]).abs().max() / ref[n].abs().max().clamp_min(1e-30))
    worst = max(worst, e)
    if e > 1e-4: print(f'rank {rank}: {n} off by {e:.3e}', flush=True)
print(f"rank {rank}: sink used for fc1: {len(red.sinks) == 1}; worst relative difference to the single-process gradients {worst:.2e}", flush=True)
assert worst < 1e-4 and len(red.sinks) == 1
dist.barrier(); dist.destroy_process_group()
