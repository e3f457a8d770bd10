"""how long the HOST needs to enqueue one step (no device sync inside the loop) next to the device time per step: if the
two are close the step is launch-bound at times and the gaps in the kernel trace are host stalls."""
import sys, time
import torch
sys.path.insert(0, '.')
import bench
from oracle import arfe_oracle as O

dev = torch.device('cuda:0')
neck, roi_head = bench.build_modules(dev)
B, R = 2, 1024
xs = [x.to(dev).requires_grad_(True) for x in O.synth_pyramid(B, 256, seed=1234)]
rois = O.synth_rois(B, 512, seed=1234).to(dev)
g = torch.Generator().manual_seed(99)
c1 = torch.randn(R, 81, generator=g).to(dev)
c2 = torch.randn(R, 320, generator=g).to(dev)
params = list(neck.parameters()) + list(roi_head.parameters())


def step():
    for p in params:
        p.grad = None
    for x in xs:
        x.grad = None
    res = roi_head._bbox_forward(neck(xs), rois)
    torch.autograd.backward([res['cls_score'], res['bbox_pred']], [c1, c2])


for _ in range(10):
    step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'host enqueue {t_enq / n * 1e3:.3f} ms/step, device-complete {t_all / n * 1e3:.3f} ms/step')
