"""Whole-detector step of BASELINE configs #2/#3/#5 on synthetic COCO-shaped input (development aid; bench.py measures the
hot path alone): Faster R-CNN R50 (or Cascade R-CNN R50/R101) + [FPN, AR-FPN] + AR-RFF, bs=2, 800 x 1344, forward_train +
backward.  usage: detector_bench.py [FasterRCNN|CascadeRCNN] [depth] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import arfe_amd
from arfe_amd import detector as D
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_harness import _small_detector

kind = sys.argv[1] if len(sys.argv) > 1 else 'FasterRCNN'
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 50
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device('cuda:0')
det = _small_detector(kind, dev)
# the AR configs' full budgets: 2000 pre-NMS / 1000 post-NMS proposals per image, 512 sampled RoIs per image
det.train_cfg['rpn_proposal'].update(nms_pre=2000, nms_post=1000, max_num=1000)
for s in ([det.roi_head.bbox_sampler] if kind == 'FasterRCNN' else det.roi_head.bbox_sampler):
    s.num = 512
if depth != 50:
    det.backbone = arfe_amd.build_backbone(dict(type='ResNet', depth=depth, frozen_stages=1, norm_eval=True)).to(dev).train()
    det.backbone.init_weights()
img, metas, gtb, gtl = D.synthetic_batch(batch=2, height=800, width=1344, max_gt=20, seed=1, device=dev)
params = [p for p in det.parameters() if p.requires_grad]


def step():
    for p in params:
        p.grad = None
    loss, _ = D.parse_losses(det.forward_train(img, metas, gtb, gtl))
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{kind} R{depth} + AR-FPN + AR-RFF, bs=2 @800x1344, fwd+bwd: {dt * 1e3:.1f} ms/step = {2 / dt:.2f} imgs/s")
