"""isolated timing of the 3-set RoIAlign backward at the bench configuration (5 levels, adaptive sampling, prepared tables)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arfe_amd import functional as Fn
from oracle import arfe_oracle as O
from tools.kernel_bench import timeit
dev = torch.device('cuda:0')
B = 2
feats = [x.to(dev) for x in O.synth_pyramid(B, 256)]
rois = O.synth_rois(B, 512, seed=1234).to(dev)
h_r, w_r, lv = Fn.rff_roi_sets_op(rois, 5)
sets = [rois, w_r, h_r]
gout = torch.randn(rois.shape[0], 768, 7, 7, device=dev)
shapes = [f.shape for f in feats]
ws = Fn.roialign_bwd_prepare_op(feats, O.STRIDES, sets, lv)
ms = timeit(lambda: Fn.roialign_bwd_op(shapes, O.STRIDES, sets, lv, gout, prepared=ws), iters=20)
print(f"roialign_bwd (prepared) {ms*1e3:8.1f} us  dbg={os.environ.get('ARFE_ROI_DBG','0')}")
out = torch.empty(rois.shape[0], 768, 7, 7, device=dev)
ms = timeit(lambda: Fn.roialign_fwd_op(feats, O.STRIDES, sets, lv, out), iters=20)
print(f"roialign_fwd            {ms*1e3:8.1f} us")
