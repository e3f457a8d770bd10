"""RoIAlign forward/backward timing with and without the RoIs whose window exceeds the 64-entry weight tables (NOBIG=1)"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arfe_amd import functional as Fn
from oracle import arfe_oracle as O
from tools.kernel_bench import timeit
dev = torch.device('cuda:0')
B = 2
feats = [x.to(dev) for x in O.synth_pyramid(B, 256)]
rois = O.synth_rois(B, 512, seed=1234)
if os.environ.get('NOBIG') == '1':
    Hs = [200, 100, 50, 25]; Ws = [336, 168, 84, 42]
    h, w = O.make_rois(rois)
    bad = torch.zeros(rois.shape[0], dtype=torch.bool)
    for r in (rois, w, h):
        lv = O.map_roi_levels(r, 4)
        for i in range(r.shape[0]):
            l = int(lv[i]); sc = 1.0 / (4 * 2 ** l)
            ex = (float(r[i, 3]) - float(r[i, 1])) * sc; ey = (float(r[i, 4]) - float(r[i, 2])) * sc
            if ex > 56 or ey > 56: bad[i] = True
    good = rois[~bad]
    print("replacing", int(bad.sum()), "rois")
    rois[bad] = good[:int(bad.sum())]
rois = rois.to(dev)
h_r, w_r, lv = Fn.rff_roi_sets_op(rois, 5)
sets = [rois, w_r, h_r]
gout = torch.randn(rois.shape[0], 768, 7, 7, device=dev)
shapes = [f.shape for f in feats]
ws = Fn.roialign_bwd_prepare_op(feats, O.STRIDES, sets, lv)
ms = timeit(lambda: Fn.roialign_bwd_op(shapes, O.STRIDES, sets, lv, gout, prepared=ws), iters=20)
print(f"roialign_bwd (prepared) {ms*1e3:8.1f} us")
out = torch.empty(rois.shape[0], 768, 7, 7, device=dev)
ms = timeit(lambda: Fn.roialign_fwd_op(feats, O.STRIDES, sets, lv, out), iters=20)
print(f"roialign_fwd            {ms*1e3:8.1f} us")
