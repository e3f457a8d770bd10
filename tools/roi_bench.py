"""Micro-timing of the 3-set RoIAlign forward / backward at the bench size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arfe_amd import functional as Fn
from oracle import arfe_oracle as O
dev = torch.device('cuda:0')
B = 2
DT = torch.float16 if 'half' in sys.argv else torch.float32      # `half`: the fp16-mode entry points (half maps and features)
SN = 0                                                           # sample_num of the bench configuration (adaptive grid)
feats = [x.to(dev).to(DT) for x in O.synth_pyramid(B, 256)][:4]
rois = O.synth_rois(B, 512, seed=1234).to(dev)
h_rois, w_rois = Fn.make_rois_op(rois)
sets = [rois, h_rois, w_rois]
levels = [Fn.map_roi_levels_op(r, 4, 56) for r in sets]
R = rois.shape[0]
out = torch.empty(R, 3 * 256, 7, 7, device=dev, dtype=DT)
gout = torch.randn_like(out)
strides = [4, 8, 16, 32]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
es = feats[0].element_size()
mb_f = sum(f.numel() for f in feats) * es / 1e6
mb_o = out.numel() * es / 1e6
print(f"roialign_fwd {t(lambda: Fn.roialign_fwd_op(feats, strides, sets, levels, out, 7, SN)):8.1f} us  (features {mb_f:.0f} MB + out {mb_o:.0f} MB -> {(mb_f+mb_o)/6.3e3*1e3:.0f} us @6.3 TB/s)")
shapes = [f.shape for f in feats]
print(f"roialign_bwd {t(lambda: Fn.roialign_bwd_op(shapes, strides, sets, levels, gout, 7, SN)):8.1f} us")
