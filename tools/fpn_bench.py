"""Micro-timing of the AR-FPN stage ops at the bench size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arfe_amd import functional as Fn
from oracle import arfe_oracle as O
dev = torch.device('cuda:0')
B = 2
xs = [x.to(dev) for x in O.synth_pyramid(B, 256)]
g = [torch.randn_like(x) for x in xs]
p = {k: v.to(dev) for k, v in O.arfpn_params(256, 5, seed=1).items()}
w1 = [p[f'reduce_convs.{i}.conv.weight'] for i in range(5)]; b1 = [p[f'reduce_convs.{i}.conv.bias'] for i in range(5)]
w2 = [p[f'reduce_convs2.{i}.conv.weight'] for i in range(5)]; b2 = [p[f'reduce_convs2.{i}.conv.bias'] for i in range(5)]
m, am = Fn.fpn_gather_op(xs, 2)
bsf = torch.randn_like(m)
a, b = Fn.fpn_gatemap_op(xs, w1, b1, w2, b2, 2)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
dpa, dpb, dbsf = Fn.fpn_scatter_bwd_op(g, bsf, a, b, 2)
MB = 91.67 * B
print(f"gather     {t(lambda: Fn.fpn_gather_op(xs, 2)):8.1f} us   (ideal {MB/6.3e3*1e3:.0f} us @6.3TB/s for {MB:.0f} MB)")
print(f"gatemap    {t(lambda: Fn.fpn_gatemap_op(xs, w1, b1, w2, b2, 2)):8.1f} us   ({MB:.0f} MB)")
print(f"gath+gate  {t(lambda: Fn.fpn_gather_gate_op(xs, w1, b1, w2, b2, 2)):8.1f} us   ({MB:.0f} MB)")
print(f"scatter    {t(lambda: Fn.fpn_scatter_op(xs, bsf, a, b, 2)):8.1f} us   ({2*MB:.0f} MB)")
print(f"scat_bwd   {t(lambda: Fn.fpn_scatter_bwd_op(g, bsf, a, b, 2)):8.1f} us   ({2*MB:.0f} MB)")
print(f"input_bwd  {t(lambda: Fn.fpn_input_bwd_op(g, dpa, dpb, w1, w2, dbsf, am, 2)):8.1f} us   ({2*MB:.0f} MB)")
print(f"gate_wgrad {t(lambda: Fn.fpn_gate_wgrad_op(xs, dpa, dpb, 2)):8.1f} us   ({MB:.0f} MB)")
