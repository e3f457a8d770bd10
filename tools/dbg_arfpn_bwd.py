"""debug aid: repeats the AR-FPN backward op sequence (same stream usage as ARFPNFunction.backward) on fixed inputs and
reports which intermediate differs from the first iteration (the non-atomic ones must be bit-identical)."""
import sys
import torch
sys.path.insert(0, '.')
import arfe_amd
from arfe_amd import functional as Fn

dev = torch.device('cuda:0')
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sizes = ((40, 64), (20, 32), (10, 16), (5, 8), (3, 4))
g = torch.Generator().manual_seed(3)
neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=C, num_levels=5)).to(dev)
neck.init_weights()
with torch.no_grad():
    for n, p in neck.named_parameters():
        if n.endswith('bias'):
            p.copy_(torch.randn(p.shape, generator=g).to(dev) * 0.1)
    neck.refine.conv_out.conv.weight.copy_(torch.randn(C, C, 1, 1, generator=g).to(dev) * 0.2)
xs = [torch.randn(2, C, h, w, generator=g).to(dev) for h, w in sizes]
gouts = [torch.randn(2, C, h, w, generator=g).to(dev) for h, w in sizes]
w1 = [m.conv.weight.detach() for m in neck.reduce_convs]
b1 = [m.conv.bias.detach() for m in neck.reduce_convs]
w2 = [m.conv.weight.detach() for m in neck.reduce_convs2]
b2 = [m.conv.bias.detach() for m in neck.reduce_convs2]
wt, bt, wp, bp, wg, bg, wo, bo = [t.detach().contiguous() for t in neck.refine.hip_params()]
r = 2


def fwd():
    a, b = Fn.on_side(dev, lambda: Fn.fpn_gatemap_op(xs, w1, b1, w2, b2, r))
    m, argmax = Fn.fpn_gather_op(xs, r, want_argmax=True)
    bsf, saved = Fn.nonlocal_fwd_op(m, wt, bt, wp, bp, wg, bg, wo, bo)
    Fn.join_side(dev)
    outs = Fn.fpn_scatter_op(xs, bsf, a, b, r)
    return a, b, m, argmax, bsf, saved, outs


def bwd(a, b, m, argmax, bsf, saved):
    dpa, dpb, dbsf = Fn.fpn_scatter_bwd_op(gouts, bsf, a, b, r)
    dw1, db1, dw2, db2 = Fn.on_side(dev, lambda: Fn.fpn_gate_wgrad_op(xs, dpa, dpb, r))
    dm, nlg = Fn.nonlocal_bwd_op(dbsf, m, wt, wp, wg, wo, saved)
    dxs = Fn.fpn_input_bwd_op(gouts, dpa, dpb, w1, w2, dm, argmax, r)
    Fn.join_side(dev)
    return dict(dpa=dpa, dpb=dpb, dbsf=[dbsf], dm=[dm], dxs=dxs, dw1=dw1, db1=db1, dw2=dw2, db2=db2)


exact = ('dpa', 'dpb', 'dbsf', 'a', 'b', 'outs')
first = None
for it in range(iters):
    a, b, m, argmax, bsf, saved, outs = fwd()
    res = bwd(a, b, m, argmax, bsf, saved)
    res.update(a=a, b=b, outs=outs, bsf=[bsf])
    torch.cuda.synchronize()
    snap = {k: [t.detach().cpu().clone() for t in v] for k, v in res.items()}
    if first is None:
        first = snap
        continue
    for k, v in snap.items():
        for i, t in enumerate(v):
            f = first[k][i]
            d = (t - f).abs()
            tol = 0.0 if k in exact else 2e-4 * max(f.abs().max().item(), 1e-30)
            nbad = int((d > tol).sum())
            if nbad:
                idx = (d.flatten() > tol).nonzero().flatten()[:4].tolist()
                print(f'iter {it}: {k}[{i}] {nbad}/{t.numel()} differ (max {d.max().item():.3e} of {f.abs().max().item():.3e}) idx {idx} '
                      f'got {t.flatten()[idx].tolist()} first {f.flatten()[idx].tolist()}', flush=True)
print('done', iters)
