"""debug aid: FPN -> AR-FPN forward+backward through autograd, repeated on FIXED weights and inputs; every result is
compared with the first iteration's (atomics allow ~1e-5 relative jitter, anything larger is a race)."""
import sys
import torch
sys.path.insert(0, '.')
import arfe_amd

dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator().manual_seed(31)
in_channels, sizes = (8, 16, 24, 32), ((40, 64), (20, 32), (10, 16), (5, 8))
neck = arfe_amd.build_neck([dict(type='FPN', in_channels=list(in_channels), out_channels=16, num_outs=5),
                            dict(type='WFPNDualSpatial', in_channels=16, num_levels=5)]).to(dev)
for m in neck:
    m.init_weights()
with torch.no_grad():
    for n, p in neck.named_parameters():
        if n.endswith('bias'):
            p.copy_(torch.randn(p.shape, generator=g).to(dev) * 0.1)
xs_cpu = [torch.randn(2, c, h, w, generator=g) for c, (h, w) in zip(in_channels, sizes)]
cots = None
first = None
for it in range(iters):
    neck.zero_grad(set_to_none=True)
    xs = [x.clone().to(dev).requires_grad_(True) for x in xs_cpu]
    pyr = neck[0](xs)
    for t in pyr:
        t.retain_grad()
    outs = neck[1](pyr)
    if cots is None:
        cots = [torch.randn(o.shape, generator=g).to(dev) for o in outs]
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    torch.cuda.synchronize()
    snap = {}
    for i in range(5):
        snap[f'pyr{i}'] = pyr[i].detach().cpu()
        snap[f'out{i}'] = outs[i].detach().cpu()
        snap[f'dpyr{i}'] = pyr[i].grad.cpu()
    for i in range(4):
        snap[f'dx{i}'] = xs[i].grad.cpu()
    for k, v in neck.named_parameters():
        snap['p.' + k] = v.grad.cpu()
    if first is None:
        first = snap
        continue
    for k, t in snap.items():
        f = first[k]
        d = (t - f).abs()
        tol = 2e-4 * max(f.abs().max().item(), 1e-30)
        if t.numel() == 1:
            tol = 1e-3 * max(f.abs().max().item(), 1.0)
        nbad = int((d > tol).sum())
        if nbad:
            idx = (d.flatten() > tol).nonzero().flatten()[:4].tolist()
            print(f'iter {it}: {k} {nbad}/{t.numel()} differ (max {d.max().item():.3e} of {f.abs().max().item():.3e}) idx {idx}',
                  flush=True)
print('done', iters)
