"""robustness sweep: the whole path (neck -> 3-RoI extractor -> head, fwd+bwd) on unusual shapes against the oracle.
Reports max-norm and L2 relative errors per configuration; exits non-zero on a mismatch beyond the L2 bound
(2e-3, the exact-fp32 mode's; `--l2 5e-3` for the default bf16x3 kernels, whose extra error on the widest cases is
ReLU-kink flips in the conv bias gradients)."""
import sys
TOL = float(sys.argv[sys.argv.index('--l2') + 1]) if '--l2' in sys.argv else 2e-3
import torch
sys.path.insert(0, '.')
import arfe_amd
from oracle import arfe_oracle as O

dev = torch.device('cuda:0')
CASES = [
    # B, C, fc, sizes, R, refine
    (1, 20, 40, ((37, 53), (19, 27), (10, 14), (5, 7), (3, 4)), 17, 2),
    (3, 36, 24, ((48, 40), (24, 20), (12, 10), (6, 5), (3, 3)), 65, 1),
    (2, 128, 64, ((64, 96), (32, 48), (16, 24), (8, 12), (4, 6)), 200, 3),
    (1, 8, 16, ((21, 30), (11, 15), (6, 8), (3, 4), (2, 2)), 1, 0),
    (2, 320, 48, ((24, 36), (12, 18), (6, 9), (3, 5), (2, 3)), 96, 4),
]


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item(), ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def poison():
    """NaN-fill the caching allocator's free blocks on both streams: any output element a kernel leaves unwritten (or any
    read of uninitialised workspace) then shows up as NaN in the comparison."""
    from arfe_amd import functional as Fn
    for st in (torch.cuda.current_stream(), Fn.side_stream(dev)):
        with torch.cuda.stream(st):
            junk = [torch.full((1 << 22,), float('nan'), device=dev) for _ in range(48)]
            junk += [torch.full((n,), float('nan'), device=dev) for n in (64, 1024, 16384, 262144) for _ in range(32)]
            del junk
    torch.cuda.synchronize()


worst = 0.0
for ci, (B, C, fc, sizes, R, r) in enumerate(CASES):
    g = torch.Generator().manual_seed(100 + ci)
    torch.manual_seed(100 + ci)
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=C, num_levels=5, refine_level=r))
    neck.init_weights()
    roi_head = arfe_amd.build_head(dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0),
                                out_channels=C, featmap_strides=[4, 8, 16, 32, 64]),
        bbox_head=dict(type='MultiRoIsBBoxHead', in_channels=C, conv_out_channels=C, fc_out_channels=fc, roi_feat_size=7,
                       num_classes=80)))
    roi_head.init_weights()
    with torch.no_grad():
        for n, p in list(neck.named_parameters()) + list(roi_head.named_parameters()):
            if n.endswith('bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        neck.refine.conv_out.conv.weight.copy_(torch.randn(C, C, 1, 1, generator=g) * 0.05)
    pn = {k: v.detach().clone().requires_grad_(True) for k, v in neck.state_dict().items()}
    ph = {k: v.detach().clone().requires_grad_(True) for k, v in roi_head.bbox_head.state_dict().items()}
    neck, roi_head = neck.to(dev), roi_head.to(dev)
    xs_cpu = [torch.randn(B, C, h, w, generator=g) for h, w in sizes]
    H0, W0 = sizes[0][0] * 4, sizes[0][1] * 4
    x1 = torch.rand(R, generator=g) * W0 * 0.8
    y1 = torch.rand(R, generator=g) * H0 * 0.8
    w = torch.exp(torch.rand(R, generator=g) * 3.5 + 1.5)
    h = torch.exp(torch.rand(R, generator=g) * 3.5 + 1.5)
    rois = torch.stack([torch.randint(0, B, (R,), generator=g).float(), x1, y1, x1 + w, y1 + h], 1)
    xr = [x.clone().requires_grad_(True) for x in xs_cpu]
    feats_r = O.arfpn_forward(xr, pn, r)
    cls_r, reg_r = O.rff_head(O.roi_extract3(feats_r, rois), ph)
    c1, c2 = torch.randn(cls_r.shape, generator=g), torch.randn(reg_r.shape, generator=g)
    ((cls_r * c1).sum() + (reg_r * c2).sum()).backward()
    xs = [x.clone().to(dev).requires_grad_(True) for x in xs_cpu]
    poison()
    res = roi_head._bbox_forward(neck(xs), rois.to(dev))
    ((res['cls_score'] * c1.to(dev)).sum() + (res['bbox_pred'] * c2.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    items = [('cls', res['cls_score'], cls_r), ('reg', res['bbox_pred'], reg_r)]
    items += [(f'dx{i}', xs[i].grad, xr[i].grad if xr[i].grad is not None else torch.zeros_like(xr[i])) for i in range(5)]
    items += [('neck.' + k, v.grad, pn[k].grad) for k, v in neck.named_parameters() if k != 'refine.phi.conv.bias']
    items += [('head.' + k, v.grad, ph[k].grad) for k, v in roi_head.bbox_head.named_parameters()]
    wmax = wl2 = 0.0
    wname = ''
    for name, a, b in items:
        if b is None:
            b = torch.zeros_like(a).cpu()
        m, l2 = rel(a, b)
        if b.numel() > 1 and l2 > wl2:
            wl2, wname = l2, name
        wmax = max(wmax, m if b.numel() > 1 else 0.0)
    print(f'case {ci}: B={B} C={C} R={R} refine={r}: worst L2 {wl2:.2e} ({wname}), worst max-norm {wmax:.2e}', flush=True)
    worst = max(worst, wl2)
print('worst L2', worst)
sys.exit(0 if worst <= TOL else 1)
