"""robustness sweep: the whole path (neck -> 3-RoI extractor -> head, fwd+bwd) on unusual shapes against the oracle.
Per configuration (oracle/parity.py): forward outputs in max-norm against the free-running oracle, the ReLU / argmax choices
that differ from it COUNTED, every gradient in max-norm against the oracle evaluated under the device's choices.  Exits
non-zero when a forward exceeds 1e-4, a gradient 1e-3 (north_star's bound; `--grad <tol>` overrides) or more than
max(3, 20 ppm) of the choices differ."""
import sys
TOL = float(sys.argv[sys.argv.index('--grad') + 1]) if '--grad' in sys.argv else 1e-3
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
    from oracle import parity as P
    c1, c2 = torch.randn(R, 81, generator=g), torch.randn(R, 320, generator=g)
    xs = [x.clone().to(dev).requires_grad_(True) for x in xs_cpu]
    poison()
    got = P.device_step(neck, roi_head, xs, rois.to(dev), c1.to(dev), c2.to(dev), refine_level=r)
    rep = P.compare(got, xs_cpu, rois, {k: v.detach() for k, v in pn.items()}, {k: v.detach() for k, v in ph.items()}, c1, c2,
                    refine_level=r)
    ok = (rep['max_rel_fwd'] <= 1e-4 and rep['max_rel_grad_agreeing'] <= TOL and
          rep['mask_flips'] <= max(3, 20e-6 * rep['decisions']))
    print(f"case {ci}: B={B} C={C} R={R} refine={r}: fwd {rep['max_rel_fwd']:.2e}, grads {rep['max_rel_grad_agreeing']:.2e} "
          f"({rep['worst_grad']}), {rep['mask_flips']} of {rep['decisions']} choices differ, free-running L2 "
          f"{rep['l2_rel_grad_free_running_max']:.2e}{'' if ok else '   <-- FAIL'}", flush=True)
    worst = max(worst, 0.0 if ok else 1.0)
print('failures' if worst else 'all cases within bounds')
sys.exit(0 if worst == 0.0 else 1)
